"""torch.autograd.Function wrappers: the training-mode graph of the MASIC path, every node a HIP launch.

The reference relies on torch autograd over ATen ops (`loss.backward()` at newtrain_codec_real.py:141).  Here
each differentiable op of HSIC.forward is a Function whose forward AND backward call the C ABI; torch's autograd
engine only orders the calls and accumulates parameter gradients (`AccumulateGrad`), so optimizers, DDP hooks and
`zero_grad` work unchanged.  Formulas: SURVEY.md appendix B (checked there against autograd in float64).
"""
import math
import os

import torch
from torch.autograd import Function

from . import ops
from ._lib import PREC_BF16, PREC_F32

PEDESTAL = 2.0 ** -36


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_DGRAD_F16K = os.environ.get("MASIC_DGRAD_F16K", "1") != "0"      # 0: input gradients on the implicit-GEMM kernel only (A/B timing)


def _gemm_1x1(mod, act=0):
    from . import nn as _mnn
    return (_GEMM_1X1 and _mnn._PRECISION != PREC_F32 and tuple(mod.kernel_size) == (1, 1) and tuple(mod.stride) == (1, 1) and tuple(mod.padding) == (0, 0)
            and mod.in_channels % 32 == 0 and mod.out_channels % 32 == 0 and act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY))


_FWD_F16K = os.environ.get("MASIC_TRAIN_FWD_F16K", "1") != "0"      # 0: training-mode forward of the hyper / context layers on the NCHW kernel (A/B timing)


def _fwd_f16k(mod, x, act):
    from . import nn as _mnn
    if not _FWD_F16K or _mnn._PRECISION == PREC_F32 or _mnn._FP8 or x.dim() != 4 or x.shape[1] != mod.in_channels:
        return False
    kh, kw, _, _ = mod._geometry()
    return (kh * kw > 1 and mod.in_channels % 16 == 0 and mod.out_channels % 32 == 0 and act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY)
            and mod.f16k_supported(x.shape[0], x.shape[2], x.shape[3]))


_GEMM_1X1 = os.environ.get("MASIC_TRAIN_GEMM_1X1", "1") != "0"      # 0: 1x1 layers of the training step on the implicit-GEMM kernel (A/B timing)


class ConvFn(Function):
    """y = act(conv(x, W) + b) for the module's Conv2d / ConvTranspose2d / MaskedConv2d.
    dx: the forward kernel run as the opposite layer kind on the same weight tensor (a Conv2d's input gradient is a
    ConvTranspose2d with Cin/Cout swapped and vice versa); dW: masic_conv2d_wgrad; db: per-channel sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, mod, act):
        x = _c(x)
        if _gemm_1x1(mod, act):
            # bf16-operand mode, 1x1 layers of the entropy-parameter stacks: the DMA-staged GEMM of the inference path (conv_f16k.hip:
            # gemm_f16k, 4x the rate of the NCHW implicit-GEMM kernel at these shapes); x is converted once to F16K
            B, _, H, W = x.shape
            x16 = ops.nchw_to_f16k(x)
            y = ops.gemm_f16k(x16, mod.packed_gemm_dma_weight(), None if bias is None else bias.detach(), B, mod.in_channels,
                              mod.out_channels, H, W, act, want_nchw=True)
            ctx.x16 = x16 if _WGRAD1_F16K else None          # kept for the weight gradient (masic_gemm_wgrad_f16k)
        elif _fwd_f16k(mod, x, act):
            # bf16 mode, the hyper transforms and context models (3x3 / 5x5, 128 ... 384 channels at latent resolution): the DMA-staged
            # F16K kernel of inference (x converted once; the 3x3 weight gradient reads the same F16K copy) instead of the NCHW
            # implicit-GEMM kernel -- 38 against 54 ... 70 us per layer at 8 x 32 x 32
            B, _, H, W = x.shape
            x16 = ops.nchw_to_f16k(x)
            y = mod.run_f16k(x16, B, H, W, act=act, want_nchw=True)[0]
            ctx.x16 = x16
        else:
            y = mod.run(x, act=act)
            ctx.x16 = None
        ctx.mod, ctx.act = mod, act
        ctx.save_for_backward(x, weight, y if act != ops.ACT_NONE else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        gx, gw, gb = conv_backward(ctx.mod, x, weight, y, g, ctx.act, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                   ctx.has_bias and ctx.needs_input_grad[2], x16=ctx.x16)
        ctx.x16 = None
        return gx, gw, gb, None, None


_PIC_END_DGRAD = os.environ.get("MASIC_PIC_END_DGRAD", "1") != "0"   # 0: input gradients of g_a_conv1 / g_s_conv4 on the float32 NCHW kernels (A/B timing)
_GDN_BWD_SMALL = os.environ.get("MASIC_GDN_BWD_SMALL", "1") != "0"   # 0: GDN(3) backward as the nine-launch generic chain (A/B timing)
_WGRAD5_F16K = os.environ.get("MASIC_WGRAD5_F16K", "1") != "0"     # 0: 5x5 stride-1 weight gradients on the float32-tile kernel (A/B timing)
_DGRAD_FEW = os.environ.get("MASIC_DGRAD_FEW", "1") != "0"         # 0: input gradients of the <= 8-channel-input layers on the float32 direct kernel (A/B timing)
_PIC_WGRAD = os.environ.get("MASIC_PIC_WGRAD", "1") != "0"         # 0: weight gradients of g_a_conv1 / g_s_conv4 on the float32-tile kernel (A/B timing)
_WGRAD1_F16K = os.environ.get("MASIC_WGRAD1_F16K", "1") != "0"     # 0: 1x1 weight gradients on the float32 NCHW kernel (A/B timing)
_WGRAD3_F16K = os.environ.get("MASIC_WGRAD3_F16K", "1") != "0"     # 0: 3x3 weight gradients on the tap-generic float32-tile kernel (A/B timing)


def conv_backward(mod, x, weight, y, g, act, need_gx=True, need_gw=True, need_gb=True, x16=None, g16=None, gb=None, gx_f16k=False,
                  x_shape=None, g_shape=None):
    """(dx, dW, db) of y = act(conv(x, W) + b) for the module's layer geometry; x, g float32 NCHW (x16: x in F16K if the caller has it).
    g16 / gb: dy in F16K and its channel sums when the caller's producer already wrote them (ops.gdn_bwd_fused_ex): no conversion and
    no reduction pass here.  gx_f16k: return dx as an F16K buffer (int16) when the F16K kernel computes it (the consumer is another
    F16K-operand kernel), float32 NCHW otherwise.
    x = None / g = None with x_shape / g_shape: the caller has that tensor in F16K only (x16 / g16 + gb) -- the picture-end layers, whose
    weight gradient reads the 128-channel operand in F16K (ops.pic_wgrad_f16k) and never needs its float32 NCHW form.
    (Issuing dW / db on a side stream next to dx was measured twice: 24.9 -> 25.2 ms per HSIC training step in round 2, 13.14 -> 13.00 in
    round 3 and 12.33 -> 12.62 on top of the two-stream forward -- the eager step is bound by the host's launch rate, and a gradient
    produced beside the node's stream is unsafe for a layer used twice in one forward, whose two contributions autograd adds on the node's
    stream before `.grad` exists.  Not kept.)"""
    if g is None:
        if g16 is None or g_shape is None or act != ops.ACT_NONE or (need_gb and gb is None):
            raise RuntimeError("masic_amd.conv_backward: g = None needs g16, g_shape, the bias gradient and no activation")
    else:
        g = _c(g)
        g_shape = tuple(g.shape)
        if act != ops.ACT_NONE:
            g = ops.elementwise(ops.EW_ACT_BWD, g, y, s0=act)
    if x is None:
        if x16 is None or x_shape is None:
            raise RuntimeError("masic_amd.conv_backward: x = None needs x16 and x_shape")
    else:
        x_shape = tuple(x.shape)
    kh, kw, s, p = mod._geometry()
    B, Cin, Hi, Wi = x_shape
    Cout, Ho, Wo = g_shape[1], g_shape[2], g_shape[3]
    from . import nn as _mnn          # both gradients use the forward's operand precision (float32 accumulate either way)
    bf16 = _mnn._PRECISION != PREC_F32
    d = d16 = d_few = None
    dx_gemm = dx_f16k = dx_few = False
    if need_gx:
        d = ops.make_conv_desc(B, Cout, Ho, Wo, Cin, kh, kw, s, p, transposed=not mod.transposed_conv, prec=_mnn._PRECISION)
        if (d.Ho, d.Wo) != (Hi, Wi):
            raise RuntimeError("masic_amd: input-gradient geometry mismatch (odd spatial size?)")
        dx_gemm = _gemm_1x1(mod)
        if not dx_gemm and _DGRAD_F16K and bf16 and Cout % 16 == 0 and kh * kw > 1:
            # bf16 mode: the DMA-staged F16K kernel of the inference path (conv_f16k.hip) -- g converted once to the channel-blocked
            # bf16 layout, float32 NCHW out; measured 1.1 ms of a 30 ms step against the implicit-GEMM kernel
            d16 = ops.make_conv_desc(B, Cout, Ho, Wo, Cin, kh, kw, s, p, transposed=not mod.transposed_conv, in_ctot=Cout, prec=_mnn._PRECISION)
            dx_f16k = ops.conv_f16k_supported(d16)
            if not dx_f16k and _DGRAD_FEW and Cin <= 8 and s == 1 and not mod.transposed_conv and not mod.masked_conv:
                # few input channels (the 3 / 6 -> 32 input layers of Independent_EN, MASIC.py:1456-1470): dx as the transposed convolution
                # to 32 zero-padded channels on the MFMA path, only the real ones stored (masic_conv_f16k_few_fwd); the generic float32
                # kernel took 263 us per launch at 8 x 512 x 512
                d_few = ops.make_conv_desc(B, Cout, Ho, Wo, 32, kh, kw, s, p, transposed=True, in_ctot=Cout, prec=_mnn._PRECISION)
                dx_few = ops.conv_f16k_supported(d_few)
    # 3x3 stride-1 layers (Independent_EN, hyper transforms): dW from both operands in F16K, transposed LDS reads (wgrad_f16k.hip)
    dw_f16k = (need_gw and _WGRAD3_F16K and bf16 and not mod.transposed_conv and not mod.masked_conv and (kh, kw, s, p) == (3, 3, 1, 1)
               and Cin % 32 == 0 and Cout % 32 == 0 and x_shape[1] == Cin)
    # 5x5 stride-1 layers at latent resolution (encode_hyper[0], the context model): the same kernel with five kernel-row waves
    dw5_f16k = (need_gw and _WGRAD5_F16K and bf16 and not mod.transposed_conv and (kh, kw, s, p) == (5, 5, 1, 2)
                and Cin % 32 == 0 and Cout % 32 == 0 and x_shape[1] == Cin)
    if g16 is None or act != ops.ACT_NONE:
        g16 = ops.nchw_to_f16k(g) if (dx_gemm or dx_f16k or dx_few or dw_f16k or dw5_f16k) else None  # dy in F16K, converted once for both gradients

    pic_end = _PIC_END_DGRAD and bf16 and need_gx and act == ops.ACT_NONE and (kh, kw, s, p) == (5, 5, 2, 2)
    # the two picture-end layers (MASIC.py:515 g_a_conv1 = Conv2d(3 -> 128), :550 g_s_conv4 = ConvTranspose2d(128 -> 3)); without
    # these forms their input gradients run on the float32 NCHW kernels (266 / 170 us per launch at 8 x 512 x 512)
    dx_d2s = pic_end and not mod.transposed_conv and Cin <= 8 and Cout % 32 == 0 and g16 is not None
    dx_conv_a = pic_end and mod.transposed_conv and (Cin, Cout) == (128, 3) and gx_f16k

    # ... and their weight gradients: the 128-channel operand in F16K (dy of g_a_conv1, x of g_s_conv4), the 3-channel one float32 NCHW
    dw_pic = 0
    if need_gw and _PIC_WGRAD and bf16 and act == ops.ACT_NONE and (kh, kw, s, p) == (5, 5, 2, 2):
        if not mod.transposed_conv and (Cin, Cout) == (3, 128) and g16 is not None and x is not None and (Hi, Wi) == (2 * Ho, 2 * Wo):
            dw_pic = 1
        elif mod.transposed_conv and (Cin, Cout) == (128, 3) and x16 is not None and g is not None and (Ho, Wo) == (2 * Hi, 2 * Wi):
            dw_pic = 2
    if (x is None or g is None) and not dw_pic:
        raise RuntimeError("masic_amd.conv_backward: x / g given in F16K only, but this layer's weight gradient needs the NCHW tensor")

    def input_gradient():
        if dx_d2s:
            # dx of Conv2d(C <= 8 -> Cout) = ConvTranspose2d(Cout -> C) on the same tensor: the depth-to-space form of inference
            dd = ops.make_conv_desc(B, Cout, Ho, Wo, 32, 3, 3, 1, 1, prec=PREC_BF16)
            if ops.conv_f16k_supported(dd) and (Hi, Wi) == (2 * Ho, 2 * Wo):
                wc, bc = ops.deconv_s2_as_conv_weight_dev(weight.detach().contiguous())
                return ops.conv2d_f16k_d2s(g16, ops.pack_conv_f16k_weight(wc, dd), bc, dd, Cin)
        if dx_conv_a:
            # dx of ConvTranspose2d(128 -> 3) = Conv2d(3 -> 128) with the same tensor: the first-layer kernel without its GDN, F16K out
            return ops.conv_a_f16k(g, ops.pack_conv_a_weight(weight.detach().contiguous()))[0]
        if dx_gemm:
            # dx = W^T g: the same GEMM kernel on the transposed weight (packed per step: the weights change with every optimizer step)
            wt = ops.pack_gemm_f16k_weight(weight.detach().contiguous(), Cout, Cin, not mod.transposed_conv)
            return ops.gemm_f16k(g16, wt, None, B, Cout, Cin, Ho, Wo, ops.ACT_NONE, want_nchw=True)
        if dx_few:
            wpad = ops.zeros((Cout, 32, kh, kw), torch.float32, weight.device)      # ConvTranspose2d layout [in = Cout][out = 32]
            wpad[:, :Cin] = weight.detach()
            return ops.conv2d_f16k_few(g16, ops.pack_conv_f16k_weight(wpad, d_few), ops.zeros(32, torch.float32, weight.device), d_few, Cin)
        if dx_f16k:
            return ops.conv2d_f16k(g16, ops.pack_conv_f16k_weight(weight.detach(), d16, persistent=weight.is_leaf and weight.is_contiguous() and not mod.masked_conv), None, d16,
                                   want_nchw=not (gx_f16k and Cin % 16 == 0))
        g32 = ops.f16k_to_nchw_dev(g16, B, Cout, Ho, Wo) if g is None else (g if g.dtype == torch.float32 else g.float())
        return ops.conv2d(g32, ops.pack_conv_weight(weight.detach(), d), None, d)

    def parameter_gradients():
        gw = None
        gb_ = gb if act == ops.ACT_NONE else None
        if dw_pic:
            gw = (ops.pic_wgrad_f16k(g16, x, B, Ho, Wo) if dw_pic == 1 else ops.pic_wgrad_f16k(x16, g, B, Hi, Wi)).view(tuple(weight.shape))
        elif need_gw and _WGRAD1_F16K and dx_gemm and x16 is not None and g16 is not None and x_shape[1] == Cin:
            # 1x1 layers: both operands are in F16K already (the forward GEMM's input, the input-gradient GEMM's dy)
            gw = (ops.gemm_wgrad_f16k(x16, g16, B, Cin, Cout, Hi * Wi) if mod.transposed_conv
                  else ops.gemm_wgrad_f16k(g16, x16, B, Cout, Cin, Hi * Wi)).view(tuple(weight.shape))
        elif dw_f16k:
            gw = ops.conv3x3_wgrad_f16k(x16 if x16 is not None else ops.nchw_to_f16k(x), g16, B, Cin, Cout, Hi, Wi)
        elif dw5_f16k:
            gw = ops.conv5x5_wgrad_f16k(x16 if x16 is not None else ops.nchw_to_f16k(x), g16, B, Cin, Cout, Hi, Wi)
        elif need_gw:
            dw = ops.make_conv_desc(B, Cin, Hi, Wi, Cout, kh, kw, s, p, transposed=mod.transposed_conv, prec=_mnn._PRECISION)
            gw = ops.conv2d_wgrad(x, g, dw, tuple(weight.shape))        # (x and g both bf16 NCHW: the bf16-input kernel)
        if need_gb and gb_ is None:
            gb_ = ops.channel_sum(g)
        return gw, gb_ if need_gb else None

    gx = input_gradient() if need_gx else None
    return (gx,) + parameter_gradients()


def conv(mod, x, act=ops.ACT_NONE):
    if getattr(mod, "masked_conv", False):
        mod.zero_masked_taps()
    return ConvFn.apply(x, mod.weight, mod.bias, mod, act)


class GdnFn(Function):
    """Forward: the fused GDN kernel.  Backward = elementwise pieces + three CxC contractions that
    reuse the conv kernels: n = gamma^ x^2 + beta^ (1x1 conv), u = gamma^T t (1x1 transposed conv),
    d gamma^ = t (x^2)^T (1x1 weight gradient)."""

    @staticmethod
    def forward(ctx, x, beta, gamma, inverse, beta_min):
        x = _c(x)
        ctx.inverse, ctx.beta_min = inverse, beta_min
        ctx.save_for_backward(x, beta, gamma)
        from . import nn as _mnn          # bf16 mode: the three-product bf16 split (float32-level accuracy at the HBM roofline)
        return ops.gdn(x, beta.detach(), gamma.detach(), inverse=inverse, beta_min=beta_min, prec=_mnn._PRECISION)

    @staticmethod
    def backward(ctx, g):
        x, beta, gamma = ctx.saved_tensors
        return gdn_backward(x, g, beta, gamma, ctx.inverse, ctx.beta_min, ctx.needs_input_grad[0]) + (None, None)


def gdn_backward(x, g, beta, gamma, inverse, beta_min, need_gx=True):
    """(dx, d beta, d gamma) of the (inverse) GDN; x, g float32 NCHW."""
    g = _c(g)
    B, C, H, W = x.shape
    from . import nn as _mnn
    if C == 128 and _mnn._PRECISION != PREC_F32:       # bf16-operand mode: the whole backward in one kernel
        return ops.gdn_bwd_fused(x, g, beta.detach(), gamma.detach(), inverse, beta_min)
    if C <= 4 and _GDN_BWD_SMALL:                      # pre_gdn / after_gdn (GDN(3) on pictures): one pass, float32 in every mode
        return ops.gdn_bwd_small(x, g, beta.detach(), gamma.detach(), inverse, beta_min)
    b_bound = float(torch.tensor((beta_min + PEDESTAL) ** 0.5, dtype=torch.float32))
    g_bound = float(torch.tensor(PEDESTAL ** 0.5, dtype=torch.float32))
    ped = float(torch.tensor(PEDESTAL, dtype=torch.float32))
    gam = ops.elementwise(ops.EW_REPARAM, gamma.detach().contiguous(), None, g_bound, ped)
    bet = ops.elementwise(ops.EW_REPARAM, beta.detach().contiguous(), None, b_bound, ped)
    x2 = ops.elementwise(ops.EW_SQUARE, x)
    d_f = ops.make_conv_desc(B, C, H, W, C, 1, 1, 1, 0)
    w4 = gam.view(C, C, 1, 1)
    nrm = ops.conv2d(x2, ops.pack_conv_weight(w4, d_f), bet, d_f)
    s, t = ops.gdn_bwd_pre(x, nrm, g, inverse)
    d_t = ops.make_conv_desc(B, C, H, W, C, 1, 1, 1, 0, transposed=True)
    u = ops.conv2d(t, ops.pack_conv_weight(w4, d_t), None, d_t)
    gx = ops.gdn_bwd_post(x, s, u) if need_gx else None
    d_w = ops.make_conv_desc(B, C, H, W, C, 1, 1, 1, 0, prec=_mnn._PRECISION)
    g_gam = ops.conv2d_wgrad(x2, t, d_w, (C, C, 1, 1)).view(C, C)
    g_bet = ops.channel_sum(t)
    g_gamma = ops.elementwise(ops.EW_REPARAM_BWD, g_gam, gamma.detach().contiguous(), g_bound)
    g_beta = ops.elementwise(ops.EW_REPARAM_BWD, g_bet, beta.detach().contiguous(), b_bound)
    return gx, g_beta, g_gamma


class EntropyBottleneckFn(Function):
    """training-mode EntropyBottleneck.forward: (z, table[C,58], noise) -> (z + noise, likelihood)."""

    @staticmethod
    def forward(ctx, z, table, noise, medians, lik_bound):
        z = _c(z)
        z_hat, lik = ops.entropy_bottleneck(z, table.detach(), medians, training=True, noise=noise, lik_bound=lik_bound)
        ctx.lik_bound = lik_bound
        ctx.save_for_backward(z_hat, table)
        return z_hat, lik

    @staticmethod
    def backward(ctx, g_zhat, g_lik):
        z_hat, table = ctx.saved_tensors
        g_lik = _c(g_lik) if g_lik is not None else ops.zeros(z_hat.shape, z_hat.dtype, z_hat.device)
        g_zhat = _c(g_zhat) if g_zhat is not None else None
        g_z, g_t = ops.entropy_bottleneck_bwd(z_hat, table.detach(), g_lik, g_zhat, ctx.lik_bound)
        return g_z, g_t, None, None, None


class EbTableFn(Function):
    """The [C,58] parameter table of an EntropyBottleneck from its 14 per-channel tensors (ops.eb_param_table) as ONE node: the backward
    splits the table gradient with one launch into contiguous blocks of one buffer (torch.cat's own backward: 14 strided views that
    AccumulateGrad then copies one by one -- 28 small launches per bottleneck and step)."""

    @staticmethod
    def forward(ctx, n_m, n_b, *params):
        ctx.shapes = [tuple(p.shape) for p in params]
        return ops.eb_param_table(list(params[:n_m]), list(params[n_m:n_m + n_b]), list(params[n_m + n_b:]))

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(ops.eb_table_split(_c(g), ctx.shapes))


def eb_param_table(matrices, biases, factors):
    return EbTableFn.apply(len(matrices), len(biases), *matrices, *biases, *factors)


class AuxLossFn(Function):
    """EntropyBottleneck.loss(): sum |logits(quantiles) - target|, density parameters detached."""

    @staticmethod
    def forward(ctx, quantiles, table, tail_mass):
        ctx.tail_mass = tail_mass
        q = _c(quantiles)
        ctx.save_for_backward(q, table)
        return ops.entropy_bottleneck_auxloss(table, q.detach(), tail_mass)

    @staticmethod
    def backward(ctx, g):
        q, table = ctx.saved_tensors
        # the incoming gradient stays on the device (float(g) would make the host wait for the stream in the middle of every step)
        return ops.entropy_bottleneck_auxloss_bwd(table, q.detach(), 1.0, ctx.tail_mass) * g, None, None


class GmmFn(Function):
    """training-mode GaussianMixtureConditional_gf.forward: (y, noise, sigma, mu, weights).  `are_logits`: the weights are the
    head's logits and the softmax over K runs inside the kernels (HSIC's own calls); False: already-normalised weights, the
    reference's signature (entropy_models.py:836-858 called as at MASIC.py:767)."""

    @staticmethod
    def forward(ctx, y, noise, sigma, mu, logits, K, scale_bound, lik_bound, are_logits=True):
        y, sigma, mu, logits = _c(y), _c(sigma), _c(mu), _c(logits)
        y_hat, lik = ops.gmm_likelihood(y, sigma, mu, logits, K, training=True, noise=noise, weights_are_logits=are_logits,
                                        scale_bound=scale_bound, lik_bound=lik_bound)
        ctx.K, ctx.sb, ctx.lb, ctx.are_logits = K, scale_bound, lik_bound, bool(are_logits)
        ctx.save_for_backward(y_hat, sigma, mu, logits)
        return y_hat, lik

    @staticmethod
    def backward(ctx, g_yhat, g_lik):
        y_hat, sigma, mu, logits = ctx.saved_tensors
        g_lik = _c(g_lik) if g_lik is not None else ops.zeros(y_hat.shape, y_hat.dtype, y_hat.device)
        g_yhat = _c(g_yhat) if g_yhat is not None else None
        g_y, g_s, g_m, g_w = ops.gmm_likelihood_bwd(y_hat, sigma, mu, logits, g_lik, g_yhat, ctx.K, ctx.are_logits, ctx.sb, ctx.lb)
        return g_y, None, g_s, g_m, g_w, None, None, None, None


class GmmLikFn(Function):
    """The likelihood half of GmmFn: (y_hat, sigma, mu, weights) -> likelihood, y_hat = y + noise drawn by the caller (AddNoiseFn) where the
    reference draws it.  Same kernels, same gradients (d lik / d y reaches y through y_hat either way); split so that the synthesis
    transform, which needs y_hat only, does not wait for the entropy parameters (HSIC._forward_graph)."""

    @staticmethod
    def forward(ctx, y_hat, sigma, mu, logits, K, scale_bound, lik_bound, are_logits=True):
        y_hat, sigma, mu, logits = _c(y_hat), _c(sigma), _c(mu), _c(logits)
        _, lik = ops.gmm_likelihood(y_hat, sigma, mu, logits, K, training=2, weights_are_logits=are_logits, scale_bound=scale_bound, lik_bound=lik_bound)
        ctx.K, ctx.sb, ctx.lb, ctx.are_logits = K, scale_bound, lik_bound, bool(are_logits)
        ctx.save_for_backward(y_hat, sigma, mu, logits)
        return lik

    @staticmethod
    def backward(ctx, g_lik):
        y_hat, sigma, mu, logits = ctx.saved_tensors
        g_y, g_s, g_m, g_w = ops.gmm_likelihood_bwd(y_hat, sigma, mu, logits, _c(g_lik), None, ctx.K, ctx.are_logits, ctx.sb, ctx.lb)
        return g_y, g_s, g_m, g_w, None, None, None, None


class AddNoiseFn(Function):
    """_quantize(x, 'noise'): x + U(-1/2,1/2); identity gradient."""

    @staticmethod
    def forward(ctx, x, noise):
        return ops.quantize(_c(x), "noise", noise=noise)

    @staticmethod
    def backward(ctx, g):
        return g, None


class WarpFn(Function):
    """kornia warp_perspective w.r.t. the source image (h_matrix is detached by every driver)."""

    @staticmethod
    def forward(ctx, src, minv, dsize):
        src = _c(src)
        ctx.src_shape = tuple(src.shape)
        ctx.save_for_backward(minv)
        return ops.warp_perspective(src, minv, dsize)

    @staticmethod
    def backward(ctx, g):
        (minv,) = ctx.saved_tensors
        return ops.warp_perspective_bwd(_c(g), minv, ctx.src_shape), None, None


class CatFn(Function):
    """torch.cat along channels: producers' outputs copied into one buffer; backward slices the gradient."""

    @staticmethod
    def forward(ctx, *ts):
        B, _, H, W = ts[0].shape
        ctx.sizes = [t.shape[1] for t in ts]
        out = torch.empty((B, sum(ctx.sizes), H, W), dtype=ts[0].dtype, device=ts[0].device)
        off = 0
        for t in ts:
            ops.copy_view(_c(t), out, off)
            off += t.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        outs, off = [], 0
        for i, c in enumerate(ctx.sizes):
            outs.append(ops.slice_copy(g, off, c) if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(outs)


class GateFn(Function):
    """x * gates[:, c:c+1] (the mask2weights products of MASIC.py:827)."""

    @staticmethod
    def forward(ctx, x, gates, c):
        x, gates = _c(x), _c(gates)
        ctx.c = c
        ctx.save_for_backward(x, gates)
        return ops.quantize(x, "copy", gate=gates, gate_c=c)

    @staticmethod
    def backward(ctx, g):
        x, gates = ctx.saved_tensors
        gx, gg = ops.gate_bwd(_c(g), x, gates, ctx.c)
        return gx, gg, None


class AbsFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.elementwise(ops.EW_ABS, x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.elementwise(ops.EW_ABS_BWD, _c(g), x)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.elementwise(ops.EW_ADD, _c(a), _c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


_EB_FUSED_EPILOGUES = os.environ.get("MASIC_EB_FUSED_EPILOGUES", "1") != "0"      # 0: g * act'(u) and the bias sums of the block nodes as separate passes (A/B timing)


class EnhancementBlockFn(Function):
    """Enhancement_Block -- three residual blocks and the skip over all of them (reference MASIC.py:149-164) -- as ONE node of the CQE
    training step in bf16 mode, forward AND backward on F16K buffers (conv_f16k.hip, wgrad_f16k.hip, f16k_ops.hip):

      (with `tail`: Independent_EN.conv2 + the residual image, reference :1487-1488, run on the block's F16K output in the same node)
      forward   x -> F16K once; per block  t = L(c1(x)),  u = L(c2(t)) (stored by the same launch as `y_pre`),  o = u + x (+ x0 on
                the last block); out -> float32 NCHW once.  Saved: x0, t_i, u_i, o_1, o_2 in bf16 (18 B / element against 48 B of
                the node-per-layer float32 graph).
      backward  per block, last to first:  g_u = g_o * L'(u);  dW2 = wgrad(t, g_u);  g_t = dgrad_c2(g_u) * L'(t) (mask in the
                epilogue);  dW1 = wgrad(x_in, g_t);  g_x = dgrad_c1(g_t) + g_o (+ g_out on the first block: the outer skip) -- the
                adds are the residual operands of the same epilogue.  Gradients travel between layers in bf16 F16K (float32
                accumulate inside every kernel), as the forward activations of this mode do.
    """

    @staticmethod
    def forward(ctx, x, eb, tail, res, *params):
        x = _c(x)
        B, C, H, W = x.shape
        x0 = ops.nchw_to_f16k(x)
        saved, cur = [x0], x0
        for i, rb in enumerate((eb.RB1, eb.RB2, eb.RB3)):
            t = rb.conv1.run_f16k_res(cur, B, H, W, act=ops.ACT_LEAKY)
            u = ops.f16k_empty(B, C, H, W, x.device)
            cur = rb.conv2.run_f16k_res(t, B, H, W, act=ops.ACT_LEAKY, res1=cur, res2=x0 if i == 2 else None, res_ctot=C, y_pre=u)
            saved += [t, u] if i == 2 and tail is None else [t, u, cur]
        ctx.eb, ctx.tail, ctx.shape = eb, tail, (B, C, H, W)
        ctx.save_for_backward(*params, *saved)
        if tail is None:
            return ops.f16k_to_nchw_dev(cur, B, C, H, W)
        return tail.run_f16k_few(cur, B, H, W, res32=None if res is None else _c(res))      # the block's output never leaves F16K

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        tail = ctx.tail
        params, acts = ctx.saved_tensors[:12], ctx.saved_tensors[12 + (0 if tail is None else 2):]
        x0, t1, u1, o1, t2, u2, o2, t3, u3 = acts[:9]
        g = _c(g)
        g_res = g_tw = g_tb = None
        if tail is None:
            g_out = ops.nchw_to_f16k(g)
        else:
            # y = conv(o3) + res with few output channels: dy zero-padded to 32 channels (two F16K records) -- the weight-gradient
            # kernel's tile, and a channel count every conv_f16k configuration takes for the input gradient (the transposed convolution
            # on the zero-padded weight)
            Ct = tail.out_channels
            w_t = ctx.saved_tensors[12]
            g_res = g if ctx.needs_input_grad[3] else None
            dy32 = ops.zeros(B * 32 * H * W, torch.int16, g.device)
            ops.nchw_to_f16k_view(g, dy32, 32, 0)
            wpad = ops.zeros((32, C, 3, 3), torch.float32, g.device)
            wpad[:Ct] = w_t.detach()
            dt = ops.make_conv_desc(B, 32, H, W, C, 3, 3, 1, 1, transposed=True, in_ctot=32, out_ctot=C, prec=PREC_BF16)
            g_out = ops.conv2d_f16k_res(dy32, ops.pack_conv_f16k_weight(wpad, dt), None, dt)
            g_tw = ops.conv3x3_wgrad_f16k(acts[9], dy32, B, C, 32, H, W)[:Ct].contiguous()
            g_tb = ops.channel_sum(g)
        blocks = ((ctx.eb.RB1, x0, t1, u1), (ctx.eb.RB2, o1, t2, u2), (ctx.eb.RB3, o2, t3, u3))
        d16 = ops.make_conv_desc(B, C, H, W, C, 3, 3, 1, 1, transposed=True, in_ctot=C, out_ctot=C, prec=PREC_BF16)
        grads = [None] * 12
        resident = ctx.eb.RB1.conv1.resident_supported(B, H, W)

        def dgrad(g16, w, **kw):            # the transposed convolution on the same weight, epilogue operands as conv2d_f16k_res
            if resident:
                return ops.conv3x3_resident(g16, ops.pack_conv3x3_resident_weight(w.detach(), transposed=True), None, B, C, C, H, W, res_ctot=C, **kw)
            return ops.conv2d_f16k_res(g16, ops.pack_conv_f16k_weight(w.detach(), d16, persistent=w.is_leaf and w.is_contiguous()), None, d16, res_ctot=C, **kw)
        # With the resident-weight kernels (32 / 64 channels) the input-gradient launches also write what the separate passes made: the
        # bias gradient of the layer they differentiate through (channel sums of their own output) and, for the launch that ends a block,
        # the NEXT block's g_u = g_o * L'(u) as a second output -- per block one elementwise pass and two reduction passes over
        # full-resolution tensors less (masic_conv3x3_resident_ex_fwd); the values are the same bf16 values summed / masked.
        fused = resident and _EB_FUSED_EPILOGUES
        go, gu, gu_sum = g_out, None, None
        for i in (2, 1, 0):
            rb, x_in, t, u = blocks[i]
            w1, w2 = params[4 * i], params[4 * i + 2]
            if gu is None:
                gu, gu_sum = ops.f16k_act_bwd_sum(go, u, 0.01, B, C, H * W) if _EB_FUSED_EPILOGUES else (ops.f16k_act_bwd(go, u, 0.01), None)
            grads[4 * i + 2] = ops.conv3x3_wgrad_f16k(t, gu, B, C, C, H, W)
            grads[4 * i + 3] = gu_sum if gu_sum is not None else ops.f16k_channel_sum(gu, B, C, H * W)
            if fused:
                gt, _, gt_sum = dgrad(gu, w2, mask=t, mask_slope=0.01, sum_of=1)
            else:
                gt = dgrad(gu, w2, mask=t, mask_slope=0.01)
                gt_sum = ops.f16k_channel_sum(gt, B, C, H * W)
            del gu
            grads[4 * i] = ops.conv3x3_wgrad_f16k(x_in, gt, B, C, C, H, W)
            grads[4 * i + 1] = gt_sum
            if fused and i > 0:
                go, gu, gu_sum = dgrad(gt, w1, res1=go, mask2=blocks[i - 1][3], mask2_slope=0.01, sum_of=2)
            else:
                go, gu, gu_sum = dgrad(gt, w1, res1=go, res2=g_out if i == 0 else None), None, None
            del gt
        gx = ops.f16k_to_nchw_dev(go, B, C, H, W) if ctx.needs_input_grad[0] else None
        return (gx, None, None, g_res) + tuple(grads) + (() if tail is None else (g_tw, g_tb))


def enhancement_block_supported(eb, x, tail=None):
    """bf16 mode, 3x3 stride-1 blocks of C % 32 channels without a channel-changing skip, F16K kernels available for the shape
    (`tail`: a 3x3 stride-1 Conv2d to <= 16 channels run on the block's F16K output inside the same node)."""
    from . import nn as _mnn
    if _mnn._PRECISION != PREC_BF16 or _mnn._FP8 or x.dim() != 4:
        return False
    B, C, H, W = x.shape
    convs = [c for rb in (eb.RB1, eb.RB2, eb.RB3) for c in (rb.conv1, rb.conv2)]
    if tail is not None and not (tail.bias is not None and tail.in_channels == C and tail.out_channels <= 16 and tail._geometry() == (3, 3, 1, 1)
                                 and not tail.transposed_conv and tail.few_supported(B, H, W)):
        return False
    return (C % 32 == 0 and all(rb.skip is None for rb in (eb.RB1, eb.RB2, eb.RB3))
            and all(c.bias is not None and c.in_channels == C and c.out_channels == C and c._geometry() == (3, 3, 1, 1) for c in convs)
            and eb.f16k_supported(B, H, W))


def enhancement_block(eb, x, tail=None, res=None):
    """Enhancement_Block(x), or tail(Enhancement_Block(x)) + res, as one node (see EnhancementBlockFn)."""
    params = [p for rb in (eb.RB1, eb.RB2, eb.RB3) for c in (rb.conv1, rb.conv2) for p in (c.weight, c.bias)]
    if tail is not None:
        params += [tail.weight, tail.bias]
    return EnhancementBlockFn.apply(x, eb, tail, res, *params)


def cat(*ts):
    """torch.cat(dim=1) on the HIP path; differentiable when any input needs a gradient."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        return CatFn.apply(*ts)
    B, _, H, W = ts[0].shape
    out = torch.empty((B, sum(t.shape[1] for t in ts), H, W), dtype=ts[0].dtype, device=ts[0].device)
    off = 0
    for t in ts:
        ops.copy_view(_c(t), out, off)
        off += t.shape[1]
    return out


class SoftmaxKFn(Function):
    """softmax over K on the (B,K,M,H,W) view (MASIC.py:389-393; K=3, M=1 for the mask2weights gates)."""

    @staticmethod
    def forward(ctx, x, K):
        y = ops.softmax_k(_c(x), K)
        ctx.K = K
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.softmax_k_bwd(_c(g), y, ctx.K), None


class RateDistortionFn(Function):
    """loss = lmbda*255^2*(MSE(x1_hat,x1)+MSE(x2_hat,x2)) + sum_t sum(log lik_t)/(-ln2*B*H*W)
    (newtrain_codec_real.py:73-87) as one node: six device reductions forward, six elementwise kernels backward."""

    @staticmethod
    def forward(ctx, lmbda, x1, x2, x1_hat, x2_hat, *liks):
        x1, x2, x1_hat, x2_hat = _c(x1), _c(x2), _c(x1_hat), _c(x2_hat)
        liks = [_c(l) for l in liks]
        B, _, H, W = x1.shape
        ctx.cb = 1.0 / (-math.log(2) * B * H * W)
        ctx.cm = lmbda * 255 ** 2
        ctx.save_for_backward(x1, x2, x1_hat, x2_hat, *liks)
        # the pieces of the criterion's report (bpp per likelihood tensor, the two MSEs) ride along as non-differentiable outputs: the
        # criterion used to evaluate all six reductions a second time under no_grad (~30 launches per step)
        if _RD_FUSED and len(liks) <= 4 and all(t.dtype == torch.float32 for t in (x1, x2, x1_hat, x2_hat, *liks)):
            # two launches: all six reductions, then every scalar of the criterion (same summation order: the same bits)
            loss, mse1, mse2, bpp, per = ops.rd_loss(x1_hat, x1, x2_hat, x2, liks, ctx.cb, ctx.cm)
            extras = (mse1, mse2, bpp) + tuple(per)
        else:
            sums = [ops.sum_log(l) for l in liks]
            bpp = sum(sums) * ctx.cb
            mse1 = ops.sse(x1_hat, x1) / x1.numel()
            mse2 = ops.sse(x2_hat, x2) / x2.numel()
            loss = (ctx.cm * (mse1 + mse2) + bpp).float()
            extras = (mse1, mse2, bpp if torch.is_tensor(bpp) else ops.zeros((), torch.float64, x1.device)) + tuple(t * ctx.cb for t in sums)
        ctx.mark_non_differentiable(*extras)
        return (loss,) + extras

    @staticmethod
    def backward(ctx, g, *_unused):
        x1, x2, x1_hat, x2_hat, *liks = ctx.saved_tensors
        # g (a device scalar, 1 for `loss.backward()`) multiplies on the device: float(g) here would make the host wait for the whole
        # forward before it may launch the first kernel of the backward
        if _RD_FUSED and len(liks) <= 4 and g.dtype == torch.float32 and g.numel() == 1 and all(t.dtype == torch.float32 for t in (x1_hat, x2_hat, *liks)):
            g1, g2, gl = ops.rd_loss_bwd(x1_hat, x1, x2_hat, x2, liks, ctx.cm * 2.0 / x1.numel(), ctx.cb, _c(g))     # one launch
            return (None, None, None, g1, g2, *gl)
        g1 = ops.elementwise(ops.EW_DIFF_SCALE, x1_hat, x1, s0=ctx.cm * 2.0 / x1.numel()).mul_(g)
        g2 = ops.elementwise(ops.EW_DIFF_SCALE, x2_hat, x2, s0=ctx.cm * 2.0 / x2.numel()).mul_(g)
        gl = [ops.elementwise(ops.EW_RECIP_SCALE, l, None, s0=ctx.cb).mul_(g) for l in liks]
        return (None, None, None, g1, g2, *gl)


_RD_FUSED = os.environ.get("MASIC_RD_FUSED", "1") != "0"             # 0: the criterion as six reductions + torch scalar arithmetic (A/B timing)


# ------------------------------------------------------------------------------------------ fused transforms (bf16-operand training)
# In the bf16-operand mode the training step runs the SAME DMA-staged kernels as inference for the analysis / synthesis transforms
# (conv_f16k.hip: both operands by DMA, (I)GDN in the epilogue) instead of one NCHW float32 launch per convolution and per GDN: the
# kernels store, per layer, the convolution's result before the GDN and the GDN's result, both F16K bf16 -- half the bytes of the
# float32 activations the per-layer nodes keep -- and the backward converts each to float32 NCHW right before the kernel that
# needs it (weight gradient: the layer's input; GDN backward: the GDN's input).  Gradients are those of the same function evaluated
# on bf16-rounded activations; the float32 mode keeps the per-layer nodes (the parity path).
_GDN_BWD_F16K = os.environ.get("MASIC_GDN_BWD_F16K", "1") != "0"
_WGRAD_B16 = os.environ.get("MASIC_WGRAD_B16", "1") != "0" and _GDN_BWD_F16K     # 0: float32 NCHW operands for the 5x5 stride-2 weight gradients (A/B timing)


def _wgrad_b16(mod, B, hw_in):
    """True if the weight gradient of `mod` on an input of B x C x hw_in takes bf16 NCHW operands (ops.conv2d_wgrad_b16_supported)."""
    if not _WGRAD_B16:
        return False
    kh, kw, s, p = mod._geometry()
    return ops.conv2d_wgrad_b16_supported(ops.make_conv_desc(B, mod.in_channels, hw_in[0], hw_in[1], mod.out_channels, kh, kw, s, p,
                                                             transposed=mod.transposed_conv, prec=PREC_BF16))


def _gdn_backward_f16k(u16, g, shape, gdn, want_f16k=True, want_b16=False, want_nchw=True):
    """GDN backward inside the fused transforms: (dx NCHW -- float32, or bf16 with want_b16 --, dx F16K | None, channel sums of dx | None,
    d beta, d gamma); u16: the GDN's saved input (F16K), g: float32 NCHW or F16K."""
    if not _GDN_BWD_F16K:
        B, C, H, W = shape
        g32 = g if g.dtype == torch.float32 else ops.f16k_to_nchw_dev(g, B, C, H, W)
        gx, gb, gg = gdn_backward(ops.f16k_to_nchw_dev(u16, B, C, H, W), g32, gdn.beta, gdn.gamma, gdn.inverse, gdn.beta_min)
        return gx, None, None, gb, gg
    return ops.gdn_bwd_fused_ex(u16, _c(g), shape, gdn.beta.detach(), gdn.gamma.detach(), gdn.inverse, gdn.beta_min, want_nchw=want_nchw,
                                want_f16k=want_f16k, want_sum=True, want_b16=want_b16 and _WGRAD_B16)


class GmmHeadsFn(Function):
    """The entropy-parameter head of one view -- three stacks (sigma, means, mixture-weight logits) of three 1x1 layers on one input,
    reference MASIC.py:330-468 -- as ONE autograd node of the bf16-mode training step, everything between its input and its three
    outputs in F16K:
      forward   x -> F16K once (read by the three stacks); layer i of the three stacks is one grouped GEMM launch (ops.gemm_f16k_group);
                the last layer writes float32 NCHW.  Saved: x and the six intermediate activations in F16K, the three outputs.
      backward  per level, last to first: act' mask (float32 elementwise on the outputs' gradients, F16K masks below), bias gradient
                (channel sums), weight gradient from the two F16K operands (masic_gemm_wgrad_f16k), input gradients of the three stacks as
                one grouped GEMM on the transposed weights; the three stacks' input gradients are summed at the end.
    Against one ConvFn per layer: 4 forward launches instead of 36, no float32 NCHW round trip between layers, 3 conversions of a
    gradient instead of 18.  Gradients travel between the layers in bf16, as the operands of the per-layer GEMMs do anyway."""

    @staticmethod
    def forward(ctx, x, head, *params):
        x = _c(x)
        B, _, H, W = x.shape
        x16 = ops.nchw_to_f16k(x)
        # all 18 weight packs of the node -- every layer for its forward GEMM and, transposed, for its input-gradient GEMM -- in one launch
        mods = [getattr(head, name)[2 * i] for name, _, _ in head._STACKS for i in range(3)]
        wds = [params[2 * n].detach().contiguous() for n in range(9)]
        jobs = ([(w, m.in_channels, m.out_channels, m.transposed_conv) for w, m in zip(wds, mods)] +
                [(w, m.out_channels, m.in_channels, not m.transposed_conv) for w, m in zip(wds, mods)])
        packs = ops.pack_gemm_f16k_weights(jobs) if _HEADS_MULTIPACK else [ops.pack_gemm_f16k_weight(*j) for j in jobs]
        ctx.wt_packs = packs[9:]
        t, inter = [x16, x16, x16], []
        for i in range(3):
            layers = []
            for k, (name, _, acts) in enumerate(head._STACKS):
                layer = getattr(head, name)[2 * i]
                layers.append(dict(x=t[k], wp=packs[3 * k + i], bias=None if layer.bias is None else layer.bias.detach(),
                                   Cin=layer.in_channels, Cout=layer.out_channels, act=acts[i], out="nchw" if i == 2 else "f16k"))
            t = ops.gemm_f16k_group(layers, B, H, W)
            if i < 2:
                inter += t
        ctx.head, ctx.shape = head, (B, H, W)
        ctx.save_for_backward(*params, x16, *inter, *t)
        return tuple(t)

    @staticmethod
    def backward(ctx, *gouts):
        head = ctx.head
        B, H, W = ctx.shape
        HW = H * W
        params, rest = ctx.saved_tensors[:18], ctx.saved_tensors[18:]
        x16, t0, t1, ys = rest[0], rest[1:4], rest[4:7], rest[7:10]
        stacks = head._STACKS
        mods = [[getattr(head, name)[2 * i] for i in range(3)] for name, _, _ in stacks]
        slope = lambda act: 0.01 if act == ops.ACT_LEAKY else 0.0
        grads = [None] * 18

        def wgrad(mod, a16, g16, with_bias=False):  # dW in the module's weight layout (+ the bias gradient: channel sums of dy, same launch)
            Cin, Cout = mod.in_channels, mod.out_channels
            dw = (ops.gemm_wgrad_f16k(a16, g16, B, Cin, Cout, HW, bias_of=2 if with_bias else 0) if mod.transposed_conv
                  else ops.gemm_wgrad_f16k(g16, a16, B, Cout, Cin, HW, bias_of=1 if with_bias else 0))
            if with_bias:
                return dw[0].view(tuple(mod.weight.shape)), dw[1]
            return dw.view(tuple(mod.weight.shape))

        def dgrad(level, g16s, out):               # the three stacks' input gradients of one level: one grouped GEMM on W^T
            layers = []
            for k in range(3):
                mod = mods[k][level]
                layers.append(dict(x=g16s[k], wp=ctx.wt_packs[3 * k + level], bias=None, Cin=mod.out_channels, Cout=mod.in_channels, act=ops.ACT_NONE, out=out))
            return ops.gemm_f16k_group(layers, B, H, W)

        # level 2: the outputs' gradients arrive in float32 NCHW
        g16 = []
        for k, (_, _, acts) in enumerate(stacks):
            g = gouts[k]
            g = ops.zeros(ys[k].shape, torch.float32, ys[k].device) if g is None else _c(g)
            if acts[2] != ops.ACT_NONE:
                g = ops.elementwise(ops.EW_ACT_BWD, g, ys[k], s0=acts[2])
            g16.append(ops.nchw_to_f16k(g))
            if mods[k][2].bias is not None and _WGRAD1_BIAS:     # (the bias gradient: sums of dy as the weight gradient sees it, bf16-rounded)
                grads[6 * k + 4], grads[6 * k + 5] = wgrad(mods[k][2], t1[k], g16[k], with_bias=True)
                continue
            if mods[k][2].bias is not None:
                grads[6 * k + 5] = ops.channel_sum(g)
            grads[6 * k + 4] = wgrad(mods[k][2], t1[k], g16[k])
        for level, saved in ((1, t1), (0, t0)):
            gin = dgrad(level + 1, g16, "f16k")
            g16 = []
            for k, (_, _, acts) in enumerate(stacks):
                gm = gin[k] if acts[level] == ops.ACT_NONE else ops.f16k_act_bwd(gin[k], saved[k], slope(acts[level]))
                g16.append(gm)
                mod = mods[k][level]
                if mod.bias is not None and _WGRAD1_BIAS:
                    grads[6 * k + 2 * level], grads[6 * k + 2 * level + 1] = wgrad(mod, x16 if level == 0 else t0[k], gm, with_bias=True)
                    continue
                if mod.bias is not None:
                    grads[6 * k + 2 * level + 1] = ops.f16k_channel_sum(gm, B, mod.out_channels, HW)
                grads[6 * k + 2 * level] = wgrad(mod, x16 if level == 0 else t0[k], gm)
        gx = None
        if ctx.needs_input_grad[0]:
            parts = dgrad(0, g16, "nchw")
            gx = parts[0].add_(parts[1]).add_(parts[2])
        return (gx, None) + tuple(grads)


def gmm_heads_supported(head, x):
    """bf16 mode (not fp8), every layer of the three stacks a biased 1x1 layer on the DMA-staged GEMM (channels multiples of 32)."""
    from . import nn as _mnn
    if _mnn._PRECISION != PREC_BF16 or _mnn._FP8 or not _GMM_HEADS_FN or x.dim() != 4:
        return False
    for name, _, _ in head._STACKS:
        seq = getattr(head, name)
        for i in (0, 2, 4):
            m = seq[i]
            if not (_gemm_1x1(m) and m.bias is not None):
                return False
    return True


def gmm_heads(head, x):
    params = []
    for name, _, _ in head._STACKS:
        seq = getattr(head, name)
        for i in (0, 2, 4):
            params += [seq[i].weight, seq[i].bias]
    return GmmHeadsFn.apply(x, head, *params)


_WGRAD1_BIAS = os.environ.get("MASIC_WGRAD1_BIAS", "1") != "0"          # 0: bias gradients of the head layers by a reduction pass of their own (A/B timing)
_HEADS_MULTIPACK = os.environ.get("MASIC_HEADS_MULTIPACK", "1") != "0"     # 0: one pack launch per weight and orientation (A/B timing)
_GMM_HEADS_FN = os.environ.get("MASIC_GMM_HEADS_FN", "1") != "0"     # 0: one ConvFn per head layer (A/B timing)


class AnalysisFn(Function):
    """conv(3->128)+GDN, conv+GDN, conv+GDN, conv(128->M) of Encoder1 / Encoder2 (reference MASIC.py:510-531): x float32 NCHW -> y float32 NCHW."""

    @staticmethod
    def forward(ctx, x, enc, *params):
        from . import nn as _mnn
        x = _c(x)
        convs = (enc.g_a_conv1, enc.g_a_conv2, enc.g_a_conv3, enc.g_a_conv4)
        gdns = (enc.g_a_gdn1, enc.g_a_gdn2, enc.g_a_gdn3)
        B, _, H, W = x.shape
        c1 = convs[0]
        u1, a1, h1, w1 = ops.conv_a_gdn_dual(x, c1.packed_first_layer_weight(), None if c1.bias is None else c1.bias.detach(),
                                             (_mnn.packed_gdn_f16k(gdns[0]), gdns[0].inverse))
        u2, a2, h2, w2 = convs[1].run_f16k_dual(a1, B, h1, w1, gdns[1])
        u3, a3, h3, w3 = convs[2].run_f16k_dual(a2, B, h2, w2, gdns[2])
        y = convs[3].run_f16k(a3, B, h3, w3, want_nchw=True)[0]
        ctx.enc, ctx.sizes = enc, ((h1, w1), (h2, w2), (h3, w3))
        ctx.save_for_backward(x, u1, a1, u2, a2, u3, a3, *params)
        return y

    @staticmethod
    def backward(ctx, g):
        x, u1, a1, u2, a2, u3, a3, *params = ctx.saved_tensors
        enc = ctx.enc
        convs = (enc.g_a_conv1, enc.g_a_conv2, enc.g_a_conv3, enc.g_a_conv4)
        gdns = (enc.g_a_gdn1, enc.g_a_gdn2, enc.g_a_gdn3)
        B = x.shape[0]
        nchw = lambda t16, hw, bf16=False: ops.f16k_to_nchw_dev(t16, B, 128, hw[0], hw[1], bf16=bf16)
        grads = {}
        # The GDN backward reads its saved input (and the gradient, when an F16K kernel produced it) as F16K and writes dx twice --
        # float32 NCHW for the weight-gradient kernel, F16K for the input-gradient convolution -- plus dx's channel sums (the bias
        # gradient): no F16K -> NCHW pass on u, no NCHW -> F16K pass and no reduction pass on dx.  MASIC_GDN_BWD_F16K=0: the separate passes.
        gx, grads["w4"], grads["b4"] = conv_backward(convs[3], nchw(a3, ctx.sizes[2]), convs[3].weight, None, g, ops.ACT_NONE, gx_f16k=_GDN_BWD_F16K)
        for i, (u, a_in, hw, hw_in) in ((2, (u3, a2, ctx.sizes[2], ctx.sizes[1])), (1, (u2, a1, ctx.sizes[1], ctx.sizes[0]))):
            b16 = _wgrad_b16(convs[i], B, hw_in)       # the weight gradient's two operands as bf16 NCHW: half the bytes on every side of it
            gu, gu16, gsum, grads["beta%d" % (i + 1)], grads["gamma%d" % (i + 1)] = _gdn_backward_f16k(u, gx, (B, 128) + tuple(hw), gdns[i], want_b16=b16)
            gx, grads["w%d" % (i + 1)], grads["b%d" % (i + 1)] = conv_backward(convs[i], nchw(a_in, hw_in, b16), convs[i].weight, None, gu, ops.ACT_NONE,
                                                                                g16=gu16, gb=gsum, gx_f16k=_GDN_BWD_F16K)
        # the first layer takes dy in F16K for both gradients (ops.pic_wgrad_f16k, the depth-to-space input gradient): no float32 NCHW dx here
        f16k_only = _GDN_BWD_F16K and _PIC_WGRAD and _PIC_END_DGRAD and tuple(x.shape[2:]) == (2 * ctx.sizes[0][0], 2 * ctx.sizes[0][1])
        gu, gu16, gsum, grads["beta1"], grads["gamma1"] = _gdn_backward_f16k(u1, gx, (B, 128) + tuple(ctx.sizes[0]), gdns[0],
                                                                             want_f16k=ctx.needs_input_grad[0] or f16k_only, want_nchw=not f16k_only)
        gimg, grads["w1"], grads["b1"] = conv_backward(convs[0], x, convs[0].weight, None, gu, ops.ACT_NONE, need_gx=ctx.needs_input_grad[0], g16=gu16, gb=gsum,
                                                       g_shape=(B, 128) + tuple(ctx.sizes[0]))
        return (gimg, None, grads["w1"], grads["b1"], grads["beta1"], grads["gamma1"], grads["w2"], grads["b2"], grads["beta2"], grads["gamma2"],
                grads["w3"], grads["b3"], grads["beta3"], grads["gamma3"], grads["w4"], grads["b4"])


def analysis(enc, x):
    c, g = (enc.g_a_conv1, enc.g_a_conv2, enc.g_a_conv3, enc.g_a_conv4), (enc.g_a_gdn1, enc.g_a_gdn2, enc.g_a_gdn3)
    return AnalysisFn.apply(x, enc, c[0].weight, c[0].bias, g[0].beta, g[0].gamma, c[1].weight, c[1].bias, g[1].beta, g[1].gamma,
                            c[2].weight, c[2].bias, g[2].beta, g[2].gamma, c[3].weight, c[3].bias)


def analysis_supported(enc, x):
    """bf16-operand mode, the reference's layer shapes, every layer with an F16K configuration at this size."""
    from . import nn as _mnn
    if _mnn._PRECISION == PREC_F32 or _mnn._FP8:
        return False
    B, C, H, W = x.shape
    c = (enc.g_a_conv1, enc.g_a_conv2, enc.g_a_conv3, enc.g_a_conv4)
    if (C, c[0].out_channels, tuple(c[0].kernel_size), tuple(c[0].stride), tuple(c[0].padding)) != (3, 128, (5, 5), (2, 2), (2, 2)):
        return False
    if c[1].out_channels != 128 or c[2].out_channels != 128 or any(m.bias is None for m in c):
        return False
    h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    for m in c[1:]:
        if not m.f16k_supported(B, h, w):
            return False
        d = m._desc_f16k(B, h, w)
        h, w = d.Ho, d.Wo
    return True


class SynthesisFn(Function):
    """deconv+IGDN x3, deconv(128->3) of Decoder1 / Decoder2 (reference MASIC.py:533-554): y_hat float32 NCHW -> float32 NCHW."""

    @staticmethod
    def forward(ctx, y_hat, dec, *params):
        y_hat = _c(y_hat)
        convs = (dec.g_s_conv1, dec.g_s_conv2, dec.g_s_conv3, dec.g_s_conv4)
        gdns = (dec.g_s_gdn1, dec.g_s_gdn2, dec.g_s_gdn3)
        B, _, H, W = y_hat.shape
        t16 = ops.nchw_to_f16k(y_hat)
        saved, sizes, hw = [], [], (H, W)
        for i in range(3):
            u, t16, ho, wo = convs[i].run_f16k_dual(t16, B, hw[0], hw[1], gdns[i])
            hw = (ho, wo)
            saved += [u, t16]
            sizes.append(hw)
        x_hat = convs[3].run_f16k_d2s(t16, B, hw[0], hw[1])
        ctx.dec, ctx.sizes = dec, sizes
        ctx.save_for_backward(y_hat, *saved, *params)
        return x_hat

    @staticmethod
    def backward(ctx, g):
        y_hat, u1, a1, u2, a2, u3, a3, *params = ctx.saved_tensors
        dec = ctx.dec
        convs = (dec.g_s_conv1, dec.g_s_conv2, dec.g_s_conv3, dec.g_s_conv4)
        gdns = (dec.g_s_gdn1, dec.g_s_gdn2, dec.g_s_gdn3)
        B = y_hat.shape[0]
        nchw = lambda t16, hw, bf16=False: ops.f16k_to_nchw_dev(t16, B, 128, hw[0], hw[1], bf16=bf16)
        grads = {}
        # the last layer's weight gradient reads its saved input where it is, in F16K (ops.pic_wgrad_f16k): no float32 NCHW copy of a3
        hw3 = tuple(ctx.sizes[2])
        f16k_only = (_GDN_BWD_F16K and _PIC_WGRAD and _PIC_END_DGRAD and tuple(convs[3].weight.shape[:2]) == (128, 3)
                     and tuple(g.shape[2:]) == (2 * hw3[0], 2 * hw3[1]))
        gx, grads["w4"], grads["b4"] = conv_backward(convs[3], None if f16k_only else nchw(a3, hw3), convs[3].weight, None, g, ops.ACT_NONE,
                                                     gx_f16k=_GDN_BWD_F16K, x16=a3, x_shape=(B, 128) + hw3)
        for i, (u, a_in, hw, hw_in) in ((2, (u3, a2, ctx.sizes[2], ctx.sizes[1])), (1, (u2, a1, ctx.sizes[1], ctx.sizes[0]))):
            b16 = _wgrad_b16(convs[i], B, hw_in)       # the weight gradient's two operands as bf16 NCHW: half the bytes on every side of it
            gu, gu16, gsum, grads["beta%d" % (i + 1)], grads["gamma%d" % (i + 1)] = _gdn_backward_f16k(u, gx, (B, 128) + tuple(hw), gdns[i], want_b16=b16)
            gx, grads["w%d" % (i + 1)], grads["b%d" % (i + 1)] = conv_backward(convs[i], nchw(a_in, hw_in, b16), convs[i].weight, None, gu, ops.ACT_NONE,
                                                                                g16=gu16, gb=gsum, gx_f16k=_GDN_BWD_F16K)
        gu, gu16, gsum, grads["beta1"], grads["gamma1"] = _gdn_backward_f16k(u1, gx, (B, 128) + tuple(ctx.sizes[0]), gdns[0])
        gy, grads["w1"], grads["b1"] = conv_backward(convs[0], y_hat, convs[0].weight, None, gu, ops.ACT_NONE, need_gx=ctx.needs_input_grad[0],
                                                     g16=gu16, gb=gsum)
        return (gy, None, grads["w1"], grads["b1"], grads["beta1"], grads["gamma1"], grads["w2"], grads["b2"], grads["beta2"], grads["gamma2"],
                grads["w3"], grads["b3"], grads["beta3"], grads["gamma3"], grads["w4"], grads["b4"])


def synthesis(dec, y_hat):
    c, g = (dec.g_s_conv1, dec.g_s_conv2, dec.g_s_conv3, dec.g_s_conv4), (dec.g_s_gdn1, dec.g_s_gdn2, dec.g_s_gdn3)
    return SynthesisFn.apply(y_hat, dec, c[0].weight, c[0].bias, g[0].beta, g[0].gamma, c[1].weight, c[1].bias, g[1].beta, g[1].gamma,
                             c[2].weight, c[2].bias, g[2].beta, g[2].gamma, c[3].weight, c[3].bias)


def synthesis_supported(dec, y_hat):
    from . import nn as _mnn
    if _mnn._PRECISION == PREC_F32 or _mnn._FP8:
        return False
    B, _, H, W = y_hat.shape
    c = (dec.g_s_conv1, dec.g_s_conv2, dec.g_s_conv3)
    hw = (H, W)
    for m in c:
        if m.out_channels != 128 or m.bias is None or not m.f16k_supported(B, *hw):
            return False
        d = m._desc_f16k(B, *hw)
        hw = (d.Ho, d.Wo)
    return dec.g_s_conv4.bias is not None and dec.g_s_conv4.d2s_supported(B, *hw)
