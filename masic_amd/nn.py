"""HIP-backed drop-ins for the nn.Conv2d / nn.ConvTranspose2d modules the reference's factories
return (compressai/models/utils.py:128-146).  They subclass the torch modules so parameter names,
initialisation (`isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))`, MASIC.py:67-72), state dicts and
optimizers are unchanged; only `forward` is replaced by the implicit-GEMM kernels of
masic_amd/csrc/conv.hip.

Weights are re-laid out ("packed") for the kernel once per weight version: the cache key is the
parameter's autograd version counter + storage pointer, so optimizer steps and load_state_dict
trigger a re-pack and eval loops do not.
"""
import torch
import torch.nn as nn

import os

from . import ops
from ._lib import PREC_BF16, PREC_F32, PREC_FP8

_PRECISION = {"f32": PREC_F32, "bf16": PREC_BF16, "fp8": PREC_BF16}[os.environ.get("MASIC_PRECISION", "f32")]
_FP8 = os.environ.get("MASIC_PRECISION", "f32") == "fp8"
_C3_RESIDENT = os.environ.get("MASIC_C3_RESIDENT", "1") != "0"


def set_precision(name):
    """Operand precision of the forward MFMA contractions: "f32" (parity path, exact float32 MFMA), "bf16" (bf16 operands,
    float32 accumulate; BASELINE's headline dtype) or "fp8" (BASELINE configs[4]: the bf16 path with OCP e4m3 operands on the
    128 -> 128 5x5 layers of the analysis / synthesis transforms and the first two layers of the entropy-parameter stacks of
    modules calibrated with masic_amd.fp8.calibrate; inference only).  Backward contractions are float32 or bf16."""
    global _PRECISION, _FP8
    _PRECISION = {"f32": PREC_F32, "bf16": PREC_BF16, "fp8": PREC_BF16}[name]
    _FP8 = name == "fp8"


def get_precision():
    return "fp8" if _FP8 else ("bf16" if _PRECISION == PREC_BF16 else "f32")


def reduced_precision():
    """bf16 or fp8 operands (the F16K chains apply)."""
    return _PRECISION == PREC_BF16


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


from .fresh import invalidate_packs, set_pack_verify, stamp, weight_key  # noqa: E402,F401  (freshness keys of everything derived from a parameter)


def _cached(obj, slot, version_key, shape_key, build):
    """Per-module pack cache: one dict per parameter version, one entry per (shape, precision) it was packed for.  A call at
    another shape adds an entry instead of evicting the previous one -- a captured HIP graph (masic_amd/graph.py) keeps
    replaying against the pack it was captured with; a new parameter version (optimizer step, load_state_dict) drops them all."""
    d = obj.__dict__.get(slot)
    if d is None or d[0] != version_key:
        d = (version_key, {})
        obj.__dict__[slot] = d
    t = d[1].get(shape_key)
    if t is None:
        t = d[1][shape_key] = build()
    return t


def pack_sources(module):
    """The tensors every cached pack / table of `module` is derived from: its parameters (all of them -- HSIC.parameters()
    hides the entropy bottlenecks) and the context models' mask buffers."""
    return [t for _, t in module.named_parameters()] + [t for n, t in module.named_buffers() if n.endswith(".mask")]


def pack_signature(tensors):
    """(version counter, storage pointer) of each source tensor.  GraphedHSIC compares it before each replay (~50 us)."""
    return tuple([(stamp(t), t.data_ptr()) for t in tensors])


def cached_packs(module):
    """Every pack / table tensor currently cached under `module` (held by GraphedHSIC so that they outlive the caches)."""
    out = []
    for m in module.modules():
        for k, v in m.__dict__.items():
            if k.endswith("_cache") and isinstance(v, tuple):
                stack = list(v[1:])
                while stack:
                    x = stack.pop()
                    if isinstance(x, torch.Tensor):
                        out.append(x)
                    elif isinstance(x, dict):
                        stack.extend(x.values())
                    elif isinstance(x, (tuple, list)):
                        stack.extend(x)
    return out


def packed_gdn_f16k(gdn):
    """Fragment-order parameters of a GDN module for conv_f16k's fused epilogue, cached per parameter version."""
    key = weight_key(gdn.beta) + weight_key(gdn.gamma)
    cache = gdn.__dict__.get("_packed_f16k_cache")
    if cache is None or cache[0] != key:
        cache = (key, ops.pack_gdn_f16k(gdn.beta.detach(), gdn.gamma.detach().contiguous(), gdn.beta_min))
        gdn.__dict__["_packed_f16k_cache"] = cache
    return cache[1]


class _PackedWeightMixin:
    transposed_conv = False
    masked_conv = False

    def _geometry(self):
        kh, kw = _pair(self.kernel_size)
        sh, sw = _pair(self.stride)
        ph, pw = _pair(self.padding)
        if sh != sw or ph != pw or _pair(self.dilation) != (1, 1) or self.groups != 1:
            raise RuntimeError("masic_amd: only square stride/padding, dilation 1, groups 1 convolutions are built")
        if self.transposed_conv and _pair(self.output_padding) != (sh - 1, sh - 1):
            raise RuntimeError("masic_amd: ConvTranspose2d needs output_padding = stride - 1 (compressai deconv())")
        return kh, kw, sh, ph

    def _desc(self, x_shape, in_ctot=None, in_coff=0, out_ctot=None, out_coff=0, in_op=ops.INOP_NONE,
              act=ops.ACT_NONE, gate_ctot=0, gate_c=0):
        kh, kw, s, p = self._geometry()
        B, _, Hi, Wi = x_shape
        return ops.make_conv_desc(B, self.in_channels, Hi, Wi, self.out_channels, kh, kw, s, p,
                                  transposed=self.transposed_conv, masked=self.masked_conv,
                                  in_ctot=in_ctot, in_coff=in_coff, out_ctot=out_ctot, out_coff=out_coff,
                                  in_op=in_op, act=act, gate_ctot=gate_ctot, gate_c=gate_c, prec=_PRECISION)

    def packed_weight(self, desc):
        w = self.weight
        return _cached(self, "_packed_cache", weight_key(w), (desc.B, desc.Hi, desc.Wi, desc.prec),
                       lambda: ops.pack_conv_weight(w.detach().contiguous(), desc))

    def packed_gemm_weight(self):
        """[ci/16][co][16] bf16 pack of a 1x1 layer for the register-streamed GEMM (masic_amd/csrc/gemm_bf16.hip)."""
        w = self.weight
        key = weight_key(w)
        cache = self.__dict__.get("_packed_gemm_cache")
        if cache is None or cache[0] != key:
            cache = (key, ops.pack_gemm1x1_weight(w.detach().contiguous(), self.in_channels, self.out_channels, self.transposed_conv))
            self.__dict__["_packed_gemm_cache"] = cache
        return cache[1]

    # ---- F16K path (bf16 operands; activations [B][C/16][H*W][16] bf16 between layers, masic_amd/csrc/conv_f16k.hip)
    def _desc_f16k(self, B, Hi, Wi, out_ctot=None, out_coff=0, act=ops.ACT_NONE, gate_ctot=0, gate_c=0, in_ctot=None, in_coff=0):
        kh, kw, s, p = self._geometry()
        cin16 = (self.in_channels + 15) // 16 * 16
        return ops.make_conv_desc(B, self.in_channels, Hi, Wi, self.out_channels, kh, kw, s, p,
                                  transposed=self.transposed_conv, masked=self.masked_conv,
                                  in_ctot=cin16 if in_ctot is None else in_ctot, in_coff=in_coff,
                                  out_ctot=out_ctot, out_coff=out_coff, act=act, gate_ctot=gate_ctot, gate_c=gate_c, prec=PREC_BF16)

    def f16k_supported(self, B, Hi, Wi):
        return ops.conv_f16k_supported(self._desc_f16k(B, Hi, Wi))

    def packed_f16k_weight(self, desc):
        w = self.weight
        return _cached(self, "_packed_f16k_cache", weight_key(w), (desc.B, desc.Hi, desc.Wi),
                       lambda: ops.pack_conv_f16k_weight(w.detach().contiguous(), desc, persistent=w.is_contiguous() and not self.masked_conv))     # (a masked layer zeroes taps in place right before its pack)

    # ---- fp8 operands (masic_amd/csrc/conv_f16k.hip: conv_f16k<..., F8>; activations F8K [B][C/32][H*W][32] fp8)
    def _desc_f8k(self, B, Hi, Wi, out_ctot=None, out_coff=0, act=ops.ACT_NONE, gate_ctot=0, gate_c=0):
        kh, kw, s, p = self._geometry()
        cin32 = (self.in_channels + 31) // 32 * 32
        return ops.make_conv_desc(B, self.in_channels, Hi, Wi, self.out_channels, kh, kw, s, p,
                                  transposed=self.transposed_conv, masked=self.masked_conv, in_ctot=cin32,
                                  out_ctot=out_ctot, out_coff=out_coff, act=act, gate_ctot=gate_ctot, gate_c=gate_c, prec=PREC_FP8)

    def f8k_supported(self, B, Hi, Wi):
        return self.in_channels % 32 == 0 and ops.conv_f16k_supported(self._desc_f8k(B, Hi, Wi))

    def _out_desc(self, make, B, Hi, Wi, act, out, out_nchw, out_coff, gate, gate_c):
        if out_nchw is not None:
            return make(B, Hi, Wi, out_ctot=out_nchw.shape[1], out_coff=out_coff, act=act, gate_ctot=0 if gate is None else gate.shape[1], gate_c=gate_c)
        if out == "nchw":
            return make(B, Hi, Wi, act=act)
        blk = 32 if out == "f8k" else 16
        return make(B, Hi, Wi, out_ctot=(self.out_channels + blk - 1) // blk * blk, act=act)

    def run_f8k(self, x8, in_scale, B, Hi, Wi, act=ops.ACT_NONE, out="f16k", out_scale=None, out_nchw=None, out_coff=0, gate=None, gate_c=0, gdn=None):
        """Inference-only: the convolution with fp8 operands on an F8K input holding fp8(x / in_scale).  Returns (y, Ho, Wo);
        y: F16K bf16 ("f16k"), F8K fp8(y / out_scale) ("f8k") or float32 NCHW ("nchw" / a view of `out_nchw`)."""
        desc = self._out_desc(self._desc_f8k, B, Hi, Wi, act, out, out_nchw, out_coff, gate, gate_c)
        w = self.weight
        vkey = weight_key(w)
        wp, ws = _cached(self, "_packed_f8k_cache", vkey, (desc.B, desc.Hi, desc.Wi), lambda: ops.pack_conv_f8k_weight(w.detach().contiguous(), desc))
        wscale = _cached(self, "_wscale_f8k_cache", vkey, (desc.B, desc.Hi, desc.Wi, float(in_scale)), lambda: (ws * float(in_scale)).contiguous())
        y = ops.conv2d_f8k(x8, wp, wscale, None if self.bias is None else self.bias.detach(), desc, out=out, out_scale=out_scale, out_nchw=out_nchw,
                           gate=gate, gdn=None if gdn is None else (packed_gdn_f16k(gdn), gdn.inverse))
        return y, desc.Ho, desc.Wo

    def run_f16k_f8out(self, x16, B, Hi, Wi, out_scale, act=ops.ACT_NONE, gdn=None):
        """bf16 operands on an F16K input, result stored as F8K fp8(y / out_scale) for an fp8-operand consumer."""
        desc = self._desc_f16k(B, Hi, Wi, out_ctot=(self.out_channels + 31) // 32 * 32, act=act)
        y = ops.conv2d_f8k(x16, self.packed_f16k_weight(desc), None, None if self.bias is None else self.bias.detach(), desc, out="f8k",
                           out_scale=out_scale, gdn=None if gdn is None else (packed_gdn_f16k(gdn), gdn.inverse))
        return y, desc.Ho, desc.Wo

    def packed_gemm_f8k_weight(self, in_scale):
        """(fp8 pack, dequantisation scales = per-channel weight scale x in_scale) of a 1x1 layer for gemm_f8k."""
        w = self.weight
        vkey = weight_key(w)
        wp, ws = _cached(self, "_packed_gemm_f8k_cache", vkey, (), lambda: ops.pack_gemm_f8k_weight(w.detach().contiguous(), self.in_channels, self.out_channels, self.transposed_conv))
        return wp, _cached(self, "_wscale_gemm_f8k_cache", vkey, (float(in_scale),), lambda: (ws * float(in_scale)).contiguous())

    def packed_first_layer_weight(self):
        """Fragment image of a Conv2d(3, 128, 5, stride 2) weight for the fused conv + GDN kernel of the first analysis layer."""
        w = self.weight
        key = weight_key(w)
        cache = self.__dict__.get("_packed_conv_a_cache")
        if cache is None or cache[0] != key:
            cache = (key, ops.pack_conv_a_weight(w.detach().contiguous()))
            self.__dict__["_packed_conv_a_cache"] = cache
        return cache[1]

    def run_f16k_d2s(self, x16, B, Hi, Wi, out=None, out_coff=0):
        """Inference-only, ConvTranspose2d(Cin -> C <= 8, k5, s2) on an F16K input: run as the equivalent stride-1 3x3 convolution
        to 4C channels with a depth-to-space store (masic_conv_f16k_d2s_fwd). Returns float32 NCHW [B, C, 2Hi, 2Wi]."""
        if not self.transposed_conv or self._geometry() != (5, 5, 2, 2):
            raise RuntimeError("masic_amd: run_f16k_d2s is for ConvTranspose2d(k=5, s=2, p=2)")
        desc = ops.make_conv_desc(B, self.in_channels, Hi, Wi, 32, 3, 3, 1, 1, prec=PREC_BF16)
        w = self.weight

        def build():
            wc, bc = ops.deconv_s2_as_conv_weight_dev(w.detach().contiguous(), None if self.bias is None else self.bias.detach())
            return ops.pack_conv_f16k_weight(wc, desc), bc
        wp, bc = _cached(self, "_packed_d2s_cache", weight_key(w) + (None if self.bias is None else weight_key(self.bias),),
                         (B, Hi, Wi), build)
        return ops.conv2d_f16k_d2s(x16, wp, bc, desc, self.out_channels, out=out, out_coff=out_coff)

    def d2s_supported(self, B, Hi, Wi):
        return (self.transposed_conv and self._geometry() == (5, 5, 2, 2) and self.out_channels <= 8 and self.in_channels % 32 == 0
                and ops.conv_f16k_supported(ops.make_conv_desc(B, self.in_channels, Hi, Wi, 32, 3, 3, 1, 1, prec=PREC_BF16)))

    def run_f16k(self, x16, B, Hi, Wi, act=ops.ACT_NONE, want_nchw=False, out=None, out_coff=0, gate=None, gate_c=0, gdn=None, out16=None, out16_ctot=0):
        """Inference-only: y = act(conv(x) + bias) on an F16K input buffer. Returns (y, Ho, Wo) with y an F16K buffer of
        ceil16(Cout) channels, or float32 NCHW when `want_nchw` / `out` (channel view of a concat buffer, optional gate).
        `gdn`: a 128-channel compressai GDN module applied to the result inside the kernel's epilogue."""
        if out16 is not None:             # channels out_coff.. of an F16K concat buffer of out16_ctot channels (optionally gated)
            desc = self._desc_f16k(B, Hi, Wi, out_ctot=out16_ctot, out_coff=out_coff, act=act,
                                   gate_ctot=0 if gate is None else gate.shape[1], gate_c=gate_c)
        elif out is not None:
            desc = self._desc_f16k(B, Hi, Wi, out_ctot=out.shape[1], out_coff=out_coff, act=act,
                                   gate_ctot=0 if gate is None else gate.shape[1], gate_c=gate_c)
        elif want_nchw:
            desc = self._desc_f16k(B, Hi, Wi, act=act)
        else:
            desc = self._desc_f16k(B, Hi, Wi, out_ctot=(self.out_channels + 15) // 16 * 16, act=act)
        bias = None if self.bias is None else self.bias.detach()
        y = ops.conv2d_f16k(x16, self.packed_f16k_weight(desc), bias, desc, out_nchw=out, want_nchw=want_nchw, gate=gate,
                            gdn=None if gdn is None else (packed_gdn_f16k(gdn), gdn.inverse), out16=out16)
        return y, desc.Ho, desc.Wo

    def run_f16k_dual(self, x16, B, Hi, Wi, gdn, products=None):
        """Training-mode forward of conv + (I)GDN on F16K: returns (pre-GDN F16K, post-GDN F16K, Ho, Wo)."""
        desc = self._desc_f16k(B, Hi, Wi, out_ctot=(self.out_channels + 15) // 16 * 16)
        pre, y = ops.conv2d_f16k_gdn_dual(x16, self.packed_f16k_weight(desc), None if self.bias is None else self.bias.detach(), desc,
                                          (packed_gdn_f16k(gdn), gdn.inverse), products=products)
        return pre, y, desc.Ho, desc.Wo

    def run_f16k_res(self, x16, B, Hi, Wi, act=ops.ACT_NONE, res1=None, res2=None, res_ctot=0, out16=None, out_ctot=None, out_coff=0, y_pre=None):
        """No autograd: F16K -> F16K (optionally a channel view of `out16`) with F16K residual tensors added after the activation;
        `y_pre` (F16K, res_ctot channels) receives the activation's output before the adds (what a backward needs for its mask)."""
        oc = (self.out_channels + 15) // 16 * 16 if out_ctot is None else out_ctot
        bias = None if self.bias is None else self.bias.detach()
        if self.resident_supported(B, Hi, Wi) and act in (ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY):
            # 32 -> 32 3x3 layers: weights resident in LDS, persistent workgroups (conv_f16k.hip: conv3x3_resident_f16k)
            w = self.weight
            wp = _cached(self, "_packed_c3_cache", weight_key(w), (), lambda: ops.pack_conv3x3_resident_weight(w.detach()))
            return ops.conv3x3_resident(x16, wp, bias, B, self.in_channels, self.out_channels, Hi, Wi, act=act, y16=out16, out_ctot=oc, out_coff=out_coff,
                                        res1=res1, res2=res2, res_ctot=res_ctot, y_pre=y_pre)
        desc = self._desc_f16k(B, Hi, Wi, out_ctot=oc, out_coff=out_coff, act=act)
        return ops.conv2d_f16k_res(x16, self.packed_f16k_weight(desc), bias, desc, y16=out16,
                                   res1=res1, res2=res2, res_ctot=res_ctot, y_pre=y_pre)

    def resident_supported(self, B, Hi, Wi):
        """Conv2d(Cin -> 32 | 64, k3, s1, p1), Cin <= Cout, at W % 32 == 0 and H % 16 (8) == 0: the resident-weight kernel
        (MASIC_C3_RESIDENT=0: conv_f16k, A/B timing)."""
        return (_C3_RESIDENT and not self.transposed_conv and not self.masked_conv and self._geometry() == (3, 3, 1, 1)
                and ops.conv3x3_resident_supported(B, self.in_channels, self.out_channels, Hi, Wi))

    def few_supported(self, B, Hi, Wi):
        kh, kw, s_, p_ = self._geometry()
        return (not self.transposed_conv and self.out_channels <= 32 and self.in_channels % 16 == 0
                and ops.conv_f16k_supported(ops.make_conv_desc(B, self.in_channels, Hi, Wi, 32, kh, kw, s_, p_, prec=PREC_BF16)))

    def run_f16k_few(self, x16, B, Hi, Wi, res32=None):
        """Inference-only, Conv2d to <= 32 channels on an F16K input -> float32 NCHW (+ float32 residual): the weight is zero-padded
        to 32 output channels for the MFMA tile and only the real channels are stored (masic_conv_f16k_few_fwd)."""
        kh, kw, s_, p_ = self._geometry()
        desc = ops.make_conv_desc(B, self.in_channels, Hi, Wi, 32, kh, kw, s_, p_, prec=PREC_BF16)
        w = self.weight
        vkey = weight_key(w) + (None if self.bias is None else weight_key(self.bias),)

        def build():
            wpad = torch.zeros((32,) + tuple(w.shape[1:]), dtype=torch.float32, device=w.device)
            wpad[:self.out_channels] = w.detach()
            bpad = torch.zeros(32, dtype=torch.float32, device=w.device)
            if self.bias is not None:
                bpad[:self.out_channels] = self.bias.detach()
            return ops.pack_conv_f16k_weight(wpad, desc), bpad
        wp, bp = _cached(self, "_packed_few_cache", vkey, (B, Hi, Wi), build)
        return ops.conv2d_f16k_few(x16, wp, bp, desc, self.out_channels, res32=res32)

    def packed_gemm_dma_weight(self):
        """Per-128-channel-block k16-major pack of a 1x1 layer for the DMA-staged GEMM (conv_f16k.hip: gemm_f16k)."""
        w = self.weight
        key = weight_key(w)
        cache = self.__dict__.get("_packed_gemm_dma_cache")
        if cache is None or cache[0] != key:
            cache = (key, ops.pack_gemm_f16k_weight(w.detach().contiguous(), self.in_channels, self.out_channels, self.transposed_conv))
            self.__dict__["_packed_gemm_dma_cache"] = cache
        return cache[1]

    def invalidate_packed_weight(self):
        self.__dict__.pop("_packed_cache", None)

    def run(self, x, in_coff=0, out=None, out_coff=0, in_op=ops.INOP_NONE, act=ops.ACT_NONE, gate=None, gate_c=0,
            res1=None, res2=None):
        """Fused form: y = act(conv(in_op(x[:, in_coff:in_coff+Cin])) + bias) [* gate[:, gate_c]],
        optionally written into channels [out_coff, out_coff+Cout) of `out` (a torch.cat target).
        When autograd needs this node the same arithmetic is composed from differentiable HIP nodes instead
        (masic_amd/autograd.py)."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return self._run_autograd(x, in_coff, out, in_op, act, gate, gate_c, res1, res2)
        desc = self._desc(x.shape, in_ctot=x.shape[1], in_coff=in_coff,
                          out_ctot=None if out is None else out.shape[1], out_coff=out_coff,
                          in_op=in_op, act=act, gate_ctot=0 if gate is None else gate.shape[1], gate_c=gate_c)
        bias = None if self.bias is None else self.bias.detach()
        return ops.conv2d(x, self.packed_weight(desc), bias, desc, out=out, gate=gate, res1=res1, res2=res2)

    def _run_autograd(self, x, in_coff, out, in_op, act, gate, gate_c, res1, res2):
        from . import autograd as A
        if out is not None:
            raise RuntimeError("masic_amd: writing into a concat buffer is an inference-only fusion; use autograd.CatFn when training")
        if in_coff != 0 or x.shape[1] != self.in_channels:
            raise RuntimeError("masic_amd: channel-slice inputs are an inference-only fusion")
        if in_op == ops.INOP_ABS:
            x = A.AbsFn.apply(x)
        elif in_op != ops.INOP_NONE:
            raise RuntimeError("masic_amd: round() inputs have no gradient path (eval-mode fusion)")
        if act == ops.ACT_SOFTMAX_C:
            y = A.SoftmaxKFn.apply(A.conv(self, x), self.out_channels)
        else:
            y = A.conv(self, x, act)
        if gate is not None:
            y = A.GateFn.apply(y, gate, gate_c)
        for r in (res1, res2):
            if r is not None:
                y = A.AddFn.apply(y, r)
        return y

    def forward(self, x):
        return self.run(x)


class Conv2d(_PackedWeightMixin, nn.Conv2d):
    pass


class ConvTranspose2d(_PackedWeightMixin, nn.ConvTranspose2d):
    transposed_conv = True

    def forward(self, x, output_size=None):
        if output_size is not None:
            raise RuntimeError("masic_amd: ConvTranspose2d(output_size=...) is not supported")
        return self.run(x)
