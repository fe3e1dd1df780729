"""oracle/hsic_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU float32 restatement, as pure functions of (state_dict, inputs, noise), of the reference's
stereo-codec hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module -- and only as the checker.  The product (masic_amd/, compressai/,
coremasic/) never routes through it.

Parity status: PINNED against the reference's own Python implementation imported in the build
container (oracle/ref_import.py; tests/golden/make_goldens.py records max |oracle - reference|
per tensor in tests/golden/pin_report.json) -- except the perspective warp, whose arithmetic
lives in kornia==0.5.0 (reference readme.md:12), which is absent from /root/reference and from
this image: `warp_perspective` below restates kornia 0.5.0's published algorithm and is
"parity unpinned" (the rest of the path is pinned relative to it).

Each function cites the reference file:line it follows (paths relative to /root/reference).
State-dict keys are the reference's (`HSIC(N,M,K).state_dict()`).
"""
import math

import torch
import torch.nn.functional as F

REPARAM_OFFSET = 2.0 ** -18
PEDESTAL = REPARAM_OFFSET ** 2          # compressai/ops/parametrizers.py:52-54
LIK_BOUND = 1e-9                        # compressai/entropy_models/entropy_models.py:66
SCALE_BOUND = 0.11                      # entropy_models.py:728


# --------------------------------------------------------------------------- LowerBound
class _LowerBoundFn(torch.autograd.Function):
    """compressai/ops/bound_ops.py:36-56: max(x, b); grad passes iff x >= b or grad < 0."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return keep * g, None


def lower_bound(x, bound):
    b = torch.tensor([float(bound)], dtype=x.dtype, device=x.device)
    return _LowerBoundFn.apply(x, b)


def nonneg_reparam(p, minimum=0.0):
    """compressai/ops/parametrizers.py:47-64."""
    bound = (float(minimum) + PEDESTAL) ** 0.5
    return lower_bound(p, bound) ** 2 - torch.tensor([PEDESTAL], dtype=p.dtype)


# --------------------------------------------------------------------------- GDN
def gdn(x, beta_p, gamma_p, inverse=False, beta_min=1e-6):
    """compressai/layers/gdn.py:77-92."""
    C = x.shape[1]
    beta = nonneg_reparam(beta_p, beta_min)
    gamma = nonneg_reparam(gamma_p, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x ** 2, gamma, beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


# --------------------------------------------------------------------------- conv factories
def conv(x, sd, name, k=5, s=2):
    """compressai/models/utils.py:128-135."""
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=s, padding=k // 2)


def deconv(x, sd, name, k=5, s=2):
    """compressai/models/utils.py:138-146."""
    return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=s,
                              padding=k // 2, output_padding=s - 1)


def masked_weight(w):
    """compressai/layers/layers.py:64-78, mask type 'A' (centre and future taps zeroed).
    The reference zeroes `weight.data` in place and then convolves with the *parameter*, so the
    forward sees zeros at masked taps while autograd still delivers a (non-zero) gradient to every
    tap: value w*m, gradient identity."""
    _, _, kh, kw = w.shape
    m = torch.ones_like(w)
    m[:, :, kh // 2, kw // 2:] = 0
    m[:, :, kh // 2 + 1:] = 0
    return w + (w * m - w).detach()


# --------------------------------------------------------------------------- transforms
def encoder1(x, sd, p="encoder1"):
    """coremasic/mywork/MASIC.py:510-531."""
    t = conv(x, sd, p + ".g_a_conv1")
    t = gdn(t, sd[p + ".g_a_gdn1.beta"], sd[p + ".g_a_gdn1.gamma"])
    t = conv(t, sd, p + ".g_a_conv2")
    t = gdn(t, sd[p + ".g_a_gdn2.beta"], sd[p + ".g_a_gdn2.gamma"])
    t = conv(t, sd, p + ".g_a_conv3")
    t = gdn(t, sd[p + ".g_a_gdn3.beta"], sd[p + ".g_a_gdn3.gamma"])
    return conv(t, sd, p + ".g_a_conv4")


def encoder2(x1_warp, x2, sd, p="encoder2"):
    """MASIC.py:556-585."""
    t = conv(torch.cat((x1_warp, x2), dim=1), sd, p + ".pre_conv", s=1)
    t = gdn(t, sd[p + ".pre_gdn.beta"], sd[p + ".pre_gdn.gamma"])
    return encoder1(t, sd, p)


def decoder1(y_hat, sd, p="decoder1"):
    """MASIC.py:533-554."""
    t = deconv(y_hat, sd, p + ".g_s_conv1")
    t = gdn(t, sd[p + ".g_s_gdn1.beta"], sd[p + ".g_s_gdn1.gamma"], inverse=True)
    t = deconv(t, sd, p + ".g_s_conv2")
    t = gdn(t, sd[p + ".g_s_gdn2.beta"], sd[p + ".g_s_gdn2.gamma"], inverse=True)
    t = deconv(t, sd, p + ".g_s_conv3")
    t = gdn(t, sd[p + ".g_s_gdn3.beta"], sd[p + ".g_s_gdn3.gamma"], inverse=True)
    return deconv(t, sd, p + ".g_s_conv4")


def decoder2(y_hat, x1_hat_warp, sd, p="decoder2"):
    """MASIC.py:587-622."""
    t = decoder1(y_hat, sd, p)
    t = gdn(t, sd[p + ".after_gdn.beta"], sd[p + ".after_gdn.gamma"], inverse=True)
    return deconv(torch.cat((t, x1_hat_warp), dim=1), sd, p + ".after_conv", s=1)


def hyper_analysis(y, sd, p):
    """MASIC.py:170-187 (`encode_hyper`)."""
    t = F.relu(conv(torch.abs(y), sd, p + ".encode_hyper.0", s=1))
    t = F.relu(conv(t, sd, p + ".encode_hyper.2"))
    return conv(t, sd, p + ".encode_hyper.4")


def hyper_synthesis_up(z_hat, sd, p):
    """MASIC.py:678-691 (`h_s{1,2}_up`)."""
    t = F.leaky_relu(deconv(z_hat, sd, p + ".0"))
    t = F.leaky_relu(deconv(t, sd, p + ".2"))
    return conv(t, sd, p + ".4", k=3, s=1)


def _softmax_over_k(t, K):
    """MASIC.py:389-393: channel k*M+m is component k of latent channel m."""
    B, KM, h, w = t.shape
    return F.softmax(t.reshape(B, K, KM // K, h, w), dim=1).reshape(B, KM, h, w)


def gmm_params_y1(cat_in, sd, K, p="_h_s1_same_resolution"):
    """MASIC.py:330-396. The first two layers of each branch are ConvTranspose2d(k=1)."""
    def tconv1(x, name):
        return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"])

    def conv1(x, name):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"])

    s = F.relu(tconv1(cat_in, p + ".gmm_sigma.0"))
    s = F.relu(tconv1(s, p + ".gmm_sigma.2"))
    s = F.relu(conv1(s, p + ".gmm_sigma.4"))
    m = F.leaky_relu(tconv1(cat_in, p + ".gmm_means.0"))
    m = F.leaky_relu(tconv1(m, p + ".gmm_means.2"))
    m = conv1(m, p + ".gmm_means.4")
    w = F.leaky_relu(tconv1(cat_in, p + ".gmm_weights.0"))
    w = F.leaky_relu(tconv1(w, p + ".gmm_weights.2"))
    w = conv1(w, p + ".gmm_weights.4")
    return s, m, _softmax_over_k(w, K)


def gmm_params_y2(cat_in, sd, K, p="_h_s2_same_resolution"):
    """MASIC.py:399-468 (plain Conv2d k=1)."""
    def conv1(x, name):
        return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"])

    s = F.relu(conv1(cat_in, p + ".gmm_sigma.0"))
    s = F.relu(conv1(s, p + ".gmm_sigma.2"))
    s = F.relu(conv1(s, p + ".gmm_sigma.4"))
    m = F.leaky_relu(conv1(cat_in, p + ".gmm_means.0"))
    m = F.leaky_relu(conv1(m, p + ".gmm_means.2"))
    m = conv1(m, p + ".gmm_means.4")
    w = F.leaky_relu(conv1(cat_in, p + ".gmm_weights.0"))
    w = F.leaky_relu(conv1(w, p + ".gmm_weights.2"))
    w = conv1(w, p + ".gmm_weights.4")
    return s, m, _softmax_over_k(w, K)


def mask2weights(mask_r, sd, p="mask2weights_unit"):
    """MASIC.py:472-506: four 3x3 s2 convs 1->3->6->6->3, softmax over the 3 gates."""
    t = F.relu(conv(mask_r, sd, p + ".maskconv.0", k=3))
    t = F.relu(conv(t, sd, p + ".maskconv.2", k=3))
    t = F.relu(conv(t, sd, p + ".maskconv.4", k=3))
    t = conv(t, sd, p + ".maskconv.6", k=3)
    return F.softmax(t, dim=1)


# --------------------------------------------------------------------------- warp (kornia 0.5.0)
def _normal_transform_pixel(h, w, dtype):
    """kornia 0.5.0 geometry/conversions.py normal_transform_pixel (eps=1e-14)."""
    wd = 1e-14 if w == 1 else float(w - 1)
    hd = 1e-14 if h == 1 else float(h - 1)
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    t[0, 0] = t[0, 0] * 2.0 / wd
    t[1, 1] = t[1, 1] * 2.0 / hd
    return t.unsqueeze(0)


def warp_matrix(M, src_hw, dst_hw):
    """src_norm <- dst_norm 3x3 (kornia 0.5.0 imgwarp.py warp_perspective +
    normalize_homography): inverse(N_dst @ (M @ inverse(N_src))), all in M's dtype."""
    n_src = _normal_transform_pixel(src_hw[0], src_hw[1], M.dtype)
    n_dst = _normal_transform_pixel(dst_hw[0], dst_hw[1], M.dtype)
    dst_norm_trans_src_norm = n_dst @ (M @ torch.inverse(n_src))
    return torch.inverse(dst_norm_trans_src_norm)


def warp_grid(Minv_norm, h_out, w_out):
    """Normalised sampling grid [B,h,w,2]: create_meshgrid(normalized) -> transform_points
    -> convert_points_from_homogeneous(eps=1e-8) (kornia 0.5.0)."""
    dt = Minv_norm.dtype
    xs = torch.linspace(0, w_out - 1, w_out, dtype=dt)
    ys = torch.linspace(0, h_out - 1, h_out, dtype=dt)
    xs = (xs / (w_out - 1) - 0.5) * 2
    ys = (ys / (h_out - 1) - 0.5) * 2
    gx = xs.view(1, w_out).expand(h_out, w_out)
    gy = ys.view(h_out, 1).expand(h_out, w_out)
    m = Minv_norm
    B = m.shape[0]
    gx = gx.unsqueeze(0)
    gy = gy.unsqueeze(0)

    def row(r):
        a = m[:, r, 0].view(B, 1, 1)
        b = m[:, r, 1].view(B, 1, 1)
        c = m[:, r, 2].view(B, 1, 1)
        # fixed evaluation order (x*a + y*b) + c, shared with the HIP kernel
        return (gx * a + gy * b) + c

    X, Y, Z = row(0), row(1), row(2)
    eps = 1e-8
    scale = torch.where(Z.abs() > eps, 1.0 / (Z + eps), torch.ones_like(Z))
    return torch.stack((X * scale, Y * scale), dim=-1)


def warp_perspective(src, M, dsize, align_corners=True):
    """kornia 0.5.0 `warp_perspective(src, M, dsize)` with its defaults (bilinear, zeros,
    align_corners=None -> True). Reference call sites: MASIC.py:638,644,781,821,833.
    PARITY UNPINNED: kornia is not available; restated from its published source.
    `align_corners=False` is kornia <= 0.4.1's default (same (W - 1) normalisation of the homography, grid_sample without
    corner alignment): the alternative SURVEY.md section 8c asks to keep selectable."""
    H, W = src.shape[-2:]
    h_out, w_out = dsize
    grid = warp_grid(warp_matrix(M, (H, W), (h_out, w_out)), h_out, w_out)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


def mask(im1, H):
    """MASIC.py:627-649. The torch.where results are discarded there, so both masks keep
    their bilinear (non-binarised) border values."""
    B, _, h, w = im1.shape
    ones = torch.ones(B, 1, h, w, dtype=im1.dtype)
    mask_r = warp_perspective(ones, H, (h, w))
    mask_l = warp_perspective(mask_r, torch.inverse(H), (h, w))
    return mask_r, mask_l


# --------------------------------------------------------------------------- entropy models
def eb_logits(v, sd, p, detach=False):
    """entropy_models.py:350-369 (`_logits_cumulative`), v: (C,1,L)."""
    x = v
    for i in range(5):
        mat = sd[f"{p}._matrices.{i}"]
        bias = sd[f"{p}._biases.{i}"]
        if detach:
            mat, bias = mat.detach(), bias.detach()
        x = torch.matmul(F.softplus(mat), x) + bias
        if i < 4:
            fac = sd[f"{p}._factors.{i}"]
            if detach:
                fac = fac.detach()
            x = x + torch.tanh(fac) * torch.tanh(x)
    return x


def entropy_bottleneck(z, sd, p, training=False, noise=None):
    """entropy_models.py:384-411 (+ :98-125 quantise, :372-382 likelihood).
    `noise` (training): U(-1/2,1/2) tensor in the reference's draw layout (C,1,B*h*w)."""
    B, C, h, w = z.shape
    v = z.permute(1, 2, 3, 0).contiguous().reshape(C, 1, -1)
    med = sd[p + ".quantiles"][:, :, 1:2]
    if training:
        vq = v + noise
    else:
        vq = torch.round(v - med) + med
    lower = eb_logits(vq - 0.5, sd, p)
    upper = eb_logits(vq + 0.5, sd, p)
    sign = -torch.sign(lower + upper).detach()
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = lower_bound(lik, LIK_BOUND)
    back = lambda t: t.reshape(C, h, w, B).permute(3, 0, 1, 2).contiguous()
    return back(vq), back(lik)


def eb_aux_loss(sd, p):
    """entropy_models.py:345-348; target = log(2/1e-9 - 1) (:295-296)."""
    t = math.log(2 / 1e-9 - 1)
    target = torch.tensor([-t, 0.0, t], dtype=sd[p + ".quantiles"].dtype)
    logits = eb_logits(sd[p + ".quantiles"], sd, p, detach=True)
    return torch.abs(logits - target).sum()


def _std_cdf(t):
    """entropy_models.py:762-767."""
    return 0.5 * torch.erfc(-(2 ** -0.5) * t)


def quantize(y, training, noise=None):
    """entropy_models.py:98-125 with means=None (the GMM path never passes means: :851-853)."""
    return y + noise if training else torch.round(y)


def gmm_likelihood(y_hat, sigma, mu, wts, K):
    """entropy_models.py:808-858: sum_k w_k [Phi((.5-|y-mu_k|)/s_k) - Phi((-.5-|y-mu_k|)/s_k)]
    with s_k = LowerBound(sigma_k, 0.11), channel slice k*M:(k+1)*M; then LowerBound(1e-9)."""
    M = y_hat.shape[1]
    lik = None
    for k in range(K):
        sl = slice(k * M, (k + 1) * M)
        v = torch.abs(y_hat - mu[:, sl])
        s = lower_bound(sigma[:, sl], SCALE_BOUND)
        term = (_std_cdf((0.5 - v) / s) - _std_cdf((-0.5 - v) / s)) * wts[:, sl]
        lik = term if lik is None else lik + term
    return lower_bound(lik, LIK_BOUND)


# --------------------------------------------------------------------------- whole forward
NOISE_KEYS = ("z1", "y1_ctx", "y1", "z2", "y2_ctx", "y1_warp", "y2")   # SURVEY appendix D order


def hsic_forward(sd, x1, x2, H, K=5, training=False, noise=None, keep=False):
    """coremasic/mywork/MASIC.py:744-851. `sd`: reference state-dict (float32 CPU tensors).
    `noise`: dict over NOISE_KEYS (training only). With keep=True also returns intermediates
    used by per-stage parity tests."""
    nz = noise or {}
    q = lambda t, key: quantize(t, training, nz.get(key))
    h, w = x1.shape[-2:]
    y1 = encoder1(x1, sd)
    z1 = hyper_analysis(y1, sd, "_h_a1")
    z1_hat, z1_lik = entropy_bottleneck(z1, sd, "entropy_bottleneck1", training, nz.get("z1"))
    params1 = hyper_synthesis_up(z1_hat, sd, "h_s1_up")
    y1_ctx_in = q(y1, "y1_ctx")
    w_ctx1 = masked_weight(sd["context_prediction1.weight"])
    ctx1 = F.conv2d(y1_ctx_in, w_ctx1, sd["context_prediction1.bias"], padding=2)
    s1, m1, w1 = gmm_params_y1(torch.cat((params1, ctx1), dim=1), sd, K)
    y1_hat = q(y1, "y1")
    y1_lik = gmm_likelihood(y1_hat, s1, m1, w1, K)
    x1_hat = decoder1(y1_hat, sd)

    x1_warp = warp_perspective(x1, H, (h, w))
    y2 = encoder2(x1_warp, x2, sd)
    z2 = hyper_analysis(y2, sd, "_h_a2")
    z2_hat, z2_lik = entropy_bottleneck(z2, sd, "entropy_bottleneck2", training, nz.get("z2"))
    params2 = hyper_synthesis_up(z2_hat, sd, "h_s2_up")
    y2_ctx_in = q(y2, "y2_ctx")
    w_ctx2 = masked_weight(sd["context_prediction2.weight"])
    ctx2 = F.conv2d(y2_ctx_in, w_ctx2, sd["context_prediction2.bias"], padding=2)

    mask_r, mask_l = mask(x1, H)
    gates = mask2weights(mask_r, sd)
    x1_hat_warp = warp_perspective(x1_hat, H, (h, w))
    y1_warp = encoder1(x1_hat_warp, sd)
    y1_warp_hat = q(y1_warp, "y1_warp")
    cat2 = torch.cat((params2 * gates[:, 0:1], ctx2 * gates[:, 1:2], y1_warp_hat * gates[:, 2:3]), dim=1)
    s2, m2, w2 = gmm_params_y2(cat2, sd, K)
    y2_hat = q(y2, "y2")
    y2_lik = gmm_likelihood(y2_hat, s2, m2, w2, K)
    x2_hat = decoder2(y2_hat, x1_hat_warp, sd)

    out = {
        "x1_hat": x1_hat, "x2_hat": x2_hat, "y1_hat": y1_hat, "z1_hat": z1_hat,
        "x1_mask_R": mask_r, "x1_mask_L": mask_l,
        "likelihoods": {"y1": y1_lik, "y2": y2_lik, "z1": z1_lik, "z2": z2_lik},
    }
    if keep:
        out["_aux"] = {
            "y1": y1, "z1": z1, "params1": params1, "ctx1": ctx1, "sigma1": s1, "mu1": m1, "w1": w1,
            "x1_warp": x1_warp, "y2": y2, "z2": z2, "z2_hat": z2_hat, "params2": params2, "ctx2": ctx2,
            "gates": gates, "x1_hat_warp": x1_hat_warp, "y1_warp": y1_warp, "cat2": cat2,
            "sigma2": s2, "mu2": m2, "w2": w2, "y2_hat": y2_hat,
        }
    return out


def symbols(out_aux, sd):
    """int32 symbol streams that feed the range coder (entropy_models.py:123-125 'symbols' mode;
    z symbols are taken about the medians, MASIC.py:863 / entropy_models.py:420-423)."""
    med1 = sd["entropy_bottleneck1.quantiles"][:, 0, 1].view(1, -1, 1, 1)
    med2 = sd["entropy_bottleneck2.quantiles"][:, 0, 1].view(1, -1, 1, 1)
    a = out_aux
    return {
        "y1": torch.round(a["y1"]).int(), "y2": torch.round(a["y2"]).int(),
        "z1": torch.round(a["z1"] - med1).int(), "z2": torch.round(a["z2"] - med2).int(),
    }


def rd_loss(out, x1, x2, lmbda):
    """coremasic/mywork/newtrain_codec_real.py:66-87."""
    B, _, h, w = x1.shape
    n = B * h * w
    bpp = sum(torch.log(l).sum() / (-math.log(2) * n) for l in out["likelihoods"].values())
    mse1 = F.mse_loss(out["x1_hat"], x1)
    mse2 = F.mse_loss(out["x2_hat"], x2)
    mse = mse1 + mse2
    return {
        "bpp_loss": bpp, "mse_loss": mse, "loss": lmbda * 255 ** 2 * mse + bpp,
        "mse1": mse1, "mse2": mse2,
        "psnr1": 10 * math.log10(1 / mse1.item()), "psnr2": 10 * math.log10(1 / mse2.item()),
        "bpp_y1": torch.log(out["likelihoods"]["y1"]).sum() / (-math.log(2) * n),
        "bpp_y2": torch.log(out["likelihoods"]["y2"]).sum() / (-math.log(2) * n),
        "bpp_z1": torch.log(out["likelihoods"]["z1"]).sum() / (-math.log(2) * n),
        "bpp_z2": torch.log(out["likelihoods"]["z2"]).sum() / (-math.log(2) * n),
    }


# --------------------------------------------------------------------------- CQE network
def _residual_block(x, sd, p):
    """compressai/layers/layers.py:160-190 (in_ch == out_ch: no 1x1 skip)."""
    t = F.leaky_relu(F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1))
    t = F.leaky_relu(F.conv2d(t, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1))
    return t + x


def _enhancement_block(x, sd, p):
    """coremasic/mywork/MASIC.py:149-164."""
    t = _residual_block(x, sd, p + ".RB1")
    t = _residual_block(t, sd, p + ".RB2")
    t = _residual_block(t, sd, p + ".RB3")
    return t + x


def mask2weights_en(m, sd, p="mask2weights_unit"):
    """MASIC.py:1411-1434: four 3x3 s1 convs 1->2->4->4->2, softmax over the 2 gates."""
    t = F.relu(conv(m, sd, p + ".maskconv.0", k=3, s=1))
    t = F.relu(conv(t, sd, p + ".maskconv.2", k=3, s=1))
    t = F.relu(conv(t, sd, p + ".maskconv.4", k=3, s=1))
    return F.softmax(conv(t, sd, p + ".maskconv.6", k=3, s=1), dim=1)


def independent_en_forward(sd, x1_hat, x2_hat, H):
    """coremasic/mywork/MASIC.py:1456-1501."""
    h, w = x1_hat.shape[-2:]
    H_inv = torch.inverse(H)
    mask_r, mask_l = mask(x1_hat, H)
    w_r = mask2weights_en(mask_r, sd)
    w_l = mask2weights_en(mask_l, sd)
    x1_warp = warp_perspective(x1_hat, H, (h, w))
    x2_warp = warp_perspective(x2_hat, H_inv, (h, w))
    c3 = lambda t, n: F.conv2d(t, sd[n + ".weight"], sd[n + ".bias"], padding=1)
    x1c, x2c = c3(x1_hat, "conv0"), c3(x2_hat, "conv0")
    o1 = torch.cat((x2_warp * w_l[:, 0:1], x1_hat * w_l[:, 1:2]), dim=1)
    o2 = torch.cat((x1_warp * w_r[:, 0:1], x2_hat * w_r[:, 1:2]), dim=1)
    o1 = _enhancement_block(c3(o1, "conv1"), sd, "EBl1")
    o2 = _enhancement_block(c3(o2, "conv1"), sd, "EBr1")
    o1w = warp_perspective(o1, H, (h, w))
    o2w = warp_perspective(o2, H_inv, (h, w))
    n1 = torch.cat((o1 * w_l[:, 1:2], o2w * w_l[:, 0:1]), dim=1)
    n2 = torch.cat((o2 * w_r[:, 1:2], o1w * w_r[:, 0:1]), dim=1)
    o1 = _enhancement_block(n1, sd, "EBl2")
    o2 = _enhancement_block(n2, sd, "EBr2")
    o1 = _enhancement_block(torch.cat((o1, x1c), dim=1), sd, "EBl3")
    o2 = _enhancement_block(torch.cat((o2, x2c), dim=1), sd, "EBr3")
    return {"x1_hat": c3(o1, "conv2") + x1_hat, "x2_hat": c3(o2, "conv2") + x2_hat}
