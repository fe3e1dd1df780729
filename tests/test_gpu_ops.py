"""GPU parity tests (-m gpu): every HIP kernel, called through the C ABI (ctypes, masic_amd/ops.py),
against the CPU oracle on the same seeded inputs.  Tolerance: 1e-4 relative (north_star) on float32
outputs; integer symbols bit-exact outside the declared tie zone."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hsic_oracle as O
from tests.util import assert_close, load_npz

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from masic_amd import ops
    return ops


def _rand(*shape, seed=0, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(size=shape) * scale).astype(np.float32))


CONV_CASES = [
    # name,            B, Cin, H,  W,  Cout, k, s, transposed, masked, in_op, act
    ("g_a_conv1",      2, 3,   64, 96, 128,  5, 2, False, False, 0, 0),
    ("g_a_conv2",      2, 128, 48, 64, 128,  5, 2, False, False, 0, 0),
    ("g_a_conv4_rag",  1, 128, 24, 56, 192,  5, 2, False, False, 0, 0),
    ("h_a0_abs_relu",  2, 192, 16, 24, 128,  5, 1, False, False, 1, 1),
    ("h_a2_small",     2, 128, 16, 8,  128,  5, 2, False, False, 0, 1),
    ("h_a4_tiny",      2, 128, 8,  4,  128,  5, 2, False, False, 0, 0),
    ("h_s_up0",        2, 128, 4,  6,  192,  5, 2, True,  False, 0, 2),
    ("h_s_up2",        1, 192, 8,  12, 288,  5, 2, True,  False, 0, 2),
    ("h_s_up4_3x3",    2, 288, 16, 24, 384,  3, 1, False, False, 0, 0),
    ("ctx_masked_rnd", 2, 192, 16, 24, 384,  5, 1, False, True,  2, 0),
    ("g_s_conv1",      2, 192, 8,  12, 128,  5, 2, True,  False, 0, 0),
    ("g_s_conv3",      1, 128, 32, 48, 128,  5, 2, True,  False, 0, 0),
    ("g_s_conv4_to3",  2, 128, 32, 48, 3,    5, 2, True,  False, 0, 0),
    ("pre_conv_6to3",  2, 6,   64, 96, 3,    5, 1, False, False, 0, 0),
    ("after_conv",     2, 6,   64, 96, 3,    5, 1, True,  False, 0, 0),
    ("gmm_1x1_conv",   2, 768, 16, 24, 960,  1, 1, False, False, 0, 1),
    ("gmm_1x1_deconv", 2, 768, 16, 24, 1152, 1, 1, True,  False, 0, 2),
    ("m2w_1to3",       2, 1,   64, 96, 3,    3, 2, False, False, 0, 1),
    ("m2w_6to6",       2, 6,   16, 24, 6,    3, 2, False, False, 0, 1),
    ("odd_sizes",      1, 20,  19, 37, 72,   5, 2, False, False, 0, 2),
    ("cqe_3x3_32",     1, 32,  40, 72, 32,   3, 1, False, False, 0, 2),
]


def _oracle_conv(x, w, b, k, s, transposed, masked, in_op, act):
    if in_op == 1:
        x = x.abs()
    elif in_op == 2:
        x = torch.round(x)
    if masked:
        w = O.masked_weight(w)
    if transposed:
        y = F.conv_transpose2d(x, w, b, stride=s, padding=k // 2, output_padding=s - 1)
    else:
        y = F.conv2d(x, w, b, stride=s, padding=k // 2)
    return {0: lambda t: t, 1: F.relu, 2: F.leaky_relu}[act](y)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_variants(case):
    ops = _ops()
    name, B, Cin, H, W, Cout, k, s, tr, masked, in_op, act = case
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w = _rand(*wshape, seed=2, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    ref = _oracle_conv(x, w, b, k, s, tr, masked, in_op, act)
    desc = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, k // 2, transposed=tr, masked=masked, in_op=in_op, act=act)
    packed = ops.pack_conv_weight(w.to(DEV), desc)
    y = ops.conv2d(x.to(DEV), packed, b.to(DEV), desc)
    assert_close(y, ref, name)


def test_conv_views_and_gate():
    """Input read from a channel slice, output written into a slice of a wider buffer, gated."""
    ops = _ops()
    B, H, W = 2, 16, 24
    xbig = _rand(B, 40, H, W, seed=4)
    w = _rand(64, 24, 3, 3, seed=5, scale=0.1)
    b = _rand(64, seed=6, scale=0.1)
    gate = torch.rand(B, 3, H, W)
    ref = F.conv2d(xbig[:, 8:32], w, b, padding=1) * gate[:, 1:2]
    out = torch.full((B, 100, H, W), 7.0, device=DEV)
    desc = ops.make_conv_desc(B, 24, H, W, 64, 3, 3, 1, 1, in_ctot=40, in_coff=8, out_ctot=100, out_coff=20, gate_ctot=3, gate_c=1)
    ops.conv2d(xbig.to(DEV), ops.pack_conv_weight(w.to(DEV), desc), b.to(DEV), desc, out=out, gate=gate.to(DEV))
    assert_close(out[:, 20:84], ref, "view+gate")
    assert float((out[:, :20] - 7).abs().max()) == 0 and float((out[:, 84:] - 7).abs().max()) == 0


def test_conv_channel_softmax_epilogue():
    ops = _ops()
    x = _rand(2, 6, 16, 24, seed=7)
    w = _rand(3, 6, 3, 3, seed=8, scale=0.3)
    b = _rand(3, seed=9, scale=0.1)
    ref = F.softmax(F.conv2d(x, w, b, stride=2, padding=1), dim=1)
    desc = ops.make_conv_desc(2, 6, 16, 24, 3, 3, 3, 2, 1, act=ops.ACT_SOFTMAX_C)
    y = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV), desc), b.to(DEV), desc)
    assert_close(y, ref, "softmax_c")


@pytest.mark.parametrize("C,H,W,inverse", [(128, 32, 48, False), (128, 16, 24, True), (128, 8, 8, False),
                                           (3, 64, 96, False), (3, 40, 56, True), (16, 12, 20, False)])
def test_gdn(C, H, W, inverse):
    ops = _ops()
    rs = np.random.RandomState(C + H)
    x = _rand(2, C, H, W, seed=10, scale=3.0)
    from masic_amd import synth
    beta = synth.synth_tensor("g.beta", (C,), rs)
    gamma = synth.synth_tensor("g.gamma", (C, C), rs)
    gamma[0, 1] = 1e-7      # below the 2^-18 reparametrisation bound
    beta[0] = 1e-5          # below sqrt(beta_min + 2^-36)
    ref = O.gdn(x, beta, gamma, inverse=inverse)
    y = ops.gdn(x.to(DEV), beta.to(DEV), gamma.to(DEV), inverse=inverse)
    assert_close(y, ref, f"gdn C={C}")


def _eb_sd(C, seed):
    from masic_amd import synth
    import MASIC
    from compressai.entropy_models import EntropyBottleneck
    eb = EntropyBottleneck(C)
    sd = synth.synth_state_dict({"eb." + k: v for k, v in eb.state_dict().items()}, seed=seed)
    return sd


@pytest.mark.parametrize("training", [False, True])
def test_entropy_bottleneck(training):
    ops = _ops()
    C, B, H, W = 24, 3, 4, 6
    sd = _eb_sd(C, 5)
    z = _rand(B, C, H, W, seed=11, scale=4.0)
    noise = torch.from_numpy(np.random.RandomState(12).uniform(-0.5, 0.5, size=(C, 1, H * W * B)).astype(np.float32))
    ref_hat, ref_lik = O.entropy_bottleneck(z, sd, "eb", training=training, noise=noise)
    table = ops.eb_param_table([sd[f"eb._matrices.{i}"].to(DEV) for i in range(5)], [sd[f"eb._biases.{i}"].to(DEV) for i in range(5)],
                               [sd[f"eb._factors.{i}"].to(DEV) for i in range(4)])
    med = sd["eb.quantiles"][:, 0, 1].contiguous().to(DEV)
    z_hat, lik = ops.entropy_bottleneck(z.to(DEV), table, med, training=training, noise=noise.to(DEV))
    assert_close(z_hat, ref_hat, "z_hat", rtol=1e-6)
    assert_close(lik, ref_lik, "z_lik")
    # elementwise relative check on the likelihoods (they span orders of magnitude)
    rel = ((lik.cpu() - ref_lik).abs() / ref_lik).max()
    assert float(rel) < 1e-3, float(rel)
    aux = ops.entropy_bottleneck_auxloss(table, sd["eb.quantiles"].to(DEV))
    assert abs(float(aux) - float(O.eb_aux_loss(sd, "eb"))) <= 1e-5 * float(O.eb_aux_loss(sd, "eb"))


@pytest.mark.parametrize("training,logits", [(False, False), (True, False), (False, True)])
def test_gmm_likelihood(training, logits):
    ops = _ops()
    B, M, K, H, W = 2, 24, 5, 6, 10
    y = _rand(B, M, H, W, seed=13, scale=6.0)
    sigma = F.relu(_rand(B, K * M, H, W, seed=14, scale=2.0))       # zeros -> 0.11 bound active
    mu = _rand(B, K * M, H, W, seed=15, scale=5.0)
    raw = _rand(B, K * M, H, W, seed=16, scale=2.0)
    y[0, 0, 0, :4] = torch.tensor([40.0, -35.0, 0.5, 2.5])          # far tails -> 1e-9 bound; exact ties of round()
    wts = O._softmax_over_k(raw, K)
    noise = torch.from_numpy(np.random.RandomState(17).uniform(-0.5, 0.5, size=y.shape).astype(np.float32))
    y_hat_ref = O.quantize(y, training, noise)
    lik_ref = O.gmm_likelihood(y_hat_ref, sigma, mu, wts, K)
    d = lambda t: t.to(DEV)
    if logits:
        y_hat, lik, wout = ops.gmm_likelihood(d(y), d(sigma), d(mu), d(raw), K, training=training, noise=d(noise),
                                              weights_are_logits=True, want_weights=True)
        assert_close(wout, wts, "softmax_k in-kernel", rtol=1e-5)
    else:
        y_hat, lik = ops.gmm_likelihood(d(y), d(sigma), d(mu), d(wts), K, training=training, noise=d(noise))
    assert torch.equal(y_hat.cpu(), y_hat_ref), "quantised latents must be identical"
    assert_close(lik, lik_ref, "gmm lik")
    big = lik_ref > 1e-6
    assert float(((lik.cpu() - lik_ref).abs() / lik_ref)[big].max()) < 2e-3
    assert float(lik.min()) >= 1e-9 * 0.999 and int((lik_ref <= 1e-9).sum()) > 0
    assert_close(ops.softmax_k(d(raw), K), wts, "softmax_k", rtol=1e-5)


def test_quantize_symbols_views():
    ops = _ops()
    B, C, H, W = 2, 12, 5, 7
    x = _rand(B, C, H, W, seed=18, scale=5.0)
    x[0, 0, 0, :6] = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 1e-8])
    gate = torch.rand(B, 3, H, W)
    out = torch.zeros(B, 30, H, W, device=DEV)
    ops.quantize(x.to(DEV), "dequantize", out=out, out_coff=10, gate=gate.to(DEV), gate_c=2)
    assert torch.equal(out[:, 10:22].cpu(), torch.round(x) * gate[:, 2:3])
    med = _rand(C, seed=19)
    sym = ops.symbols(x.to(DEV), med.to(DEV))
    assert sym.dtype == torch.int32
    assert torch.equal(sym.cpu(), torch.round(x - med.view(1, -1, 1, 1)).int())
    assert torch.equal(ops.symbols(x.to(DEV)).cpu(), torch.round(x).int())


@pytest.mark.parametrize("H,W", [(64, 96), (40, 56)])
def test_warp_and_masks(H, W):
    ops = _ops()
    from masic_amd import synth
    x, _, hm = synth.synth_inputs(2, H, W, seed=4)
    ref = O.warp_perspective(x, hm, (H, W))
    from masic_amd.homography import warp_matrices
    m, mb = warp_matrices(hm.to(DEV), (H, W), (H, W), want_inverse=True)
    assert torch.equal(m.cpu(), O.warp_matrix(hm, (H, W), (H, W))), "host float32 chain must equal the oracle's bitwise"
    y = ops.warp_perspective(x.to(DEV), m, (H, W))
    assert_close(y, ref, "warp")
    mr_ref, ml_ref = O.mask(x, hm)
    mr = ops.warp_perspective(None, m, (H, W), ones_like=(2, H, W))
    ml = ops.warp_perspective(mr, mb, (H, W))
    assert_close(mr, mr_ref, "mask_R")
    assert_close(ml, ml_ref, "mask_L")
    # float64 device kernel: agrees with the float32 chain to the conditioning of that chain
    md = ops.warp_matrix(hm.to(DEV), (H, W), (H, W))
    assert_close(md, m, "device f64 warp matrix", rtol=1e-5)
    assert_close(ops.warp_perspective(x.to(DEV), md, (H, W)), ref, "warp (device matrix)", rtol=1e-3)
    assert int(((mr_ref > 0) & (mr_ref < 1)).sum()) > 0     # non-binarised border is exercised


def test_warp_align_corners_false_convention():
    """The warp's grid_sample convention is a setting (kornia, absent here, changed its default between 0.4.1 and
    the reference's pinned 0.5.0 and nothing in the reference pins a result): with align_corners=False the forward, the backward
    and the F16K gated warp follow the oracle run with the same setting, and the default is restored afterwards."""
    ops = _ops()
    from masic_amd import synth
    from masic_amd.homography import warp_matrices
    H, W = 48, 80
    x, _, hm = synth.synth_inputs(2, H, W, seed=6)
    m, _ = warp_matrices(hm.to(DEV), (H, W), (H, W), want_inverse=True)
    assert ops.get_warp_align_corners()
    ref_true = O.warp_perspective(x, hm, (H, W))
    xr = x.clone().requires_grad_(True)
    ref_false = O.warp_perspective(xr, hm, (H, W), align_corners=False)
    g = _rand(2, 3, H, W, seed=61)
    ref_false.backward(g)
    assert float((ref_false.detach() - ref_true).abs().max()) > 1e-3          # the two conventions differ on this input
    ops.set_warp_align_corners(False)
    try:
        assert not ops.get_warp_align_corners()
        assert_close(ops.warp_perspective(x.to(DEV), m, (H, W)), ref_false.detach(), "warp, align_corners=False")
        assert_close(ops.warp_perspective_bwd(g.to(DEV), m, tuple(x.shape)), xr.grad, "warp backward, align_corners=False", rtol=1e-4)
        # F16K gated warp (Independent_EN's bf16 path): the same sampling on bf16 records
        C = 16
        f = _rand(2, C, H, W, seed=62)
        f16 = ops.nchw_to_f16k(f.to(DEV))
        dst = ops.f16k_empty(2, C, H, W, DEV)
        ops.f16k_gate(f16, 2, C, H, W, dst, C, 0, minv=m)
        want = O.warp_perspective(f.bfloat16().float(), hm, (H, W), align_corners=False)
        got = ops.f16k_to_nchw_dev(dst, 2, C, H, W)
        assert float((got.cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())      # bf16 store of the result
    finally:
        ops.set_warp_align_corners(True)
    assert_close(ops.warp_perspective(x.to(DEV), m, (H, W)), ref_true, "warp, default convention restored")


def test_reductions_and_small_helpers():
    ops = _ops()
    lik = torch.rand(3, 50, 17, 11) * 0.9 + 1e-6
    a, b = _rand(2, 3, 64, 96, seed=20), _rand(2, 3, 64, 96, seed=21)
    s = ops.sum_log(lik.to(DEV))
    assert abs(float(s) - float(torch.log(lik.double()).sum())) <= 1e-5 * abs(float(torch.log(lik.double()).sum()))
    e = ops.sse(a.to(DEV), b.to(DEV))
    assert abs(float(e) - float(((a.double() - b.double()) ** 2).sum())) <= 1e-6 * float(((a.double() - b.double()) ** 2).sum())
    x = _rand(1000, seed=22)
    m = (torch.rand(1000) > 0.5).float()
    xd = x.to(DEV)
    ops.mul_inplace(xd, m.to(DEV))
    assert torch.equal(xd.cpu(), x * m)
    assert torch.equal(ops.lower_bound(x.to(DEV), 0.11).cpu(), torch.clamp(x, min=0.11))
    g = _rand(1000, seed=23)
    ref = torch.where((x >= 0.11) | (g < 0), g, torch.zeros_like(g))
    assert torch.equal(ops.lower_bound_bwd(x.to(DEV), g.to(DEV), 0.11).cpu(), ref)


BF16_CASES = [c for c in CONV_CASES if c[5] > 8]


@pytest.mark.parametrize("case", BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_conv_variants_bf16_operands(case):
    """bf16-operand contraction (float32 accumulate): against the float32 oracle evaluated on bf16-rounded operands
    (tight: only the accumulation order differs) and against the plain float32 oracle (loose: operand rounding)."""
    ops = _ops()
    from masic_amd._lib import PREC_BF16
    name, B, Cin, H, W, Cout, k, s, tr, masked, in_op, act = case
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w = _rand(*wshape, seed=2, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    xin = x.abs() if in_op == 1 else (torch.round(x) if in_op == 2 else x)
    ref_q = _oracle_conv(xin.bfloat16().float(), w.bfloat16().float(), b, k, s, tr, masked, 0, act)
    ref = _oracle_conv(x, w, b, k, s, tr, masked, in_op, act)
    desc = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, k // 2, transposed=tr, masked=masked, in_op=in_op, act=act, prec=PREC_BF16)
    y = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV), desc), b.to(DEV), desc)
    assert_close(y, ref_q, name + ":bf16 vs bf16-rounded oracle", rtol=2e-5)
    assert_close(y, ref, name + ":bf16 vs f32 oracle", rtol=3e-2)


@pytest.mark.parametrize("H,W,inverse", [(32, 48, False), (16, 24, True), (8, 8, False)])
def test_gdn_bf16x3_split_product(H, W, inverse):
    """C=128 GDN with the contraction as a bf16 hi/lo split product (used when the forward runs with bf16 operands):
    stays within a few 1e-5 of the float32 oracle."""
    ops = _ops()
    from masic_amd import synth
    from masic_amd._lib import PREC_BF16
    C = 128
    rs = np.random.RandomState(H)
    x = _rand(2, C, H, W, seed=10, scale=3.0)
    beta = synth.synth_tensor("g.beta", (C,), rs)
    gamma = synth.synth_tensor("g.gamma", (C, C), rs)
    gamma[0, 1] = 1e-7
    ref = O.gdn(x, beta, gamma, inverse=inverse)
    y = ops.gdn(x.to(DEV), beta.to(DEV), gamma.to(DEV), inverse=inverse, prec=PREC_BF16)
    e = assert_close(y, ref, "gdn bf16x3", rtol=5e-5)
    print(f"gdn bf16x3 relative error {e:.2e}")


@pytest.mark.parametrize("B,Cin,Cmid,Cout,H,W,tr", [(2, 768, 1152, 960, 16, 24, True), (1, 960, 1152, 768, 8, 12, False), (2, 100, 72, 50, 5, 7, False)])
def test_gemm1x1_bf16_stack_f16k(B, Cin, Cmid, Cout, H, W, tr):
    """Two chained 1x1 layers through the F16K activation layout against the float32 oracle on bf16-rounded operands."""
    ops = _ops()
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    w0 = _rand(*((Cin, Cmid) if tr else (Cmid, Cin)), seed=2, scale=(2.0 / Cin) ** 0.5)
    b0 = _rand(Cmid, seed=3, scale=0.1)
    w1 = _rand(Cout, Cmid, seed=4, scale=(2.0 / Cmid) ** 0.5)
    b1 = _rand(Cout, seed=5, scale=0.1)
    q = lambda t: t.bfloat16().float()
    w0c = (q(w0).t() if tr else q(w0)).reshape(Cmid, Cin, 1, 1)
    mid = F.leaky_relu(F.conv2d(q(x), w0c, b0))
    ref = F.relu(F.conv2d(q(mid), q(w1).reshape(Cout, Cmid, 1, 1), b1))
    xf = ops.nchw_to_f16k(x.to(DEV))
    p0 = ops.pack_gemm1x1_weight(w0.to(DEV), Cin, Cmid, tr)
    p1 = ops.pack_gemm1x1_weight(w1.to(DEV), Cmid, Cout, False)
    t = ops.gemm1x1_bf16(xf, p0, b0.to(DEV), B, Cin, Cmid, H, W, ops.ACT_LEAKY)
    y = ops.gemm1x1_bf16(t, p1, b1.to(DEV), B, Cmid, Cout, H, W, ops.ACT_RELU, want_nchw=True)
    assert_close(y, ref, "gemm1x1 bf16 stack", rtol=1e-3)     # an intermediate that re-rounds to the neighbouring bf16 value moves the output by ~1e-4


F16K_CASES = [c for c in CONV_CASES if c[2] % 16 == 0 and c[5] >= 64 and c[5] % 32 == 0 and c[6] > 1 and c[10] == 0] + [
    ("ctx_masked_f16k", 2, 192, 16, 24, 384, 5, 1, False, True, 0, 0),
    ("g_a_conv2_big",   1, 128, 136, 200, 128, 5, 2, False, False, 0, 0),     # several tiles, ragged right / bottom edges
    ("g_s_conv3_big",   1, 128, 40, 72, 128, 5, 2, True, False, 0, 0),
    ("g_a_conv2_b8",    8, 128, 40, 72, 128, 5, 2, False, False, 0, 0),       # 8 images: one image per XCD (the block -> tile map of B % 8 == 0)
    ("g_s_conv3_b16",  16, 128, 12, 20, 192, 5, 2, True, False, 0, 0),        # two image slots per XCD, two output-channel blocks, 4 phases
]


@pytest.mark.parametrize("case", F16K_CASES, ids=[c[0] for c in F16K_CASES])
def test_conv_f16k(case):
    """Convolutions on F16K (channel-blocked bf16) activations, both operands DMA-staged (conv_f16k.hip): float32 NCHW
    output against the float32 oracle on bf16-rounded operands (only the accumulation order differs), F16K output against
    the same rounded to bf16; ragged tiles, transposed phases and the masked conv included."""
    ops = _ops()
    from masic_amd._lib import PREC_BF16
    name, B, Cin, H, W, Cout, k, s, tr, masked, in_op, act = case
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    w = _rand(*wshape, seed=2, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    ref = _oracle_conv(x.bfloat16().float(), w.bfloat16().float(), b, k, s, tr, masked, 0, act)
    d = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, k // 2, transposed=tr, masked=masked, act=act, prec=PREC_BF16)
    assert ops.conv_f16k_supported(d), "layer shapes of the codec must have an F16K configuration"
    x16 = ops.nchw_to_f16k(x.to(DEV))
    wp = ops.pack_conv_f16k_weight(w.to(DEV), d)
    y = ops.conv2d_f16k(x16, wp, b.to(DEV), d, want_nchw=True)
    assert_close(y, ref, name + ":f16k -> nchw", rtol=2e-5)
    y16 = ops.conv2d_f16k(x16, wp, b.to(DEV), d)
    yb = ops.f16k_to_nchw(y16, B, Cout, d.Ho, d.Wo)
    assert_close(yb, ref, name + ":f16k -> f16k", rtol=2.0 ** -8)      # bf16 rounding of the output


def test_conv_f16k_views_gate_and_unsupported():
    """Channel views on both sides (a slice of a wider F16K buffer in, a slice of a concat buffer out, gated), and the
    shapes the path declines (the caller falls back to the NCHW kernels)."""
    ops = _ops()
    from masic_amd._lib import PREC_BF16
    B, H, W, k = 2, 16, 24, 3
    xall = _rand(B, 224, H, W, seed=4)
    w = _rand(384, 192, k, k, seed=5, scale=0.05)
    b = _rand(384, seed=6, scale=0.1)
    gates = torch.rand(B, 3, H, W)
    ref = F.conv2d(xall[:, 32:].bfloat16().float(), w.bfloat16().float(), b, padding=1) * gates[:, 1:2]
    d = ops.make_conv_desc(B, 192, H, W, 384, k, k, 1, 1, in_ctot=224, in_coff=32, out_ctot=768, out_coff=128, gate_ctot=3, gate_c=1,
                           prec=PREC_BF16)
    out = torch.full((B, 768, H, W), 7.0, device=DEV)
    ops.conv2d_f16k(ops.nchw_to_f16k(xall.to(DEV)), ops.pack_conv_f16k_weight(w.to(DEV), d), b.to(DEV), d, out_nchw=out, gate=gates.to(DEV))
    assert_close(out[:, 128:512], ref, "f16k views+gate", rtol=2e-5)
    assert torch.all(out[:, :128] == 7.0) and torch.all(out[:, 512:] == 7.0)
    for cin, cout, kk, in_op in ((3, 128, 5, 0), (128, 3, 5, 0), (768, 960, 1, 0), (192, 128, 5, 1), (20, 72, 5, 0)):
        dd = ops.make_conv_desc(1, cin, 16, 16, cout, kk, kk, 1, kk // 2, in_op=in_op, prec=PREC_BF16)
        assert not ops.conv_f16k_supported(dd)


@pytest.mark.parametrize("Cin,H,W,tr,inverse", [(128, 48, 64, False, False), (192, 8, 12, True, True), (128, 24, 40, True, True)])
def test_conv_f16k_fused_gdn(Cin, H, W, tr, inverse):
    """(Inverse) GDN in the epilogue of the 128-channel convolutions: against oracle GDN of the oracle convolution
    (bf16-rounded conv operands; the GDN contraction itself is a bf16 hi/lo split product, ~1e-5)."""
    ops = _ops()
    from masic_amd import synth
    from masic_amd._lib import PREC_BF16
    B, C, k, s = 2, 128, 5, 2
    rs = np.random.RandomState(H)
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    w = _rand(*((Cin, C, k, k) if tr else (C, Cin, k, k)), seed=2, scale=(2.0 / (Cin * k * k)) ** 0.5)
    b = _rand(C, seed=3, scale=0.1)
    beta = synth.synth_tensor("g.beta", (C,), rs)
    gamma = synth.synth_tensor("g.gamma", (C, C), rs)
    ref = O.gdn(_oracle_conv(x.bfloat16().float(), w.bfloat16().float(), b, k, s, tr, False, 0, 0), beta, gamma, inverse=inverse)
    d = ops.make_conv_desc(B, Cin, H, W, C, k, k, s, 2, transposed=tr, prec=PREC_BF16)
    x16 = ops.nchw_to_f16k(x.to(DEV))
    wp = ops.pack_conv_f16k_weight(w.to(DEV), d)
    gp = ops.pack_gdn_f16k(beta.to(DEV), gamma.to(DEV))
    ops.set_fused_gdn_products(3)
    try:
        y = ops.conv2d_f16k(x16, wp, b.to(DEV), d, want_nchw=True, gdn=(gp, inverse))
        assert_close(y, ref, "conv+gdn fused (bf16 hi/lo split product), nchw out", rtol=5e-5)
    finally:
        ops.set_fused_gdn_products(1)
    # the default: ONE bf16 product gamma^ x x^2 -- within the rounding of the bf16 store that follows
    y1 = ops.conv2d_f16k(x16, wp, b.to(DEV), d, want_nchw=True, gdn=(gp, inverse))
    e1 = assert_close(y1, ref, "conv+gdn fused (one bf16 product), nchw out", rtol=2.0 ** -9)
    print(f"fused GDN, one bf16 product: relative error {e1:.2e} (three-product split: {float((y.cpu() - ref).abs().max() / ref.abs().max()):.2e})")
    yb = ops.f16k_to_nchw(ops.conv2d_f16k(x16, wp, b.to(DEV), d, gdn=(gp, inverse)), B, C, d.Ho, d.Wo)
    assert_close(yb, ref, "conv+gdn fused, f16k out", rtol=2.0 ** -8 + 2.0 ** -9)
    # the standalone F16K-output GDN kernel (first layer of the chains)
    t = _rand(B, C, H, W, seed=9, scale=3.0)
    g16 = ops.gdn_f16k(t.to(DEV), beta.to(DEV), gamma.to(DEV), inverse=inverse)
    assert_close(ops.f16k_to_nchw(g16, B, C, H, W), O.gdn(t, beta, gamma, inverse=inverse), "gdn -> f16k", rtol=2.0 ** -8)


@pytest.mark.parametrize("B,H,W,ctot,coff,inverse", [(1, 64, 64, 3, 0, False), (2, 40, 72, 6, 3, True), (1, 37, 51, 3, 0, False)])
def test_first_analysis_layer_fused(B, H, W, ctot, coff, inverse):
    """g_a_conv1 + g_a_gdn1 in one persistent kernel (Conv2d(3->128, k5, s2) + GDN -> F16K), odd sizes and a channel view."""
    ops = _ops()
    from masic_amd import synth
    x = _rand(B, ctot, H, W, seed=1)
    w = _rand(128, 3, 5, 5, seed=2, scale=75 ** -0.5)
    b = _rand(128, seed=3, scale=0.1)
    rs = np.random.RandomState(3)
    beta = synth.synth_tensor("g.beta", (128,), rs)
    gamma = synth.synth_tensor("g.gamma", (128, 128), rs)
    q = lambda t: t.bfloat16().float()
    ref = O.gdn(F.conv2d(q(x[:, coff:coff + 3]), q(w), b, stride=2, padding=2), beta, gamma, inverse=inverse)
    gp = ops.pack_gdn_f16k(beta.to(DEV), gamma.to(DEV))
    y16, Ho, Wo = ops.conv_a_gdn_f16k(x.to(DEV), ops.pack_conv_a_weight(w.to(DEV)), b.to(DEV), (gp, inverse), in_coff=coff)
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    assert_close(ops.f16k_to_nchw(y16, B, 128, Ho, Wo), ref, "conv_a + gdn -> f16k", rtol=2.0 ** -8 + 2.0 ** -9)


@pytest.mark.parametrize("B,H,W,C", [(1, 16, 24, 3), (2, 37, 50, 3), (1, 8, 8, 8)])
def test_last_synthesis_layer_depth_to_space(B, H, W, C):
    """g_s_conv4 = ConvTranspose2d(128 -> 3, k5, s2) as a 3x3 convolution to 4C channels with a depth-to-space store."""
    ops = _ops()
    from masic_amd._lib import PREC_BF16
    x = _rand(B, 128, H, W, seed=1)
    w = _rand(128, C, 5, 5, seed=2, scale=(4.0 / (128 * 25)) ** 0.5)
    b = _rand(C, seed=3)
    q = lambda t: t.bfloat16().float()
    ref = F.conv_transpose2d(q(x), q(w), b, stride=2, padding=2, output_padding=1)
    wc, bc = ops.deconv_s2_as_conv_weight(w.to(DEV), b.to(DEV))
    d = ops.make_conv_desc(B, 128, H, W, 32, 3, 3, 1, 1, prec=PREC_BF16)
    out = torch.full((B, C + 2, 2 * H, 2 * W), 5.0, device=DEV)
    ops.conv2d_f16k_d2s(ops.nchw_to_f16k(x.to(DEV)), ops.pack_conv_f16k_weight(wc, d), bc, d, C, out=out, out_coff=1)
    assert_close(out[:, 1:1 + C], ref, "deconv as conv + depth-to-space", rtol=2e-5)
    assert torch.all(out[:, 0] == 5.0) and torch.all(out[:, 1 + C:] == 5.0)


def test_entropy_bottleneck_compress_decompress_vs_reference_streams():
    """EntropyBottleneck.compress: symbols by the HIP kernel, rANS on the host -- byte-identical to the streams the reference
    produced for the same parameters and input (tests/golden/rans_vectors.npz); decompress returns the reference's tensor."""
    import os
    from compressai.entropy_models import EntropyBottleneck
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rans_vectors.npz"))
    eb = EntropyBottleneck(12)
    sd = {k[len("eb_state/"):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("eb_state/")}
    for k in ("_offset", "_quantized_cdf", "_cdf_length"):
        sd.pop(k)
    eb.load_state_dict(sd, strict=False)
    eb = eb.to(DEV)
    with pytest.raises(ValueError):
        eb.compress(torch.from_numpy(G["eb_x"]).to(DEV))              # tables not built yet (reference :144-146)
    eb.update()
    x = torch.from_numpy(G["eb_x"]).to(DEV)
    strings = eb.compress(x)
    assert [bytes(s) for s in strings] == [G["eb_string_0"].tobytes(), G["eb_string_1"].tobytes()]
    xh = torch.cat([eb.decompress([s], x.shape[-2:]) for s in strings])
    assert torch.equal(xh.cpu(), torch.from_numpy(G["eb_xhat"]))
    with pytest.raises(ValueError):
        eb.decompress(strings, x.shape[-2:])                          # the reference's one-stream-at-a-time shape check (:221-224)


@pytest.mark.parametrize("transposed,where", [(False, "out"), (True, "in"), (False, "none")])
def test_conv5s1_pair_with_fused_gdn3(transposed, where):
    """encoder2.pre_conv + pre_gdn / decoder2.after_gdn + cat + after_conv as one launch on two 3-channel sources."""
    ops = _ops()
    from masic_amd import synth
    B, H, W = 2, 37, 53
    xa, xb = _rand(B, 3, H, W, seed=1), _rand(B, 3, H, W, seed=2)
    w = _rand(*((6, 3, 5, 5) if transposed else (3, 6, 5, 5)), seed=3, scale=0.1)
    b = _rand(3, seed=4, scale=0.1)
    rs = np.random.RandomState(5)
    beta, gamma = synth.synth_tensor("g.beta", (3,), rs), synth.synth_tensor("g.gamma", (3, 3), rs)
    inv = transposed
    src = O.gdn(xa, beta, gamma, inverse=inv) if where == "in" else xa
    ref = _oracle_conv(torch.cat((src, xb), 1), w, b, 5, 1, transposed, False, 0, 0)
    if where == "out":
        ref = O.gdn(ref, beta, gamma, inverse=inv)
    d = ops.make_conv_desc(B, 6, H, W, 3, 5, 5, 1, 2, transposed=transposed)
    g = (beta.to(DEV), gamma.to(DEV), inv)
    y = ops.conv5s1_pair(xa.to(DEV), xb.to(DEV), ops.pack_conv_weight(w.to(DEV), d), b.to(DEV),
                         gdn_in=g if where == "in" else None, gdn_out=g if where == "out" else None)
    assert_close(y, ref, f"conv5s1_pair gdn {where}")


@pytest.mark.parametrize("B,Cin,Cmid,Cout,H,W,tr", [(2, 768, 1152, 960, 16, 24, True), (1, 960, 1152, 768, 8, 12, False), (3, 64, 96, 32, 5, 7, False)])
def test_gemm_f16k_dma_stack(B, Cin, Cmid, Cout, H, W, tr):
    """The DMA-staged GEMM of the head layers (conv_f16k.hip: gemm_f16k, both chunk sizes): two chained layers against the
    float32 oracle on bf16-rounded operands, and bit-identical to the register-streamed kernel."""
    ops = _ops()
    x = _rand(B, Cin, H, W, seed=1, scale=2.0)
    w0 = _rand(*((Cin, Cmid) if tr else (Cmid, Cin)), seed=2, scale=(2.0 / Cin) ** 0.5)
    b0 = _rand(Cmid, seed=3, scale=0.1)
    w1 = _rand(Cout, Cmid, seed=4, scale=(2.0 / Cmid) ** 0.5)
    b1 = _rand(Cout, seed=5, scale=0.1)
    q = lambda t: t.bfloat16().float()
    w0c = (q(w0).t() if tr else q(w0)).reshape(Cmid, Cin, 1, 1)
    mid = F.leaky_relu(F.conv2d(q(x), w0c, b0))
    ref = F.relu(F.conv2d(q(mid), q(w1).reshape(Cout, Cmid, 1, 1), b1))
    xf = ops.nchw_to_f16k(x.to(DEV))
    t = ops.gemm_f16k(xf, ops.pack_gemm_f16k_weight(w0.to(DEV), Cin, Cmid, tr), b0.to(DEV), B, Cin, Cmid, H, W, ops.ACT_LEAKY)
    y = ops.gemm_f16k(t, ops.pack_gemm_f16k_weight(w1.to(DEV), Cmid, Cout, False), b1.to(DEV), B, Cmid, Cout, H, W, ops.ACT_RELU, want_nchw=True)
    assert_close(y, ref, "gemm_f16k stack", rtol=1e-3)
    t0 = ops.gemm1x1_bf16(xf, ops.pack_gemm1x1_weight(w0.to(DEV), Cin, Cmid, tr), b0.to(DEV), B, Cin, Cmid, H, W, ops.ACT_LEAKY)
    y0 = ops.gemm1x1_bf16(t0, ops.pack_gemm1x1_weight(w1.to(DEV), Cmid, Cout, False), b1.to(DEV), B, Cmid, Cout, H, W, ops.ACT_RELU, want_nchw=True)
    assert torch.equal(y, y0), "same k order, same rounding: the two GEMM kernels must agree bit for bit"
    out = torch.full((B, Cout + 40, H, W), 3.0, device=DEV)
    ops.gemm_f16k(t, ops.pack_gemm_f16k_weight(w1.to(DEV), Cmid, Cout, False), b1.to(DEV), B, Cmid, Cout, H, W, ops.ACT_RELU, out_nchw=out, out_coff=8)
    assert torch.equal(out[:, 8:8 + Cout], y) and torch.all(out[:, :8] == 3.0) and torch.all(out[:, 8 + Cout:] == 3.0)


@pytest.mark.parametrize("B,H,W,f8", [(2, 16, 24, False), (1, 13, 9, False), (2, 16, 24, True)])
def test_gemm_f16k_group_equals_single_layer_calls(B, H, W, f8):
    """masic_gemm_f16k_group_fwd -- layer i of the three entropy-parameter stacks of a GMM head (reference MASIC.py:330-468) in ONE
    launch -- against the single-layer launches: same kernel, same k order, so every output (F16K, F8K, float32 NCHW; ragged pixel
    counts; different Cin / Cout / activation per group) must agree bit for bit.  Also the argument checks of the grouped entry."""
    ops = _ops()
    shapes = [(768, 1152, ops.ACT_RELU, "f16k"), (960, 768, ops.ACT_LEAKY, "nchw"), (768, 960, ops.ACT_NONE, "f8k" if f8 else "f16k")]
    layers, singles = [], []
    for k, (Cin, Cout, act, out) in enumerate(shapes):
        x = _rand(B, Cin, H, W, seed=10 + k, scale=1.0).to(DEV)
        w = _rand(Cout, Cin, seed=20 + k, scale=(2.0 / Cin) ** 0.5).to(DEV)
        b = _rand(Cout, seed=30 + k, scale=0.1).to(DEV)
        if f8:
            x8 = ops.nchw_to_f8k(x, 0.02)
            wp, ws = ops.pack_gemm_f8k_weight(w, Cin, Cout, False)
            ws = (ws * 0.02).contiguous()
            layers.append(dict(x=x8, wp=wp, ws=ws, bias=b, Cin=Cin, Cout=Cout, act=act, out=out, out_scale=0.05))
            singles.append(ops.gemm_f8k(x8, wp, ws, b, B, Cin, Cout, H, W, act, out=out, out_scale=0.05))
        else:
            xf = ops.nchw_to_f16k(x)
            wp = ops.pack_gemm_f16k_weight(w, Cin, Cout, False)
            layers.append(dict(x=xf, wp=wp, bias=b, Cin=Cin, Cout=Cout, act=act, out=out))
            singles.append(ops.gemm_f16k(xf, wp, b, B, Cin, Cout, H, W, act, want_nchw=out == "nchw"))
    for n in (3, 2, 1):
        got = ops.gemm_f16k_group(layers[:n], B, H, W)
        for k in range(n):
            assert got[k].dtype == singles[k].dtype and torch.equal(got[k], singles[k]), f"group of {n}, layer {k} ({shapes[k]})"
    if not f8:
        with pytest.raises(RuntimeError):
            ops.gemm_f16k_group(layers + layers[:1], B, H, W)                       # more than three groups
        with pytest.raises(RuntimeError):
            ops.gemm_f16k_group([dict(layers[0], Cout=1000)], B, H, W)              # Cout % 32


def test_elementwise_binary_op_without_second_operand_is_an_error_not_a_fault():
    from masic_amd import _lib, ops
    x = torch.ones(64, device=DEV)
    y = torch.empty_like(x)
    for op in (ops.EW_ACT_BWD, ops.EW_MUL, ops.EW_ADD, ops.EW_DIFF_SCALE):
        assert _lib.lib.masic_elementwise(ops._p(x), None, ops._p(y), 64, int(op), 0.0, 0.0, ops._stream()) != 0
    assert _lib.lib.masic_elementwise(ops._p(x), None, ops._p(y), 64, int(ops.EW_SQUARE), 0.0, 0.0, ops._stream()) == 0


def test_homography_from_corners_vs_restatement():
    """corner offsets -> h_matrix (get_perspective_transform + inverse + h_adjust) in one kernel against the
    float64 restatement (oracle/udh_oracle.py; kornia absent: parity unpinned), plus the defining property: before h_adjust
    the inverse maps corners + delta back onto the corners."""
    ops = _ops()
    from oracle import udh_oracle as U
    rs = np.random.RandomState(7)
    B = 67
    base = np.array([[32, 32], [160, 32], [160, 160], [32, 160]], dtype=np.float32)
    corners = np.repeat(base[None], B, 0) + rs.randint(-8, 9, (B, 1, 2)).astype(np.float32)
    delta = rs.uniform(-16, 16, (B, 4, 2)).astype(np.float32)
    for ori, patch in (((512, 512), (128, 128)), ((512, 896), (128, 128)), ((128, 128), (128, 128))):
        got = ops.homography_from_corners(torch.from_numpy(corners).to(DEV), torch.from_numpy(delta).to(DEV), ori, patch).cpu().numpy()
        want = U.h_matrix_from_corners(corners, delta, ori, patch)
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), (ori, np.abs(got - want).max())
    hat = np.concatenate([corners + delta, np.ones((B, 4, 1), np.float32)], axis=2).astype(np.float64)      # a = b = 1: h_adjust is the identity
    back = np.einsum("bij,bkj->bki", got.astype(np.float64), hat)
    assert np.abs(back[..., :2] / back[..., 2:] - corners).max() <= 1e-3
    bad = ops.homography_from_corners(torch.zeros(1, 4, 2, device=DEV), torch.zeros(1, 4, 2, device=DEV), (8, 8), (8, 8))
    assert torch.isnan(bad).all()                                       # degenerate quadrilateral: NaN, not garbage


@pytest.mark.parametrize("H,W,start,crop,xy", [(600, 700, (31, 57), (512, 512), (45, 83)),      # 2x decimation: OpenCV's INTER_AREA case
                                              (400, 640, (3, 64), (384, 512), (0, 128)),        # scales 1.5 / 2.0: fixed-point bilinear
                                              (300, 260, (0, 0), (200, 260), (100, 17)),        # upscaling rows, scale ~1.016 columns
                                              (256, 256, (0, 0), (256, 256), (64, 64))])         # identity resize
def test_pair_prep_vs_restatement(H, W, start, crop, xy):
    """crop + ToTensor and resize + normalise + grey + patch of one view in two launches, against the numpy
    restatement of the reference's cv2 / torchvision chain (oracle/data_oracle.py; cv2 absent: parity unpinned) -- equal bit for
    bit: the resize is integer arithmetic, the float32 steps are done in the same order."""
    ops = _ops()
    from oracle import data_oracle as D
    rs = np.random.RandomState(H + W)
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    img[:40, :60] = 255                                               # saturated / flat regions as well
    img[-30:, -50:] = 0
    pic, patch = ops.pair_prep(torch.from_numpy(img).to(DEV), start, crop, 256, xy, 128)
    want_pic, want_patch = D.prepare_view(img, start[0], start[1], crop[0], crop[1], 256, xy[0], xy[1], 128)
    assert np.array_equal(pic.cpu().numpy(), want_pic)
    assert np.array_equal(patch.cpu().numpy(), want_patch)
    with pytest.raises(RuntimeError):
        ops.pair_prep(torch.from_numpy(img).to(DEV), (H - 10, 0), crop)     # crop outside the picture


def test_api_surface_gaussian_conditional_and_gdn1_vs_reference():
    """Classes of the compressai surface that MASIC itself does not call: GaussianConditional.forward / build_indexes (reference
    entropy_models.py:527-561) and GDN1 (layers/gdn.py:95-121) against the reference's outputs (tests/golden/misc_api.npz)."""
    from compressai.entropy_models import GaussianConditional
    from compressai.layers import GDN1
    fx = load_npz("misc_api.npz")
    gc = GaussianConditional([float(v) for v in fx["gc/scale_table"]]).to(DEV).eval()
    x, sc, mu = (torch.from_numpy(fx["gc/" + k]).to(DEV) for k in ("x", "scales", "means"))
    with torch.no_grad():
        y0, l0 = gc(x, sc)
        y1, l1 = gc(x, sc, mu)
        idx = gc.build_indexes(sc)
    assert torch.equal(y0.cpu(), torch.from_numpy(fx["gc/y_nomeans"]))
    assert_close(y1, torch.from_numpy(fx["gc/y_means"]), "GaussianConditional: dequantised about the means", 1e-6)
    for got, key in ((l0, "gc/lik_nomeans"), (l1, "gc/lik_means")):
        ref = torch.from_numpy(fx[key])
        assert float(((got.cpu() - ref).abs() / (ref.abs() + 1e-9)).max()) <= 1e-3 and float((got.cpu() - ref).abs().max()) <= 1e-6, key
    assert idx.cpu().numpy().tolist() == fx["gc/indexes"].tolist()
    gc.train()
    with pytest.raises(NotImplementedError):
        gc(x.clone().requires_grad_(True), sc)
    for tag, inverse in (("gdn1/", False), ("gdn1_inv/", True)):
        m = GDN1(6, inverse=inverse).to(DEV)
        with torch.no_grad():
            m.beta.copy_(torch.from_numpy(fx[tag + "beta"]))
            m.gamma.copy_(torch.from_numpy(fx[tag + "gamma"]))
            out = m(torch.from_numpy(fx[tag + "x"]).to(DEV))
        assert_close(out, torch.from_numpy(fx[tag + "y"]), "GDN1 " + tag, 1e-5)
