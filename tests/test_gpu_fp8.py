"""GPU tests (-m gpu) of the fp8 (OCP e4m3fn) operand path -- BASELINE.json configs[4], masic_amd/fp8.py.

There is no reference for reduced precision (the reference computes in float32), so this file (i) proves the kernels exact
where fp8 arithmetic is exact -- operands that e4m3 represents exactly give the float32 oracle's convolution bit for bit up to
the final dequantisation multiply, which pins the MFMA operand layout, the slab pairing and the scale handling --, (ii) checks
every quantising store against torch's own float8_e4m3fn cast, and (iii) holds the whole forward to a DECLARED budget
against the oracle (the numbers are in BUDGET below and in DESIGN.md)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hsic_oracle as O
from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"

# fp8-operand forward vs the CPU oracle, HSIC(128,192,5).  The budget the mode is HELD TO is declared where it means something -- at a trained
# operating point, tests/test_gpu_convergence.py::TRAINED_BUDGET["fp8"]: rate within 1 %, PSNR within 0.15 dB, <= 2 % of the int32 symbols off
# by at most one (measured: -0.08 % / -0.02 dB / 0.45 %).  The numbers below are the STRESS point of the parity tests -- synthetic random-gain
# weights, latents spanning +-20, reconstructions at 5 dB, where a symbol sits within fp8 rounding of a boundary far more often (measured
# 27 % of the symbols, by up to 2); they bound the kernels' behaviour there, they are not what the mode costs a codec.
BUDGET = {"bpp_rel": 0.01, "psnr_db": 0.05, "symbol_mismatch": 0.35, "symbol_max_abs": 3}


def _exact(shape, values, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.tensor(values, dtype=torch.float32)
    return v[torch.randint(0, len(values), shape, generator=g)]


def _fp8_round(t, scale):
    """what a quantising store must produce, decoded: torch's e4m3fn cast (round to nearest even, saturating by the clamp)"""
    return (t / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale


@pytest.mark.parametrize("transposed,B,H,W", [(False, 2, 64, 64), (True, 2, 16, 32), (False, 1, 40, 72), (True, 1, 24, 24)])
def test_conv_f8k_exact_on_representable_operands(transposed, B, H, W):
    """128 -> 128, 5x5, stride 2 (the strided form pairs taps per MFMA, the transposed form channel blocks): inputs and weights
    drawn from values e4m3 holds exactly, so the fp8 contraction equals the float32 one; outputs float32 / bf16 / fp8."""
    from masic_amd import nn as mnn, ops
    from masic_amd._lib import PREC_FP8
    C = 128
    x = _exact((B, C, H, W), [0.0, 0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -3.0], 1)
    w = _exact((C, C, 5, 5), [0.0, 0.0, 0.0, 0.5, -0.5, 1.0, -1.0, 0.25], 2)
    w[:, :, 0, 0] = 2.0              # every output channel's largest magnitude is 2: w / (2 / 448) is exact in e4m3
    if transposed:
        w = w.permute(1, 0, 2, 3).contiguous()      # ConvTranspose2d layout [Cin, Cout, kh, kw]; max over (ci, taps) per co is still 2
    bias = 0.25 * torch.randn(C, generator=torch.Generator().manual_seed(3))
    ref = (F.conv_transpose2d(x, w, bias, stride=2, padding=2, output_padding=1) if transposed else F.conv2d(x, w, bias, stride=2, padding=2))
    mod = (mnn.ConvTranspose2d(C, C, 5, stride=2, padding=2, output_padding=1) if transposed else mnn.Conv2d(C, C, 5, stride=2, padding=2)).to(DEV)
    with torch.no_grad():
        mod.weight.copy_(w)
        mod.bias.copy_(bias)
    assert mod.f8k_supported(B, H, W)
    in_scale = 0.5                   # x / 0.5 in {0, +-1, ..., +-8}: exact; exercises the input-scale factor of the dequantisation
    x8 = ops.nchw_to_f8k(x.to(DEV), in_scale)
    assert torch.equal(ops.f8k_to_nchw(x8, B, C, H, W, in_scale).cpu(), x)
    y32, Ho, Wo = mod.run_f8k(x8, in_scale, B, H, W, out="nchw")
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    e = assert_close(y32, ref, "conv_f8k -> float32", 2e-7)        # the only rounding: accumulator x (weight scale x input scale)
    y16, _, _ = mod.run_f8k(x8, in_scale, B, H, W, out="f16k")
    assert_close(ops.f16k_to_nchw(y16, B, C, Ho, Wo), ref.bfloat16().float(), "conv_f8k -> F16K", 2e-7 + 2 ** -8)
    out_scale = float(ref.abs().max()) / 448.0
    y8, _, _ = mod.run_f8k(x8, in_scale, B, H, W, out="f8k", out_scale=out_scale)
    got, want = ops.f8k_to_nchw(y8, B, C, Ho, Wo, out_scale).cpu(), _fp8_round(ref, out_scale)
    off = (got != want)
    # a value that the 1e-7 dequantisation rounding moves across an fp8 rounding boundary may land on the neighbouring code
    assert float(off.float().mean()) < 1e-3 and float((got - want).abs().max()) <= float(ref.abs().max()) * 2 ** -3, float(off.float().mean())
    print(f"conv_f8k exact data (transposed={transposed}): float32 out rel err {e:.1e}, fp8 out: {int(off.sum())}/{off.numel()} codes differ by one step")


def test_conv_f8k_fused_gdn_and_bf16_producer():
    """fp8 operands + the fused (inverse) GDN epilogue, and a bf16-operand producer writing F8K: against the oracle's GDN of the
    float32 convolution on the same (exactly representable) operands."""
    from compressai.layers import GDN
    from masic_amd import nn as mnn, ops, synth
    C, B, H, W = 128, 2, 32, 32
    x = _exact((B, C, H, W), [0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0], 5)
    w = _exact((C, C, 5, 5), [0.0, 0.0, 0.0, 0.125, -0.125, 0.25, -0.25], 6)
    w[:, :, 0, 0] = 0.5
    rs = np.random.RandomState(7)
    beta, gamma = synth.synth_tensor("g.beta", (C,), rs), synth.synth_tensor("g.gamma", (C, C), rs)
    for inverse in (False, True):
        mod = mnn.Conv2d(C, C, 5, stride=2, padding=2).to(DEV)
        gdn = GDN(C, inverse=inverse).to(DEV)
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.zero_(); gdn.beta.copy_(beta); gdn.gamma.copy_(gamma)
        ref = O.gdn(F.conv2d(x, w, None, stride=2, padding=2), beta, gamma, inverse=inverse)
        x8 = ops.nchw_to_f8k(x.to(DEV), 1.0)
        y16, Ho, Wo = mod.run_f8k(x8, 1.0, B, H, W, out="f16k", gdn=gdn)
        assert_close(ops.f16k_to_nchw(y16, B, C, Ho, Wo), ref, f"conv_f8k + GDN(inverse={inverse}) -> F16K", 2 ** -8 + 1e-3)
        sc = float(ref.abs().max()) / 448.0
        y8, _, _ = mod.run_f8k(x8, 1.0, B, H, W, out="f8k", out_scale=sc, gdn=gdn)
        got = ops.f8k_to_nchw(y8, B, C, Ho, Wo, sc).cpu()
        assert float((got - ref).abs().max()) <= float(ref.abs().max()) * (2 ** -4 + 2e-3)      # half an fp8 step at the top binade
        assert float(((got - ref).abs() > ref.abs() * 2 ** -4 + sc * 2 ** -9 * 1.01 + 2e-3 * ref.abs().max()).float().mean()) == 0.0
        # bf16-operand producer with an F8K output == the F16K result pushed through the fp8 cast
        x16 = ops.nchw_to_f16k(x.to(DEV))
        ref16, _, _ = mod.run_f16k(x16, B, H, W, gdn=gdn)
        y8b, _, _ = mod.run_f16k_f8out(x16, B, H, W, sc, gdn=gdn)
        a, b = ops.f8k_to_nchw(y8b, B, C, Ho, Wo, sc).cpu(), ops.f16k_to_nchw(ref16, B, C, Ho, Wo).cpu()
        assert float((a - b).abs().max()) <= float(b.abs().max()) * (2 ** -4 + 2 ** -8)


@pytest.mark.parametrize("Cin,Cmid,Cout,tr", [(768, 1152, 768, True), (960, 1152, 960, False)])
def test_gemm_f8k_stack_exact_then_bounded(Cin, Cmid, Cout, tr):
    """The first two layers of an entropy-parameter stack with fp8 operands (F8K -> F8K -> F16K): exact on representable
    operands for the first layer; the pair against float32 within the fp8 quantisation of the intermediate."""
    from masic_amd import ops
    B, H, W = 2, 16, 24
    x = _exact((B, Cin, H, W), [0.0, 0.0, 0.5, -0.5, 1.0, -1.0, 2.0], 11)
    w0 = _exact((Cmid, Cin), [0.0, 0.0, 0.0, 0.0, 0.125, -0.125, 0.25, -0.25], 12)
    w0[:, 0] = 0.5
    b0 = 0.1 * torch.randn(Cmid, generator=torch.Generator().manual_seed(13))
    ref0 = F.leaky_relu(torch.einsum("oc,bchw->bohw", w0, x) + b0.view(1, -1, 1, 1), 0.01)
    w0d = (w0.t().contiguous() if tr else w0).to(DEV)
    x8 = ops.nchw_to_f8k(x.to(DEV), 1.0)
    wp, ws = ops.pack_gemm_f8k_weight(w0d, Cin, Cmid, tr)
    assert torch.allclose(ws.cpu(), torch.full((Cmid,), 0.5 / 448.0))
    y = ops.gemm_f8k(x8, wp, ws, b0.to(DEV), B, Cin, Cmid, H, W, ops.ACT_LEAKY, out="nchw")
    assert_close(y, ref0, "gemm_f8k -> float32 (exact operands)", 2e-7)
    s1 = float(ref0.abs().max()) * 1.25 / 448.0
    t8 = ops.gemm_f8k(x8, wp, ws, b0.to(DEV), B, Cin, Cmid, H, W, ops.ACT_LEAKY, out="f8k", out_scale=s1)
    mid = ops.f8k_to_nchw(t8, B, Cmid, H, W, s1).cpu()
    assert float((mid != _fp8_round(ref0, s1)).float().mean()) < 1e-3
    w1 = 0.05 * torch.randn(Cout, Cmid, generator=torch.Generator().manual_seed(14))
    b1 = 0.1 * torch.randn(Cout, generator=torch.Generator().manual_seed(15))
    wp1, ws1 = ops.pack_gemm_f8k_weight(w1.to(DEV), Cmid, Cout, False)
    y16 = ops.gemm_f8k(t8, wp1, (ws1 * s1).contiguous(), b1.to(DEV), B, Cmid, Cout, H, W, ops.ACT_NONE, out="f16k")
    got = ops.f16k_to_nchw(y16, B, Cout, H, W).cpu()
    # the kernel's own arithmetic, emulated: fp8 weights (per-channel scale) x the fp8 intermediate, float32 accumulate
    w1q = _fp8_round(w1 / ws1.cpu().view(-1, 1), 1.0) * ws1.cpu().view(-1, 1)
    emu = torch.einsum("oc,bchw->bohw", w1q, mid) + b1.view(1, -1, 1, 1)
    assert_close(got, emu, "gemm_f8k second layer vs its fp8 emulation", 2 ** -8 + 1e-5)
    full = torch.einsum("oc,bchw->bohw", w1, ref0) + b1.view(1, -1, 1, 1)
    print(f"gemm_f8k pair vs float32: relative error {float((got - full).abs().max() / full.abs().max()):.3e} (fp8 quantisation of weights and intermediate)")


def test_quantising_helpers_absmax_and_first_layer():
    from compressai.layers import GDN
    from masic_amd import nn as mnn, ops, synth
    g = torch.Generator().manual_seed(21)
    x = 3.0 * torch.randn(2, 70, 9, 13, generator=g)
    x[0, 3, 2, 5] = 1e4            # saturates
    sc = 0.05
    x8 = ops.nchw_to_f8k(x.to(DEV), sc)
    assert torch.equal(ops.f8k_to_nchw(x8, 2, 70, 9, 13, sc).cpu(), _fp8_round(x, sc))
    assert float(ops.f8k_to_nchw(x8, 2, 96, 9, 13, sc)[:, 70:].abs().max()) == 0.0          # channels padded to 32 are zeros
    assert float(ops.absmax(x.to(DEV))) == float(x.abs().max())
    t16 = ops.nchw_to_f16k(x.to(DEV))
    assert float(ops.absmax(t16)) == float(x.bfloat16().float().abs().max())
    # first analysis layer + GDN with an F8K output == its F16K output pushed through the fp8 cast
    conv, gdn = mnn.Conv2d(3, 128, 5, stride=2, padding=2).to(DEV), GDN(128).to(DEV)
    img = torch.rand(2, 3, 64, 96, generator=g).to(DEV)
    pack = conv.packed_first_layer_weight()
    gp = (mnn.packed_gdn_f16k(gdn), False)
    y16, Ho, Wo = ops.conv_a_gdn_f16k(img, pack, conv.bias.detach(), gp)
    ref = ops.f16k_to_nchw(y16, 2, 128, Ho, Wo).cpu()
    s = float(ref.abs().max()) * 1.5 / 448.0
    y8, _, _ = ops.conv_a_gdn_f8k(img, pack, conv.bias.detach(), gp, s)
    got = ops.f8k_to_nchw(y8, 2, 128, Ho, Wo, s).cpu()
    assert float((got - ref).abs().max()) <= float(ref.abs().max()) * (2 ** -4 + 2 ** -8)


def test_fp8_forward_budget_vs_oracle_and_graph():
    """configs[4] in miniature (one 256x256 pair; the bench reports the same quantities at 512x512): calibrate on one batch,
    run the fp8-operand forward on ANOTHER pair, compare with the oracle: rate, PSNR, int32 symbols within BUDGET; masks (float32
    warp) at 1e-4; deterministic; a HIP-graph replay equals the eager launch; the mode really ran fp8 kernels."""
    import MASIC
    from masic_amd import fp8, nn as mnn, ops, synth
    from masic_amd.graph import GraphedHSIC
    from masic_amd.loss import rate_distortion
    N, M, K = 128, 192, 5
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=100)
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    cal = [tuple(t.to(DEV) for t in synth.synth_inputs(2, 256, 256, seed=300))]
    table = fp8.calibrate(net, cal)
    assert {"a1", "a2"} <= set(table["encoder1"]) and {"d1", "d2"} <= set(table["decoder2"]) and "c" in table["_h_s2_same_resolution"], table.keys()
    x1, x2, hm = synth.synth_inputs(1, 256, 256, seed=301)
    with torch.no_grad():
        ref = O.hsic_forward(sd, x1, x2, hm, K=K, keep=True)
    oc = O.rd_loss(ref, x1, x2, 0.01)
    osym = O.symbols(ref["_aux"], sd)
    x1d, x2d, hmd = x1.to(DEV), x2.to(DEV), hm.to(DEV)
    timer = ops.KernelTimer()
    mnn.set_precision("fp8")
    try:
        with torch.no_grad():
            ops.set_kernel_timer(timer)
            out = net(x1d, x2d, hmd)
            ops.set_kernel_timer(None)
            again = net(x1d, x2d, hmd)
            sym = net.symbol_streams(x1d, x2d, hmd)
            crit = rate_distortion(out, x1d, x2d, 0.01)
            rep = GraphedHSIC(net, x1d, x2d, hmd)(x1d, x2d, hmd)
            torch.cuda.synchronize()
            mnn.set_precision("bf16")
            crit16 = rate_distortion(net(x1d, x2d, hmd), x1d, x2d, 0.01)
    finally:
        ops.set_kernel_timer(None)
        mnn.set_precision("f32")
    for k in ("x1_hat", "x2_hat", "y1_hat"):
        assert torch.equal(out[k], again[k]) and torch.equal(out[k], rep[k]), k
    for k in ("x1_mask_R", "x1_mask_L"):
        assert_close(out[k], ref[k], "fp8:" + k, 1e-4)
    for k, v in out["likelihoods"].items():
        assert float(v.min()) > 0.0 and float(v.max()) <= 1.0 + 1e-6, k
    total = sum(v.numel() for v in osym.values())
    bad = sum(int((sym[k].cpu() != osym[k]).sum()) for k in osym)
    worst = max(int((sym[k].cpu().to(torch.int64) - osym[k].to(torch.int64)).abs().max()) for k in osym)
    d_bpp = float(crit["bpp_loss"]) / float(oc["bpp_loss"]) - 1.0
    print(f"fp8 operands vs oracle: bpp {float(crit['bpp_loss']):.4f} vs {float(oc['bpp_loss']):.4f} ({100 * d_bpp:+.2f} %; bf16 {100 * (float(crit16['bpp_loss']) / float(oc['bpp_loss']) - 1):+.2f} %), "
          f"psnr1 {crit['psnr1']:.3f} vs {oc['psnr1']:.3f} (bf16 {crit16['psnr1']:.3f}), psnr2 {crit['psnr2']:.3f} vs {oc['psnr2']:.3f}; "
          f"{bad}/{total} symbols differ ({100.0 * bad / total:.1f} %), largest difference {worst}")
    assert abs(d_bpp) <= BUDGET["bpp_rel"]
    for k in ("psnr1", "psnr2"):
        assert abs(crit[k] - oc[k]) <= BUDGET["psnr_db"], (k, crit[k], oc[k])
    assert bad <= BUDGET["symbol_mismatch"] * total and worst <= BUDGET["symbol_max_abs"]
    assert any(k.startswith("conv_f16k<") and k.endswith(", true>") for k in timer.summary()), list(timer.summary())      # conv_f16k<..., F8 = true> launches were timed
