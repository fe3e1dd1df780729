"""conv_f16k vs the float32-NCHW-input bf16 kernel (same operand rounding) on the layer shapes of the codec; timing."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from masic_amd import ops, _lib
torch.manual_seed(0)
dev = "cuda"
def run(name, B, Cin, H, W, Cout, k, s, transposed=False, masked=False, act=ops.ACT_NONE, reps=0):
    pad = k // 2
    x = torch.randn(B, Cin, H, W, device=dev)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = torch.randn(wshape, device=dev) / (Cin * k * k) ** 0.5
    if masked:
        w[:, :, k // 2, k // 2:] = 0; w[:, :, k // 2 + 1:] = 0
    bias = torch.randn(Cout, device=dev)
    d0 = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, masked=masked, act=act, prec=_lib.PREC_BF16)
    ref = ops._conv2d(x, ops.pack_conv_weight(w, d0), bias, d0)
    cin16 = (Cin + 15) // 16 * 16
    d1 = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, masked=masked, act=act, in_ctot=cin16, prec=_lib.PREC_BF16)
    if not ops.conv_f16k_supported(d1):
        print(name, "unsupported"); return
    x16 = ops.nchw_to_f16k(x)
    wp = ops.pack_conv_f16k_weight(w, d1)
    y = ops.conv2d_f16k(x16, wp, bias, d1, want_nchw=True)
    cout16 = (Cout + 15) // 16 * 16
    d2 = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, masked=masked, act=act, in_ctot=cin16, out_ctot=cout16, prec=_lib.PREC_BF16)
    y16 = ops.conv2d_f16k(x16, wp, bias, d2)
    torch.cuda.synchronize()
    yb = ops.f16k_to_nchw(y16, B, Cout, d2.Ho, d2.Wo)
    scale = ref.abs().max().item()
    e1 = (y - ref).abs().max().item() / scale
    e2 = (yb - ref).abs().max().item() / scale
    msg = f"{name:28s} nchw-out err {e1:.2e}  f16k-out err {e2:.2e}"
    if reps:
        for fn, tag in ((lambda: ops._conv2d(x, ops.pack_conv_weight, bias, d0), None),):
            pass
        pk0 = ops.pack_conv_weight(w, d0)
        for tag, fn in (("old", lambda: ops._conv2d(x, pk0, bias, d0)), ("f16k->nchw", lambda: ops.conv2d_f16k(x16, wp, bias, d1, want_nchw=True)),
                        ("f16k->f16k", lambda: ops.conv2d_f16k(x16, wp, bias, d2))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            flops = 2.0 * B * Cin * Cout * k * k * (H * W if transposed else d0.Ho * d0.Wo)
            msg += f" | {tag} {dt * 1e6:7.1f} us {flops / dt / 1e12:6.1f} TF"
    print(msg, flush=True)

def run_gdn(name, B, Cin, H, W, k, s, transposed=False, inverse=False, reps=0):
    Cout, pad = 128, k // 2
    x = torch.randn(B, Cin, H, W, device=dev)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = torch.randn(wshape, device=dev) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, device=dev)
    beta = torch.rand(Cout, device=dev) + 0.5
    gamma = (torch.rand(Cout, Cout, device=dev) * 0.2).contiguous()
    d1 = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, in_ctot=Cin, prec=_lib.PREC_BF16)
    d2 = ops.make_conv_desc(B, Cin, H, W, Cout, k, k, s, pad, transposed=transposed, in_ctot=Cin, out_ctot=Cout, prec=_lib.PREC_BF16)
    x16 = ops.nchw_to_f16k(x)
    wp = ops.pack_conv_f16k_weight(w, d1)
    gp = ops.pack_gdn_f16k(beta, gamma)
    t = ops.conv2d_f16k(x16, wp, bias, d1, want_nchw=True)
    ref = ops.gdn(t, beta, gamma, inverse=inverse)                       # exact float32 GDN kernel
    y = ops.conv2d_f16k(x16, wp, bias, d1, want_nchw=True, gdn=(gp, inverse))
    y16 = ops.conv2d_f16k(x16, wp, bias, d2, gdn=(gp, inverse))
    torch.cuda.synchronize()
    yb = ops.f16k_to_nchw(y16, B, Cout, d2.Ho, d2.Wo)
    scale = ref.abs().max().item()
    msg = f"{name:28s} fused-gdn nchw err {(y - ref).abs().max().item() / scale:.2e}  f16k err {(yb - ref).abs().max().item() / scale:.2e}"
    if reps:
        for tag, fn in (("conv->nchw + gdn_f16k", lambda: ops.gdn_f16k(ops.conv2d_f16k(x16, wp, bias, d1, want_nchw=True), beta, gamma, inverse=inverse)),
                        ("fused", lambda: ops.conv2d_f16k(x16, wp, bias, d2, gdn=(gp, inverse)))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); msg += f" | {tag} {(time.perf_counter() - t0) / reps * 1e6:7.1f} us"
    print(msg, flush=True)

big = len(sys.argv) > 1 and sys.argv[1] == "big"
if len(sys.argv) > 1 and sys.argv[1] == "hyper":
    run("h_a1 conv5s1 192->128 32^2", 8, 192, 32, 32, 128, 5, 1, act=ops.ACT_RELU, reps=20)
    run("h_a2 conv5s2 128->128 32^2", 8, 128, 32, 32, 128, 5, 2, act=ops.ACT_RELU, reps=20)
    run("h_a3 conv5s2 128->128 16^2", 8, 128, 16, 16, 128, 5, 2, reps=20)
    run("h_s1 deconv 128->192 8^2", 8, 128, 8, 8, 192, 5, 2, transposed=True, act=ops.ACT_LEAKY, reps=20)
    run("h_s2 deconv 192->288 16^2", 8, 192, 16, 16, 288, 5, 2, transposed=True, act=ops.ACT_LEAKY, reps=20)
    run("h_s3 conv3 288->384 32^2", 8, 288, 32, 32, 384, 3, 1, reps=20)
    run("ctx masked5 192->384 32^2", 8, 192, 32, 32, 384, 5, 1, masked=True, reps=20)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "ablate":      # timing only (results of ablated builds are garbage)
    run("B  conv5s2 128->128 256^2", 8, 128, 256, 256, 128, 5, 2, reps=10)
    run("B' deconv 128->128 128^2", 8, 128, 128, 128, 128, 5, 2, transposed=True, reps=10)
    run_gdn("B  conv+gdn 256^2", 8, 128, 256, 256, 5, 2, reps=10)
    run_gdn("C  conv+gdn 128^2", 8, 128, 128, 128, 5, 2, reps=10)
    run_gdn("B' deconv+igdn 128^2", 8, 128, 128, 128, 5, 2, transposed=True, inverse=True, reps=10)
    sys.exit(0)
run("conv5s2 128->128 64x64", 2, 128, 64, 64, 128, 5, 2)
run("conv5s2 128->192 40x24", 1, 128, 40, 24, 192, 5, 2, act=ops.ACT_RELU)
run("deconv5s2 128->128 32x32", 2, 128, 32, 32, 128, 5, 2, transposed=True)
run("deconv5s2 192->128 9x13", 1, 192, 9, 13, 128, 5, 2, transposed=True, act=ops.ACT_LEAKY)
run("conv3s1 192->128 32x32", 2, 192, 32, 32, 128, 3, 1)
run("masked5 192->384 32x32", 1, 192, 32, 32, 384, 5, 1, masked=True)
run("conv3s1 288->384 16x16", 1, 288, 16, 16, 384, 3, 1)
run_gdn("gdn conv5s2 64x64", 2, 128, 64, 64, 5, 2)
run_gdn("igdn deconv 192 24x40", 1, 192, 24, 40, 5, 2, transposed=True, inverse=True)
if big:
    run_gdn("B  conv+gdn 256^2", 8, 128, 256, 256, 5, 2, reps=10)
    run_gdn("B' deconv+igdn 128^2", 8, 128, 128, 128, 5, 2, transposed=True, inverse=True, reps=10)
    run("B  conv5s2 128->128 256^2", 8, 128, 256, 256, 128, 5, 2, reps=10)
    run("C  conv5s2 128->128 128^2", 8, 128, 128, 128, 128, 5, 2, reps=10)
    run("D  conv5s2 128->192 64^2", 8, 128, 64, 64, 192, 5, 2, reps=10)
    run("D' deconv 192->128 32^2", 8, 192, 32, 32, 128, 5, 2, transposed=True, reps=10)
    run("C' deconv 128->128 64^2", 8, 128, 64, 64, 128, 5, 2, transposed=True, reps=10)
    run("B' deconv 128->128 128^2", 8, 128, 128, 128, 128, 5, 2, transposed=True, reps=10)
    run("ctx masked5 192->384 32^2", 8, 192, 32, 32, 384, 5, 1, masked=True, reps=10)
    run("h_a conv3 192->128 32^2", 8, 192, 32, 32, 128, 3, 1, reps=10)
