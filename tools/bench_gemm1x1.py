import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from masic_amd import ops
torch.manual_seed(0)
B, H, W = 8, 32, 32
for cin, cout in ((768, 1152), (1152, 768), (768, 960), (960, 1152), (1152, 960), (960, 960)):
    x = torch.randn(B, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    b = torch.randn(cout, device="cuda")
    xf = ops.nchw_to_f16k(x)
    wp = ops.pack_gemm1x1_weight(w, cin, cout, False)
    for _ in range(3): ops.gemm1x1_bf16(xf, wp, b, B, cin, cout, H, W, ops.ACT_RELU)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.gemm1x1_bf16(xf, wp, b, B, cin, cout, H, W, ops.ACT_RELU)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    ref = ops.gemm1x1_bf16(xf, wp, b, B, cin, cout, H, W, ops.ACT_RELU, want_nchw=True)
    wp2 = ops.pack_gemm_f16k_weight(w, cin, cout, False)
    y2 = ops.gemm_f16k(xf, wp2, b, B, cin, cout, H, W, ops.ACT_RELU, want_nchw=True)
    err = float((y2 - ref).abs().max() / ref.abs().max())
    y16a = ops.gemm1x1_bf16(xf, wp, b, B, cin, cout, H, W, ops.ACT_RELU); y16b = ops.gemm_f16k(xf, wp2, b, B, cin, cout, H, W, ops.ACT_RELU)
    same16 = bool((y16a == y16b).float().mean() > 0.999)
    for _ in range(3): ops.gemm_f16k(xf, wp2, b, B, cin, cout, H, W, ops.ACT_RELU)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.gemm_f16k(xf, wp2, b, B, cin, cout, H, W, ops.ACT_RELU)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 50
    print(f"{cin}->{cout}: register-streamed {dt * 1e6:6.1f} us {2.0 * B * H * W * cin * cout / dt / 1e12:6.1f} TF | dma {dt2 * 1e6:6.1f} us {2.0 * B * H * W * cin * cout / dt2 / 1e12:6.1f} TF | rel diff {err:.1e} f16k same {same16}")
