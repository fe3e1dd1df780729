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
    print(f"{cin}->{cout}: {dt * 1e6:6.1f} us  {2.0 * B * H * W * cin * cout / dt / 1e12:6.1f} TF")
