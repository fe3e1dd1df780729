"""Times masic_gemm_wgrad_f16k against the float32 NCHW 1x1 weight-gradient kernel at the head-layer shapes (8 x 32 x 32 latents)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from masic_amd import ops, _lib
B, H, W = 8, 32, 32
for Cout, Cin in ((1152, 768), (768, 1152), (960, 768), (960, 960), (1152, 960)):
    x = torch.randn(B, Cin, H, W, device="cuda"); dy = torch.randn(B, Cout, H, W, device="cuda")
    x16, g16 = ops.nchw_to_f16k(x), ops.nchw_to_f16k(dy)
    d = ops.make_conv_desc(B, Cin, H, W, Cout, 1, 1, 1, 0, prec=_lib.PREC_BF16)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    a = t(lambda: ops.gemm_wgrad_f16k(g16, x16, B, Cout, Cin, H * W))
    b = t(lambda: ops.conv2d_wgrad(x, dy, d, (Cout, Cin, 1, 1)))
    gf = 2.0 * B * H * W * Cin * Cout / 1e9
    print(f"{Cout}x{Cin}: f16k {a:6.1f} us ({gf / a:.2f} PFLOP/s)   float32 NCHW {b:6.1f} us")
