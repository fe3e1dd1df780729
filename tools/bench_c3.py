"""C -> C (C = 32 | 64: argv[1]) 3x3 layer at 8 x 512^2 on F16K: conv_f16k configuration vs the resident-weight persistent kernel."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from masic_amd import ops, _lib
torch.manual_seed(0)
dev = "cuda"
B, C, H, W = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 512, 512
x16 = ops.nchw_to_f16k(torch.randn(B, C, H, W, device=dev))
r16 = ops.nchw_to_f16k(torch.randn(B, C, H, W, device=dev))
w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
bias = torch.randn(C, device=dev)
d = ops.make_conv_desc(B, C, H, W, C, 3, 3, 1, 1, in_ctot=C, out_ctot=C, act=ops.ACT_LEAKY, prec=_lib.PREC_BF16)
wp = ops.pack_conv_f16k_weight(w, d)
wr = ops.pack_conv3x3_resident_weight(w)
y = ops.f16k_empty(B, C, H, W, dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for name, fn in (("conv_f16k          ", lambda: ops.conv2d_f16k_res(x16, wp, bias, d, y16=y)),
                 ("conv_f16k + res    ", lambda: ops.conv2d_f16k_res(x16, wp, bias, d, y16=y, res1=r16, res_ctot=C)),
                 ("resident           ", lambda: ops.conv3x3_resident(x16, wr, bias, B, C, C, H, W, act=ops.ACT_LEAKY, y16=y)),
                 ("resident + res     ", lambda: ops.conv3x3_resident(x16, wr, bias, B, C, C, H, W, act=ops.ACT_LEAKY, y16=y, res1=r16, res_ctot=C))):
    us = t(fn)
    print(f"{name} {us:7.1f} us   {2 * 9 * C * C * B * H * W / us / 1e6:6.0f} TFLOP/s   {(2 * B * C * H * W * 2) / us / 1e6:5.2f} TB/s in+out")
