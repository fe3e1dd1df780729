import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import torch.nn.functional as F
from masic_amd import ops, _lib
dev = "cuda"
def run(B, H, W, C=3, reps=0):
    torch.manual_seed(0)
    x = torch.randn(B, 128, H, W, device=dev)
    w = torch.randn(128, C, 5, 5, device=dev) / (128 * 25 / 4) ** 0.5
    b = torch.randn(C, device=dev)
    q = lambda t: t.bfloat16().float()
    ref = F.conv_transpose2d(q(x), q(w), b, stride=2, padding=2, output_padding=1)
    wc, bc = ops.deconv_s2_as_conv_weight(w, b)
    d = ops.make_conv_desc(B, 128, H, W, 32, 3, 3, 1, 1, prec=_lib.PREC_BF16)
    wp = ops.pack_conv_f16k_weight(wc, d)
    x16 = ops.nchw_to_f16k(x)
    y = ops.conv2d_f16k_d2s(x16, wp, bc, d, C)
    msg = f"B{B} {H}x{W}: err {(y - ref).abs().max().item() / ref.abs().max().item():.2e}"
    if reps:
        d0 = ops.make_conv_desc(B, 128, H, W, C, 5, 5, 2, 2, transposed=True, prec=_lib.PREC_BF16)
        pk = ops.pack_conv_weight(w, d0)
        for tag, fn in (("old", lambda: ops._conv2d(x, pk, b, d0)), ("d2s", lambda: ops.conv2d_f16k_d2s(x16, wp, bc, d, C))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); msg += f" | {tag} {(time.perf_counter() - t0) / reps * 1e6:7.1f} us"
    print(msg, flush=True)
run(1, 16, 24); run(2, 37, 50); run(8, 256, 256, reps=10)
