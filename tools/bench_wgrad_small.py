"""Times ops.conv2d_wgrad on the pre_conv / after_conv shapes (8 x 512 x 512):  python tools/bench_wgrad_small.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from masic_amd import ops
from masic_amd._lib import PREC_BF16
B, H, W = 8, 512, 512
for tr in (False, True):
    d = ops.make_conv_desc(B, 6, H, W, 3, 5, 5, 1, 2, transposed=tr, prec=PREC_BF16)
    x = torch.randn(B, 6, H, W, device="cuda"); dy = torch.randn(B, 3, H, W, device="cuda")
    ws = (3, 6, 5, 5) if not tr else (6, 3, 5, 5)
    for _ in range(3): ops.conv2d_wgrad(x, dy, d, ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.conv2d_wgrad(x, dy, d, ws)
    torch.cuda.synchronize(); print("transposed" if tr else "conv", "us/call", (time.perf_counter() - t0) / 20 * 1e6)
