"""Times the fused GDN backward (masic_gdn_bwd_fused_ex2) at the three resolutions of a transform, with the timing-only ablations of
MASIC_GDNB_DBG (1 no stores, 2 no third contraction, 4 no loads, 8 no first two contractions):  python tools/bench_gdn_bwd.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from masic_amd import ops
B = 8
for H in (256, 128, 64):
    x = torch.randn(B, 128, H, H, device="cuda"); g = torch.randn(B, 128, H, H, device="cuda")
    x16, g16 = ops.nchw_to_f16k(x), ops.nchw_to_f16k(g)
    beta = torch.rand(128, device="cuda") + 1.0; gamma = torch.rand(128, 128, device="cuda") * 0.1
    kw = dict(want_nchw=False, want_f16k=True, want_sum=True, want_b16=True)
    for _ in range(3): ops.gdn_bwd_fused_ex(x16, g16, (B, 128, H, H), beta, gamma, False, 1e-6, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.gdn_bwd_fused_ex(x16, g16, (B, 128, H, H), beta, gamma, False, 1e-6, **kw)
    torch.cuda.synchronize(); print(H, "us/call", round((time.perf_counter() - t0) / 20 * 1e6, 1))
