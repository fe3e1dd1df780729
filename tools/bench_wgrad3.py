import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from masic_amd import ops, _lib
dev = "cuda"
for C in (32, 64, 96):
    B, H, W = 8, 512, 512
    x = torch.randn(B, C, H, W, device=dev); g = torch.randn(B, C, H, W, device=dev)
    x16, g16 = ops.nchw_to_f16k(x), ops.nchw_to_f16k(g)
    d = ops.make_conv_desc(B, C, H, W, C, 3, 3, 1, 1, prec=_lib.PREC_BF16)
    res = {}
    for tag, fn in (("old (conv_wgrad_f32<1,3,true>, NCHW f32)", lambda: ops.conv2d_wgrad(x, g, d, (C, C, 3, 3))),
                    ("new (F16K, tr reads)", lambda: ops.conv3x3_wgrad_f16k(x16, g16, B, C, C, H, W))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        res[tag] = dt
        print(f"C={C}: {tag}: {dt * 1e6:8.1f} us  {2.0 * B * H * W * C * C * 9 / dt / 1e12:7.1f} TF/s", flush=True)
