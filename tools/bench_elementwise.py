"""HBM-bound glue kernels of the training steps at Independent_EN's shapes (8 x C x 512 x 512), each alone: us per launch and the
algorithmic bytes it moves per second, against the ~6.3 TB/s a float4 copy reaches on this chip (MI355X_MICROARCH.md).

    python tools/bench_elementwise.py [C=32] [B=8] [H=512]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from masic_amd import ops

C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
W, HW = H, H * H
dev = "cuda"
x = torch.randn(B, C, H, W, device=dev)
g = torch.randn(B, C, H, W, device=dev)
big = torch.empty(B, 3 * C, H, W, device=dev)
gate = torch.rand(B, 2, H, W, device=dev)
x16 = ops.nchw_to_f16k(x)
g16 = ops.nchw_to_f16k(g)
n = B * C * HW


def t(name, fn, bytes_):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:44s} {us:8.1f} us   {bytes_ / us / 1e6:6.2f} TB/s   ({bytes_ / 1e6:.0f} MB)", flush=True)


print(f"B {B}, C {C}, {H} x {W}")
t("nchw_to_f16k", lambda: ops.nchw_to_f16k(x), n * 6)
t("f16k_to_nchw (float32)", lambda: ops.f16k_to_nchw_dev(x16, B, C, H, W), n * 6)
t("f16k_to_nchw (bf16)", lambda: ops.f16k_to_nchw_dev(x16, B, C, H, W, bf16=True), n * 4)
t("f16k_act_bwd", lambda: ops.f16k_act_bwd(g16, x16, 0.01), n * 6)
t("f16k_channel_sum", lambda: ops.f16k_channel_sum(x16, B, C, HW), n * 2)
t("channel_sum (float32 NCHW)", lambda: ops.channel_sum(x), n * 4)
t("copy_view into 3C concat", lambda: ops.copy_view(x, big, C), n * 8)
t("slice_copy out of 3C concat", lambda: ops.slice_copy(big, C, C), n * 8)
t("quantize copy + gate into concat", lambda: ops.quantize(x, "copy", out=big, out_coff=C, gate=gate, gate_c=1), n * 8 + B * HW * 4)
t("gate_bwd", lambda: ops.gate_bwd(g, x, gate, 1), n * 12 + B * HW * 8)
t("elementwise act_bwd (float32)", lambda: ops.elementwise(ops.EW_ACT_BWD, g, x, s0=ops.ACT_LEAKY), n * 12)
t("torch add (float32)", lambda: torch.add(x, g), n * 12)
t("torch copy (float32)", lambda: x.clone(), n * 8)
