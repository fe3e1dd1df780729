import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from masic_amd import ops, synth
dev = "cuda"
torch.manual_seed(1)
B, H, W = 1, 64, 64
x = torch.randn(B, 3, H, W)
q = lambda t: t.bfloat16().float()
def conv_only(w, b, tag):
    ref = F.conv2d(q(x), q(w), b, stride=2, padding=2)
    wp = ops.pack_conv_a_weight(w.to(dev))
    y16, Ho, Wo = ops.conv_a_f16k(x.to(dev), wp, None if b is None else b.to(dev))
    y = ops.f16k_to_nchw(y16, B, 128, Ho, Wo).cpu()
    d = (y - ref).abs()
    print(tag, "max err / max ref", float(d.max() / ref.abs().max()), "worst idx", np.unravel_index(int(d.argmax()), d.shape))
    return y, ref
w = torch.randn(128, 3, 5, 5) / 75 ** 0.5
conv_only(w, None, "full weight, no bias")
conv_only(w, torch.randn(128) * 0.1, "full weight + bias")
for ci in range(3):
    for kh in range(5):
        errs = []
        for kw in range(5):
            w1 = torch.zeros(128, 3, 5, 5); w1[:, ci, kh, kw] = torch.randn(128)
            ref = F.conv2d(q(x), q(w1), None, stride=2, padding=2)
            wp = ops.pack_conv_a_weight(w1.to(dev))
            y16, Ho, Wo = ops.conv_a_f16k(x.to(dev), wp, None)
            y = ops.f16k_to_nchw(y16, B, 128, Ho, Wo).cpu()
            errs.append(float((y - ref).abs().max() / ref.abs().max()))
        print("tap ci", ci, "kh", kh, " ".join(f"{e:.1e}" for e in errs))
