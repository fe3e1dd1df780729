"""One training step in the bf16-operand mode against the float32 path, same weights / inputs / noise: loss and per-parameter
gradient agreement (cosine, relative L2).  Exercises the fused GDN backward, the bf16 weight-gradient kernels and the F16K
input gradients in situ."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from compressai.entropy_models import EntropyModel
from masic_amd import synth, nn as mnn
from masic_amd.loss import rate_distortion

def grads(prec, N, M, K, B, H, W, seed, perturb=False):
    mnn.set_precision(prec)
    net = MASIC.HSIC(N, M, K); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=seed)); net = net.cuda().train()
    x1, x2, hm = (t.cuda() for t in synth.synth_inputs(B, H, W, seed=seed))
    if perturb:      # the size of a bf16 rounding of the inputs, in the float32 mode: how sensitive are the gradients themselves?
        x1, x2 = x1.bfloat16().float(), x2.bfloat16().float()
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    orig = EntropyModel._get_noise_cached
    EntropyModel._get_noise_cached = lambda self, x: (torch.rand(x.shape, device=x.device, generator=g) - 0.5)
    try:
        out = net(x1, x2, hm)
    finally:
        EntropyModel._get_noise_cached = orig
    crit = rate_distortion(out, x1, x2, 0.01)
    crit["loss"].backward()
    mnn.set_precision("f32")
    return float(crit["loss"]), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}

if __name__ == "__main__":
    N, M, K = 128, 192, 5
    lf, gf = grads("f32", N, M, K, 2, 128, 256, 7)
    lb, gb = grads("bf16", N, M, K, 2, 128, 256, 7) if len(sys.argv) < 2 else grads("f32", N, M, K, 2, 128, 256, 7, perturb=True)
    print("loss f32 %.6f bf16 %.6f rel %.2e" % (lf, lb, abs(lb - lf) / abs(lf)))
    rows = []
    for n in gf:
        a, b = gf[n].flatten(), gb[n].flatten()
        na, nb = float(a.norm()), float(b.norm())
        cos = float((a @ b) / (na * nb + 1e-300))
        rows.append((cos, float((a - b).norm()) / (na + 1e-300), na, n))
    rows.sort()
    for r in rows[:12]:
        print("cos %.4f relL2 %.3f |g| %.3e %s" % r)
    print("min cos", rows[0][0], "median relL2", sorted(r[1] for r in rows)[len(rows) // 2], "max relL2", max(r[1] for r in rows))
