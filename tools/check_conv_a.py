import sys, os, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import torch.nn.functional as F
from masic_amd import ops, _lib, synth
from oracle import hsic_oracle as O
dev = "cuda"
def run(B, H, W, ctot=3, coff=0, inverse=False, reps=0):
    torch.manual_seed(1)
    x = torch.randn(B, ctot, H, W)
    w = torch.randn(128, 3, 5, 5) / 75 ** 0.5
    b = torch.randn(128) * 0.1
    rs = np.random.RandomState(3)
    beta = synth.synth_tensor("g.beta", (128,), rs); gamma = synth.synth_tensor("g.gamma", (128, 128), rs)
    q = lambda t: t.bfloat16().float()
    ref = O.gdn(F.conv2d(q(x[:, coff:coff + 3]), q(w), b, stride=2, padding=2), beta, gamma, inverse=inverse)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    gp = ops.pack_gdn_f16k(beta.to(dev), gamma.to(dev))
    wp = ops.pack_conv_a_weight(wd)
    y16, Ho, Wo = ops.conv_a_gdn_f16k(xd, wp, bd, (gp, inverse), in_coff=coff)
    y = ops.f16k_to_nchw(y16, B, 128, Ho, Wo).cpu()
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    msg = f"B{B} {H}x{W} ctot{ctot} coff{coff} inv{inverse}: err {err:.2e}"
    if reps:
        d = ops.make_conv_desc(B, 3, H, W, 128, 5, 5, 2, 2, in_ctot=ctot, in_coff=coff, prec=_lib.PREC_BF16)
        pk = ops.pack_conv_weight(wd, d)
        bt, gm = beta.to(dev), gamma.to(dev)
        for tag, fn in (("old conv + gdn_f16k", lambda: ops.gdn_f16k(ops._conv2d(xd, pk, bd, d), bt, gm, inverse=inverse)),
                        ("fused", lambda: ops.conv_a_gdn_f16k(xd, wp, bd, (gp, inverse), in_coff=coff))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); msg += f" | {tag} {(time.perf_counter() - t0) / reps * 1e6:7.1f} us"
    print(msg, flush=True)
run(1, 64, 64)
run(2, 40, 72, ctot=6, coff=3, inverse=True)
run(1, 37, 51)
run(8, 512, 512, reps=10)
