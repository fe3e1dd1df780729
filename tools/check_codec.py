"""compress -> decompress round trip on synthetic weights: sizes, timing, equality (run on the GPU box)."""
import os, sys, time, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn

def run(N, M, K, H, W, prec, seed=100):
    mnn.set_precision(prec)
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=seed))
    net = net.cuda().eval()
    net.update()
    x1, x2, hm = (t.cuda() for t in synth.synth_inputs(1, H, W, seed=seed))
    d = tempfile.mkdtemp()
    with torch.no_grad():
        fwd = net(x1, x2, hm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc = net.compress(x1, x2, hm, "pair", d)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dec = net.decompress(None, None, hm, "pair", d)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    est = sum(float((-torch.log2(v)).sum()) for v in fwd["likelihoods"].values()) / 8
    print(f"{prec} HSIC({N},{M},{K}) {H}x{W}: {enc['bytes']} bytes (estimate {est:.0f}), enc {1e3*(t1-t0):.1f} ms, dec {1e3*(t2-t1):.1f} ms,"
          f" minmax {int(enc['y1_hat'].abs().max())}/{int(enc['y2_hat'].abs().max())}")
    for k in ("y1_hat", "y2_hat", "z1_hat", "z2_hat", "x1_hat", "x2_hat"):
        same = torch.equal(enc[k], dec[k])
        print("   ", k, "equal" if same else f"DIFFERENT ({int((enc[k] != dec[k]).sum())} elements)")
    print("    fwd y1_hat equal:", torch.equal(fwd["y1_hat"], enc["y1_hat"]))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        run(128, 192, 5, 512, 896, "bf16")
        run(128, 192, 5, 1216, 2176, "bf16")
        sys.exit(0)
    run(32, 48, 3, 128, 192, "f32")
    run(32, 48, 3, 128, 192, "bf16")
    run(128, 192, 5, 256, 256, "bf16")
    run(128, 192, 5, 512, 512, "bf16")
