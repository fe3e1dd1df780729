"""eval forward: eager vs graph replay equality and timing (new stream schedule)."""
import os, sys, time, faulthandler
faulthandler.enable()
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.graph import GraphedHSIC
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
with torch.no_grad():
    out = net(x1, x2, hm); torch.cuda.synchronize(); print("eager ok", flush=True)
    g = GraphedHSIC(net, x1, x2, hm); print("capture ok", flush=True)
    rep = g(x1, x2, hm); torch.cuda.synchronize()
    for k in ("x1_hat", "x2_hat", "y1_hat", "z1_hat"):
        print(k, torch.equal(out[k], rep[k]))
    for k, v in out["likelihoods"].items():
        print("lik", k, torch.equal(v, rep["likelihoods"][k]))
    for name, fn in (("graph", g), ("eager", net)):
        for _ in range(3): fn(x1, x2, hm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn(x1, x2, hm)
        torch.cuda.synchronize(); print(name, "ms/step", (time.perf_counter() - t0) / 20 * 1e3)
