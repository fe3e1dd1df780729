import sys, os, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.graph import GraphedHSIC
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
y1, y2, hm2 = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=101))
with torch.no_grad():
    ref = net(y1, y2, hm2)
    ref = {k: (v.clone() if torch.is_tensor(v) else {a: b.clone() for a, b in v.items()}) for k, v in ref.items()}
    g = GraphedHSIC(net, x1, x2, hm)
    out = g(y1, y2, hm2)
    torch.cuda.synchronize()
    print("graph == eager:", all(torch.equal(out[k], ref[k]) for k in ("x1_hat", "x2_hat", "y1_hat")), torch.equal(out["likelihoods"]["y2"], ref["likelihoods"]["y2"]))
    for name, fn in (("eager", lambda: net(x1, x2, hm)), ("graph", lambda: g(x1, x2, hm))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); print(name, "ms/step", (time.perf_counter() - t0) / 20 * 1e3)
