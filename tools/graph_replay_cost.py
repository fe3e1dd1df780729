"""Host time of one HIP-graph replay call and device time per replay, for the eval forward graph and the training-step graph."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.graph import GraphedHSIC, GraphedTrainStep
B = int(os.environ.get("B", "8"))
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(B, 512, 512, seed=100))
def measure(name, g, n=30):
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3 * (t1 - t0) / n:.3f} ms per replay call, total {1e3 * (t2 - t0) / n:.3f} ms per replay")
with torch.no_grad():
    ge = GraphedHSIC(net, x1, x2, hm)
measure(f"eval graph B={B}", ge.graph)
net.train()
gt = GraphedTrainStep(net, x1, x2, hm, 0.01)
measure(f"train graph B={B}", gt.graph, 10)
def timeit(name, fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3 * (t1 - t0) / n:.3f} ms, total {1e3 * (t2 - t0) / n:.3f} ms per call")
timeit("train __call__", lambda: gt(x1, x2, hm))
timeit("replay + touch", lambda: (gt.graph.replay(), gt._touch()))
timeit("replay + copies", lambda: (gt.d1.copy_(x1), gt.d2.copy_(x2), gt.h.copy_(hm), gt.graph.replay()))
timeit("touch only", gt._touch, 50)
