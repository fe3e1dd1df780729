import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import make_optimizers, train_step
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(int(os.environ.get("TRAIN_PROF_B", "8")), int(os.environ.get("TRAIN_PROF_H", "512")), int(os.environ.get("TRAIN_PROF_H", "512")), seed=100))
opt, aopt = make_optimizers(net, fused=os.environ.get("TRAIN_PROF_FUSED", "1") != "0")
if os.environ.get("TRAIN_PROF_GRAPH"):            # the whole step as one HIP-graph replay (masic_amd/graph.py: GraphedTrainStep)
    from masic_amd.graph import GraphedTrainStep
    g = GraphedTrainStep(net, x1, x2, hm, 0.01)
    train_step = lambda net, opt, aopt, x1, x2, hm, l: g(x1, x2, hm)
for _ in range(2): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("TRAIN_PROF_STEPS", "3"))
for _ in range(n): train_step(net, opt, aopt, x1, x2, hm, 0.01)
th = time.perf_counter()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / n * 1e3, " of which the host needs", (th - t0) / n * 1e3, "to enqueue a step")
