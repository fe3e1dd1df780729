import sys, os, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import make_optimizers, train_step
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
opt, aopt = make_optimizers(net)
for _ in range(3): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): train_step(net, opt, aopt, x1, x2, hm, 0.01)
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"cpu enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step, gpu tail after last enqueue {1e3*(t2-t1):.2f} ms")
