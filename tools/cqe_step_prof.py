"""The CQE training step as bench.py times it (masic_amd.train.cqe_train_step: HSIC eval forward under no_grad + Independent_EN step)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import cqe_train_step
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
en = MASIC.Independent_EN(); en.load_state_dict(synth.synth_state_dict(en.state_dict(), seed=101)); en = en.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
fused = os.environ.get("CQE_FUSED", "1") != "0"
opt = torch.optim.Adam(list(en.parameters()), lr=1e-4, fused=fused)
for _ in range(2): cqe_train_step(net, en, opt, x1, x2, hm, 0.01)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("STEPS", "5"))
for _ in range(n): cqe_train_step(net, en, opt, x1, x2, hm, 0.01)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / n * 1e3, "fused" if fused else "foreach")
