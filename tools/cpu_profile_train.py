"""cProfile of the host side of HSIC training steps (bf16 mode, 8 x 512 x 512): where the Python / launch time of a step goes."""
import sys, os, time, torch, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import make_optimizers, train_step
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(int(os.environ.get("CPU_PROF_B", "8")), int(os.environ.get("CPU_PROF_H", "512")), int(os.environ.get("CPU_PROF_H", "512")), seed=100))
opt, aopt = make_optimizers(net)
for _ in range(3): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
