"""Which torch (ATen) ops a training step still launches, with the Python frames that issue them (torch.profiler, one step)."""
import sys, os, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import make_optimizers, train_step
from torch.profiler import profile, ProfilerActivity
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(2, 256, 256, seed=100))
opt, aopt = make_optimizers(net)
for _ in range(2): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_step(net, opt, aopt, x1, x2, hm, 0.01)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::zero_", "aten::fill_", "aten::mul", "aten::mul_", "aten::cat", "aten::clone", "aten::contiguous", "aten::sum", "aten::div", "aten::to", "aten::_to_copy"):
        frames = [f for f in (ev.stack or []) if "/repo/" in f or "masic" in f.lower()][:2]
        cnt[(ev.name, tuple(frames))] += 1
for (name, frames), n in cnt.most_common(40):
    print(n, name, " <- ".join(f.split("/")[-1] for f in frames))
