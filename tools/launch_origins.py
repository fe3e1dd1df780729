"""Who launches what in one training step: every call through the C ABI during one step of masic_amd.train.train_step (or, with `cqe`,
cqe_train_step), counted by (symbol, Python call chain above masic_amd/ops.py).  For the launch-count work of DESIGN.md section 6.

    python tools/launch_origins.py [hsic|cqe] [depth=3]
"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import torch  # noqa: E402
import MASIC  # noqa: E402
from masic_amd import nn as mnn, ops, synth  # noqa: E402
from masic_amd.train import cqe_train_step, make_optimizers, train_step  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "hsic"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
counts = collections.Counter()
on = [False]


class Proxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            if on[0]:
                fr = [f for f in traceback.extract_stack()[:-1] if not f.filename.endswith(("ops.py", "launch_origins.py")) and "torch/" not in f.filename]
                chain = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in reversed(fr[-depth:]))
                counts[(name, chain)] += 1
            return fn(*a)
        return call


ops.lib = Proxy(ops.lib)
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
if what == "cqe":
    net.eval()
    en = MASIC.Independent_EN().cuda().train()
    opt = torch.optim.Adam(en.parameters(), lr=1e-4, fused=True)
    step = lambda: cqe_train_step(net, en, opt, x1, x2, hm, 0.01)
else:
    net.train()
    opt, aopt = make_optimizers(net)
    step = lambda: train_step(net, opt, aopt, x1, x2, hm, 0.01)
for _ in range(2):
    step()
on[0] = True
step()
on[0] = False
torch.cuda.synchronize()
by_sym = collections.Counter()
for (s, c), n in counts.items():
    by_sym[s] += n
print("C-ABI calls in one step:", sum(counts.values()))
for s, n in by_sym.most_common():
    print(f"{n:4d}  {s}")
    for (s2, c), m in sorted(counts.items(), key=lambda kv: -kv[1]):
        if s2 == s:
            print(f"        {m:3d}  {c}")
