import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
train = len(sys.argv) > 2 and sys.argv[2] == "train"
en = MASIC.Independent_EN(); en.load_state_dict(synth.synth_state_dict(en.state_dict(), seed=101)); en = en.cuda()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
if train:
    from masic_amd.loss import distortion
    en.train(); opt = torch.optim.Adam(en.parameters(), lr=1e-4)
    def step():
        opt.zero_grad(); distortion(en(x1, x2, hm), x1, x2, 0.01)["loss"].backward(); opt.step()
else:
    en.eval()
    def step():
        with torch.no_grad(): en(x1, x2, hm)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
