import os, sys, time, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval(); net.update()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(1, 512, 512, seed=100))
d = tempfile.mkdtemp()
with torch.no_grad():
    net.compress(x1, x2, hm, "p", d)
    net.decompress(None, None, hm, "p", d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.decompress(None, None, hm, "p", d)
    torch.cuda.synchronize(); print("decode ms", (time.perf_counter() - t0) * 1e3)
