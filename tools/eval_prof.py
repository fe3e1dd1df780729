import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.graph import GraphedHSIC
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(8, 512, 512, seed=100))
if prec == "fp8":
    from masic_amd import fp8
    fp8.calibrate(net, [tuple(t.cuda() for t in synth.synth_inputs(2, 512, 512, seed=1100))])
mnn.set_precision(prec)
net.serial_schedule = serial
with torch.no_grad():
    step = net if serial else GraphedHSIC(net, x1, x2, hm)
    kw = {"next_h_matrix": hm} if (not serial and os.environ.get("EVAL_PROF_AHEAD", "1") != "0") else {}
    for _ in range(3): step(x1, x2, hm, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("EVAL_PROF_STEPS", "5"))
    for _ in range(n): step(x1, x2, hm, **kw)
    torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / n * 1e3)
