"""Lists the host <-> device synchronisation points of one HSIC training step (torch's sync debug mode): each one is a place where
the host stops launching until the device has drained."""
import sys, os, warnings, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.train import make_optimizers, train_step
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().train()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(2, 256, 256, seed=100))
opt, aopt = make_optimizers(net)
for _ in range(2): train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    train_step(net, opt, aopt, x1, x2, hm, 0.01)
torch.cuda.set_sync_debug_mode("default")
print(len(w), "synchronising calls in one step")
for x in w:
    print(" ", x.filename.replace(ROOT, "."), x.lineno, str(x.message)[:100])
