import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coremasic', 'mywork'))
import MASIC
from compressai.entropy_models import EntropyModel
from masic_amd import nn as mnn, synth, train
from masic_amd.graph import GraphedTrainStep
from oracle import hsic_oracle as O
DEV="cuda"
N, M, K, B, H, W = 16, 32, 3, 2, 64, 64
sd0 = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=55)
batches = [tuple(t.to(DEV) for t in synth.synth_inputs(B, H, W, seed=60 + i)) for i in range(3)]
noises = [synth.synth_noise(B, N, M, H, W, seed=70 + i) for i in range(3)]
slots = [noises[0][k].to(DEV).clone() for k in O.NOISE_KEYS]
state = {"i": 0}
def static_noise(self, x):
    t = slots[state["i"] % len(slots)]; state["i"] += 1
    return t.reshape(x.shape)
def load_noise(it):
    state["i"] = 0
    for s, k in zip(slots, O.NOISE_KEYS): s.copy_(noises[it][k].to(DEV))
EntropyModel._get_noise_cached = static_noise
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
def fresh():
    net = MASIC.HSIC(N, M, K); net.load_state_dict(sd0); return net.to(DEV).train()
KEEP = []
def eager():
    net = fresh()
    if "keep" in sys.argv: KEEP.append(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True); aopt = torch.optim.Adam(net.aux_parameters(), lr=1e-3, capturable=True)
    out = []
    for it in range(3):
        load_noise(it)
        crit, aux = train.train_step(net, opt, aopt, *batches[it], 0.01)
        out.append((float(crit["loss"]), float(aux)))
    return out
def eager_params(nsteps):
    net = fresh()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=True); aopt = torch.optim.Adam(net.aux_parameters(), lr=1e-3, capturable=True)
    for it in range(nsteps):
        load_noise(it); train.train_step(net, opt, aopt, *batches[it], 0.01)
    return {n: p.detach().clone() for n, p in net.named_parameters()}, {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in net.named_parameters()}
P1, G1 = eager_params(1)
print("eager A", eager())
print("eager B", eager())
net = fresh(); load_noise(0)
step = GraphedTrainStep(net, *batches[0], 0.01)
if "check" in sys.argv:
    for n, p in net.named_parameters(): assert torch.equal(p.detach().cpu(), sd0[n])
if "diff" in sys.argv:
    load_noise(0); step(*batches[0]); torch.cuda.synchronize()
    for n, p in net.named_parameters():
        d = float((p.detach() - P1[n]).abs().max()); ref = float((P1[n] - sd0[n].to(DEV)).abs().max())
        gd = float((p.grad - G1[n]).abs().max()) if (p.grad is not None and G1[n] is not None) else -1.0
        gref = float(G1[n].abs().max()) if G1[n] is not None else 0.0
        if d > 1e-7 or gd > 1e-6 * max(gref, 1e-30): print(f"  {n:60s} param diff {d:.3e} (step {ref:.3e})  grad diff {gd:.3e} (max {gref:.3e})")
    sys.exit(0)
out = []
for it in range(3):
    load_noise(it); crit, aux = step(*batches[it]); out.append((float(crit["loss"]), float(aux)) + ((float(crit["psnr1"]),) if "psnr" in sys.argv else ()))
print("graph  ", out)
