"""Is the two-stream training forward / backward (MASIC._TRAIN_STREAMS) the same computation as the one-stream one?  Gradients of one step with
fixed noise: one-stream twice (the run-to-run floor of the float atomics), then two-stream runs against the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import torch
import MASIC
from compressai.entropy_models import EntropyModel
from masic_amd import nn as mnn, synth, autograd as ag
from masic_amd.loss import rate_distortion
from oracle import hsic_oracle as O
N, M, K, B, H, W = 128, 192, 5, int(os.environ.get("SD_B", "2")), int(os.environ.get("SD_H", "256")), int(os.environ.get("SD_H", "256"))
sd0 = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=55)
batch = tuple(t.cuda() for t in synth.synth_inputs(B, H, W, seed=60))
noise = synth.synth_noise(B, N, M, H, W, seed=70)
slots = [noise[k].cuda() for k in O.NOISE_KEYS]
state = {"i": 0}
def static_noise(self, x):
    t = slots[state["i"] % len(slots)]; state["i"] += 1
    return t.reshape(x.shape)
EntropyModel._get_noise_cached = static_noise
mnn.set_precision(os.environ.get("SD_PREC", "bf16"))
net = MASIC.HSIC(N, M, K); net.load_state_dict(sd0); net = net.cuda().train()
def grads(streams):
    MASIC._TRAIN_STREAMS = streams
    state["i"] = 0
    for _, p in net.named_parameters():
        p.grad = None
    out = net(*batch)
    rate_distortion(out, batch[0], batch[1], 0.01)["loss"].backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    worst = sorted(((float((a[n] - b[n]).norm() / (a[n].norm() + 1e-30)), n) for n in a), reverse=True)[:3]
    print(f"{tag:34s} worst relative gradient differences: " + ", ".join(f"{n} {v:.2e}" for v, n in worst), flush=True)
for _ in range(2): grads(False)
g0 = grads(False)
cmp(g0, grads(False), "one stream vs one stream")
for i in range(int(os.environ.get("SD_RUNS", "12"))): cmp(g0, grads(True), f"two streams (run {i}) vs one stream")
