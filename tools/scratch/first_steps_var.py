"""Five training steps from default init in float32 (the convergence test's loop), repeated: one stream vs side streams, loss of every step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import torch
import MASIC
from masic_amd import nn as mnn, synth, trainrun
from masic_amd.train import make_optimizers, train_step
from oracle import hsic_oracle as O
from tests.test_gpu_driver_loop import _Noise
N, M, K, B, H, W = 128, 192, 5, 1, 128, 128
net0 = trainrun.default_init(N, M, K, seed=0, device="cpu")
sd0 = {k: v.detach().clone() for k, v in net0.state_dict().items()}
x1, x2, hm = synth.synth_inputs(B, H, W, seed=8100)
noises = [synth.synth_noise(B, N, M, H, W, seed=8100 + i) for i in range(5)]
d1, d2, h = x1.cuda(), x2.cuda(), hm.cuda()
mnn.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")
def run(streams, sync=False):
    MASIC._TRAIN_STREAMS = streams
    net = MASIC.HSIC(N, M, K); net.load_state_dict(sd0); net = net.cuda().train()
    opt, aopt = make_optimizers(net, fused=False)
    out = []
    for it in range(5):
        with _Noise([noises[it][k].cuda() for k in O.NOISE_KEYS]):
            if sync: torch.cuda.synchronize()
            crit, aux = train_step(net, opt, aopt, d1, d2, h, 0.0932)
        out.append(float(crit["loss"]))
    return out
for streams in (False, False, True, True, True, True, False, True):
    print("side streams" if streams else "one stream  ", " ".join(f"{v:.4f}" for v in run(streams)), flush=True)
