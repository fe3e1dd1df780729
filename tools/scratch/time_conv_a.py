import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from masic_amd import ops, synth
dev = "cuda"
torch.manual_seed(1)
B, H, W = int(os.environ.get("CA_B", "8")), 512, 512
x = torch.rand(B, 3, H, W, device=dev)
w = (torch.randn(128, 3, 5, 5) / 75 ** 0.5).to(dev)
b = (torch.randn(128) * 0.1).to(dev)
rs = np.random.RandomState(3)
beta = synth.synth_tensor("g.beta", (128,), rs).to(dev); gamma = synth.synth_tensor("g.gamma", (128, 128), rs).to(dev)
gp = ops.pack_gdn_f16k(beta, gamma)
wp = ops.pack_conv_a_weight(w)
for _ in range(5): ops.conv_a_gdn_f16k(x, wp, b, (gp, False))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 100
e0.record()
for _ in range(n): ops.conv_a_gdn_f16k(x, wp, b, (gp, False))
e1.record(); torch.cuda.synchronize()
print(os.environ.get("MASIC_HIP_LIB", "default"), f"{e0.elapsed_time(e1) / n * 1e3:.1f} us per launch")
