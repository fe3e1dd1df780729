import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from masic_amd import ops, synth, _lib
dev = "cuda"
torch.manual_seed(1)
B, H, W = 8, 512, 512
x = torch.rand(B, 3, H, W, device=dev)
w = (torch.randn(128, 3, 5, 5) / 75 ** 0.5).to(dev)
b = (torch.randn(128) * 0.1).to(dev)
rs = np.random.RandomState(3)
beta = synth.synth_tensor("g.beta", (128,), rs).to(dev); gamma = synth.synth_tensor("g.gamma", (128, 128), rs).to(dev)
gp = ops.pack_gdn_f16k(beta, gamma)
wp = ops.pack_conv_a_weight(w)
for _ in range(5): ops.conv_a_gdn_f16k(x, wp, b, (gp, False))
torch.cuda.synchronize()
st = torch.zeros(80, dtype=torch.int64, device=dev)
_lib.lib.masic_conv_f16k_set_stamps(ctypes.c_void_p(st.data_ptr()))
ops.conv_a_gdn_f16k(x, wp, b, (gp, False))
torch.cuda.synchronize()
_lib.lib.masic_conv_f16k_set_stamps(None)
raw = st.cpu().numpy()
print("workgroup 0   : entry, prologue issued, landed, synced, loop done, stores drained:", " ".join(f"{(v - raw[48]) / 100.0:7.2f}" for v in raw[48:54]))
print("last workgroup: (relative to workgroup 0's entry)                               :", " ".join(f"{(v - raw[48]) / 100.0:7.2f}" for v in raw[56:62]))
s = raw[:48].reshape(-1, 6)[:8]
t0 = s[0, 0]
print("per tile (us): 6 stamps, see kernel")
for r in s:
    print(" ".join(f"{(v - t0) / 100.0:7.2f}" for v in r))
