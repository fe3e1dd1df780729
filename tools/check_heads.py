import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'coremasic', 'mywork'))
import MASIC
from masic_amd import synth, nn as mnn, ops
from masic_amd.loss import rate_distortion
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(2, 256, 256, seed=100))
with torch.no_grad():
    mnn.set_precision("f32"); of = net(x1, x2, hm); cf = rate_distortion(of, x1, x2, 0.01)
    mnn.set_precision("bf16"); ob = net(x1, x2, hm); cb = rate_distortion(ob, x1, x2, 0.01)
    cat = torch.randn(2, 768, 16, 16, device="cuda")
    h = net._h_s1_same_resolution
    s_g, m_g, l_g = h.heads(cat)
    s_c = h._branch(h.gmm_sigma, cat, (1, 1, 1))
    print("bpp f32", float(cf["bpp_loss"]), "bf16", float(cb["bpp_loss"]))
    print("heads gemm vs conv-bf16 sigma maxdiff", float((s_g - s_c).abs().max()), "max", float(s_c.abs().max()))
    mnn.set_precision("f32"); s_f = h._branch(h.gmm_sigma, cat, (1, 1, 1))
    print("heads gemm vs f32 sigma rel", float((s_g - s_f).abs().max() / s_f.abs().max()))
