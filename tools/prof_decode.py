"""Where a decode step of HSIC.decompress spends its time (host-side sections; the .cpu() wait absorbs the GPU work)."""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn, codec

mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval(); net.update()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(1, 512, 512, seed=100))
d = tempfile.mkdtemp()
T = {}
def tick(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
orig_tables, orig_rows = codec.gmm_tables, codec.AdaptiveDecoder.decode_rows
def tables(*a, **k):
    t0 = time.perf_counter(); r = orig_tables(*a, **k); tick("tables_launch", t0); return r
def rows(self, s):
    t0 = time.perf_counter(); r = orig_rows(self, s); tick("host_decode", t0); return r
codec.gmm_tables, codec.AdaptiveDecoder.decode_rows = tables, rows
with torch.no_grad():
    net.compress(x1, x2, hm, "p", d)
    for rep in range(2):
        T.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.decompress(None, None, hm, "p", d)
        torch.cuda.synchronize(); print("decode ms", (time.perf_counter() - t0) * 1e3, {k: round(v * 1e3, 1) for k, v in T.items()})
    # GPU time of one params pass
    z = torch.zeros(1, 192, 32, 32, device="cuda")
    fn = net._left_params_fn(torch.zeros(1, 128, 8, 8, device="cuda"), 32, 32)
    for _ in range(3): fn(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn(z)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("params pass: issue %.3f ms, complete %.3f ms per call" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
