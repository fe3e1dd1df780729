"""Where the 18 ms of HSIC.decompress go: cProfile of one call with a synchronisation after every decode_view (host wall time per phase)."""
import os, sys, time, tempfile, torch, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn, codec
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval(); net.update()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(1, 512, 512, seed=100))
d = tempfile.mkdtemp()
orig = codec.decode_view
def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); print(f"   decode_view {1e3 * (time.perf_counter() - t0):.2f} ms")
    return r
with torch.no_grad():
    net.compress(x1, x2, hm, "p", d)
    net.decompress(None, None, hm, "p", d)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.decompress(None, None, hm, "p", d)
        torch.cuda.synchronize(); print("decode ms", (time.perf_counter() - t0) * 1e3)
    codec.decode_view = timed
    import masic_amd.codec as C
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.decompress(None, None, hm, "p", d)
    torch.cuda.synchronize(); print("decode ms with per-view syncs", (time.perf_counter() - t0) * 1e3)
    codec.decode_view = orig
    pr = cProfile.Profile(); pr.enable()
    net.decompress(None, None, hm, "p", d)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(28); print(s.getvalue()[:5000])
