"""Host wall time of the phases inside codec.decode_view (synchronised between phases): where the 2 ms per view beside the replays go."""
import os, sys, time, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn, codec
mnn.set_precision("bf16")
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.cuda().eval(); net.update()
x1, x2, hm = (t.cuda() for t in synth.synth_inputs(1, 512, 512, seed=100))
d = tempfile.mkdtemp()
marks = []
def mark(name):
    torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
# instrument by wrapping the pieces decode_view uses
orig_graph = torch.cuda.graph
class TimedGraph(orig_graph):
    def __enter__(self):
        mark("before capture"); return super().__enter__()
    def __exit__(self, *a):
        r = super().__exit__(*a); mark("after capture"); return r
orig_dv = codec.decode_view
def dv(*a, **k):
    mark("decode_view enter")
    r = orig_dv(*a, **k)
    mark("decode_view exit")
    return r
orig_replay = torch.cuda.CUDAGraph.replay
first = [True]
def replay(self):
    if first[0]:
        mark("first replay"); first[0] = False
    return orig_replay(self)
with torch.no_grad():
    net.compress(x1, x2, hm, "p", d)
    net.decompress(None, None, hm, "p", d)
    net.decompress(None, None, hm, "p", d)
    codec.decode_view = dv; torch.cuda.graph = TimedGraph; torch.cuda.CUDAGraph.replay = replay
    mark("decompress enter")
    net.decompress(None, None, hm, "p", d)
    mark("decompress exit")
t0 = marks[0][1]
prev = t0
for n, t in marks:
    print(f"{n:22s} +{1e3 * (t - prev):6.2f} ms   (at {1e3 * (t - t0):6.2f})"); prev = t
