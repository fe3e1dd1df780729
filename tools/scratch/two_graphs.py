import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "coremasic", "mywork"))
import MASIC
from masic_amd import synth, nn as mnn
from masic_amd.graph import GraphedHSIC
mnn.set_precision("bf16")
dev = "cuda"
net = MASIC.HSIC(128, 192, 5); net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=100)); net = net.to(dev).eval()
x1, x2, hm = (t.to(dev) for t in synth.synth_inputs(8, 512, 512, seed=100))
with torch.no_grad():
    ga = GraphedHSIC(net, x1, x2, hm)
    gb = GraphedHSIC(net, x1, x2, hm)
    xa, xb = ga.inputs; ya, yb = gb.inputs
    K = 40
    for _ in range(4): ga(xa, xb, hm, next_h_matrix=hm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): ga(xa, xb, hm, next_h_matrix=hm)
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"one graph, one stream: {t1 / K * 1e3:.3f} ms/step, {8 * K / t1:.0f} pairs/s")
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for s in (sA, sB): s.wait_stream(torch.cuda.current_stream())
    def run(n):
        for i in range(n):
            g, s, a, b = (ga, sA, xa, xb) if i % 2 == 0 else (gb, sB, ya, yb)
            with torch.cuda.stream(s):
                g(a, b, hm, next_h_matrix=hm)
    run(4); torch.cuda.synchronize(); t0 = time.perf_counter()
    run(K); torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"two graphs alternating on two streams: {t2 / K * 1e3:.3f} ms/step, {8 * K / t2:.0f} pairs/s")
    # bare replays (no homography work) for reference
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        g, s = (ga, sA) if i % 2 == 0 else (gb, sB)
        with torch.cuda.stream(s): g.graph.replay()
    torch.cuda.synchronize(); t3 = time.perf_counter() - t0
    print(f"bare replays alternating on two streams: {t3 / K * 1e3:.3f} ms/step")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): ga.graph.replay()
    torch.cuda.synchronize(); t4 = time.perf_counter() - t0
    print(f"bare replays, one graph: {t4 / K * 1e3:.3f} ms/step")
