"""hipMemsetAsync captured by torch.cuda.graph, followed by torch kernels: which rounds of which replays come out wrong, and as what?
dw is pre-filled with 7 before the capture's memset (a kernel inside the graph), so: correct = 2x; memset skipped / run too early = 7 + 2x;
memset run late = 0, x or garbage."""
import ctypes, os, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
ROUNDS, n = 12, 6144
x = torch.ones(n, device="cuda")
for keep_alive, on_stream in ((False, False), (True, False), (False, True)):
    outs = torch.empty(ROUNDS, n, device="cuda")
    keep = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(ROUNDS):
            dw = torch.empty(n, device="cuda")
            dw.fill_(7.0)
            assert hip.hipMemsetAsync(dw.data_ptr(), 0, n * 4, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            dw.add_(x).add_(x)
            outs[i].copy_(dw)
            if keep_alive:
                keep.append(dw)
            del dw
    launch = torch.cuda.Stream() if on_stream else torch.cuda.current_stream()
    launch.wait_stream(torch.cuda.current_stream())
    for r in range(4):
      with torch.cuda.stream(launch):
        outs.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        vals = [sorted(set(outs[i].tolist()))[:3] for i in range(ROUNDS)]
        print(f"keep_alive={keep_alive} replayed on {'a side stream' if on_stream else 'the default (null) stream'}, replay {r}: " + " ".join(("ok" if v == [2.0] else str(v)) for v in vals), flush=True)
    del g
