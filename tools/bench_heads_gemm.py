#!/usr/bin/env python3
"""The 1x1 layers of the entropy-parameter heads (reference MASIC.py:330-468) as launched by the forward: layer i of the three stacks of
a head in ONE grouped launch (masic_gemm_f16k_group_fwd).  Checks every grouped launch against float32 matmuls on bf16-rounded
operands, and prints the algorithmic bytes / FLOPs of each launch; device times come from a kernel trace of this script
(tools/scratch/prof.sh tools/bench_heads_gemm.py, or rocprofv3 --kernel-trace --stats).  MASIC_GEMM_V2=0 selects the round-2 kernel."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from masic_amd import ops  # noqa: E402

torch.manual_seed(0)
B, H, W = int(os.environ.get("GB", "8")), 32, 32
M, K = 192, 5
# (Cin, [Cout of sigma, means, weights]) of the three layers of head 1 (768-channel input) and head 2 (960)
LAYERS = {"head1": [(4 * M, [6 * M] * 3), (6 * M, [4 * M, 4 * M, K * M]), (4 * M, [K * M] * 2 + [K * M])],
          "head2": [(5 * M, [6 * M] * 3), (6 * M, [4 * M, 4 * M, K * M]), (4 * M, [K * M] * 2 + [K * M])]}
LAYERS["head1"][2] = (None, None)     # inputs differ per stack in layers 1, 2: built below
dev = "cuda"


def q(t):
    return t.bfloat16().float()


def run(name, cin_list, cout_list, shared_input):
    xs, layers, refs = [], [], []
    x0 = torch.randn(B, cin_list[0], H, W, device=dev)
    for i, (cin, cout) in enumerate(zip(cin_list, cout_list)):
        x = x0 if (shared_input or i == 0) else torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        b = torch.randn(cout, device=dev)
        xf = ops.nchw_to_f16k(x)
        layers.append({"x": xf, "wp": ops.pack_gemm_f16k_weight(w, cin, cout, False), "bias": b, "Cin": cin, "Cout": cout, "act": ops.ACT_LEAKY, "out": "f16k"})
        refs.append(torch.nn.functional.leaky_relu(torch.einsum("oc,bchw->bohw", q(w), q(x)) + b.view(1, -1, 1, 1), 0.01))
        xs.append(x)
    outs = ops.gemm_f16k_group(layers, B, H, W)
    worst = 0.0
    for o, r, L in zip(outs, refs, layers):
        y = ops.f16k_to_nchw(o, B, L["Cout"], H, W)
        worst = max(worst, float((y - q(r)).abs().max() / r.abs().max()))
    # single-layer launches must give the same bits
    same = all(torch.equal(ops.gemm_f16k(L["x"], L["wp"], L["bias"], B, L["Cin"], L["Cout"], H, W, L["act"]), o) for L, o in zip(layers, outs))
    nchw = ops.gemm_f16k_group([dict(L, out="nchw") for L in layers], B, H, W)
    worst32 = max(float((y - r).abs().max() / r.abs().max()) for y, r in zip(nchw, refs))
    for _ in range(20):
        ops.gemm_f16k_group(layers, B, H, W)
    torch.cuda.synchronize()
    px = B * H * W
    flops = sum(2.0 * px * L["Cin"] * L["Cout"] for L in layers)
    act_bytes = (px * cin_list[0] * 2) if shared_input else sum(px * c * 2 for c in cin_list)
    alg = act_bytes + sum(L["Cin"] * L["Cout"] * 2 + px * L["Cout"] * 2 for L in layers)
    print(f"{name}: {flops / 1e9:6.1f} GFLOP, algorithmic bytes {alg / 1e6:6.1f} MB | F16K out err {worst:.1e} (bf16 rounding), NCHW out err {worst32:.1e}, "
          f"grouped == single-layer launches: {same}")


run("head1.layer0 768->3x1152", [768] * 3, [1152] * 3, True)
run("head1.layer1 1152->768,768,960", [1152] * 3, [768, 768, 960], False)
run("head1.layer2 768,768,960->960", [768, 768, 960], [960] * 3, False)
run("head2.layer0 960->3x1152", [960] * 3, [1152] * 3, True)
run("ragged 48->96 (B=%d, one group)" % B, [48], [96], True)
