"""Golden vectors for the host-side entropy coder, produced by the REFERENCE's own code in the build
container: its pybind11 extensions compiled from the reference sources into oracle/_ref/ (oracle/Makefile) and its
EntropyBottleneck (imported through oracle/ref_import.py).  Run:  python tests/golden/make_rans_goldens.py
Output: tests/golden/rans_vectors.npz (inputs and expected outputs only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402


def main():
    ref_import.load()                          # leaves the reference's `compressai` (with oracle/_ref on its path) in sys.modules
    from compressai import _CXX as cxx, ans
    rs = np.random.RandomState(2024)
    out = {}
    # ---- pmf_to_quantized_cdf: smooth, spiky, many near-zero entries (forces the frequency stealing), short, long
    pmfs = []
    for n, kind in ((5, "flat"), (17, "gauss"), (40, "spiky"), (64, "tiny"), (3, "one"), (120, "gauss"), (33, "zeros")):
        x = np.arange(n, dtype=np.float64)
        if kind == "flat":
            p = np.ones(n)
        elif kind == "gauss":
            p = np.exp(-0.5 * ((x - n / 2.3) / (n / 9.0)) ** 2)
        elif kind == "spiky":
            p = rs.rand(n) ** 8
        elif kind == "tiny":
            p = np.full(n, 1e-7); p[n // 2] = 1.0
        elif kind == "one":
            p = np.array([0.0, 1.0, 0.0])
        else:
            p = rs.rand(n); p[rs.rand(n) < 0.5] = 0.0; p[0] = 0.3
        p = (p / p.sum()).astype(np.float32)
        pmfs.append(p)
    for prec in (16, 12):
        for i, p in enumerate(pmfs):
            if prec == 12 and p.size > 60:
                continue
            out[f"pmf{prec}_{i}"] = p
            out[f"cdf{prec}_{i}"] = np.asarray(cxx.pmf_to_quantized_cdf(p.tolist(), prec), dtype=np.int64)
    # ---- rANS streams: several tables of different lengths, symbols inside, at and far outside the tables (bypass mode)
    ntab = 6
    tables, sizes, offsets = [], [], []
    for t in range(ntab):
        n = int(rs.randint(3, 50))
        p = rs.rand(n) ** 3 + 1e-4
        p = (p / p.sum()).astype(np.float32)
        cdf = cxx.pmf_to_quantized_cdf(p.tolist(), 16)
        tables.append(cdf)
        sizes.append(len(cdf))
        offsets.append(int(rs.randint(-20, 5)))
    width = max(sizes)
    tab = np.zeros((ntab, width), dtype=np.int32)
    for t, c in enumerate(tables):
        tab[t, :len(c)] = c
    out["tables"], out["sizes"], out["offsets"] = tab, np.asarray(sizes, np.int32), np.asarray(offsets, np.int32)
    for case, n in (("short", 7), ("mid", 1000), ("long", 50000), ("escapes", 400)):
        idx = rs.randint(0, ntab, size=n).astype(np.int32)
        sym = np.empty(n, dtype=np.int32)
        for k in range(n):
            m = sizes[idx[k]] - 2
            r = rs.rand()
            if case == "escapes" or r < 0.03:
                sym[k] = offsets[idx[k]] + int(rs.choice([-1, -2, -17, -300, -70000, m, m + 1, m + 15, m + 16, m + 4000, m + 1234567]))
            else:
                sym[k] = offsets[idx[k]] + int(rs.randint(0, max(m, 1)))
        enc = ans.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), [list(map(int, c)) for c in tables], sizes, offsets)
        dec = ans.RansDecoder().decode_with_indexes(enc, idx.tolist(), [list(map(int, c)) for c in tables], sizes, offsets)
        assert dec == sym.tolist()
        out[f"sym_{case}"], out[f"idx_{case}"], out[f"enc_{case}"] = sym, idx, np.frombuffer(enc, dtype=np.uint8)
    # ---- the reference EntropyBottleneck: tables after update(), streams of compress(), result of decompress()
    from compressai.entropy_models import EntropyBottleneck  # the reference's
    torch.manual_seed(7)
    C = 12
    eb = EntropyBottleneck(C)
    with torch.no_grad():
        for p in list(eb._matrices) + list(eb._biases) + list(eb._factors):
            p.add_(0.3 * torch.randn_like(p))
        eb.quantiles[:, 0, 0] = -torch.rand(C) * 25 - 1
        eb.quantiles[:, 0, 1] = torch.randn(C) * 2
        eb.quantiles[:, 0, 2] = torch.rand(C) * 30 + 3
    eb.update(force=True)
    x = torch.randn(2, C, 5, 7) * 12
    x[0, 0, 0, 0] = 300.0                     # far outside every table
    x[1, 3, 2, 2] = -250.0
    strings = eb.compress(x)
    # the reference's decompress accepts one stream at a time only (its means/indexes shape check, entropy_models.py:221-224)
    xh = torch.cat([eb.decompress([s], x.shape[-2:]) for s in strings])
    for k, v in eb.state_dict().items():
        out["eb_state/" + k] = v.numpy()
    out["eb_x"], out["eb_xhat"] = x.numpy(), xh.numpy()
    for i, s in enumerate(strings):
        out[f"eb_string_{i}"] = np.frombuffer(s, dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "rans_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
