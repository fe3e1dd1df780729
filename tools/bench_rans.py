"""Host rANS coder throughput (this build's C ABI through numpy) beside the reference's pybind11 extension (lists), same
symbols / tables (tests/golden/rans_vectors.npz 'long' case, repeated).  Build container only (needs oracle/_ref)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from masic_amd import rans
G = np.load(os.path.join(ROOT, "tests", "golden", "rans_vectors.npz"))
tab, sizes, offs = G["tables"], G["sizes"], G["offsets"]
sym, idx = np.tile(G["sym_long"], 20), np.tile(G["idx_long"], 20)          # 1M symbols
t0 = time.perf_counter(); enc = rans.encode_with_indexes(sym, idx, tab, sizes, offs); t1 = time.perf_counter()
dec = rans.decode_with_indexes(enc, idx, tab, sizes, offs); t2 = time.perf_counter()
assert dec.tolist() == sym.tolist()
print(f"this build : encode {sym.size / (t1 - t0) / 1e6:6.1f} Msym/s  decode {sym.size / (t2 - t1) / 1e6:6.1f} Msym/s  ({len(enc)} bytes)")
try:
    from oracle import ref_import
    ref_import.load()
    from compressai import ans
    rows = [tab[i, :sizes[i]].tolist() for i in range(len(sizes))]
    s, i = sym.tolist(), idx.tolist()
    t0 = time.perf_counter(); e2 = ans.RansEncoder().encode_with_indexes(s, i, rows, sizes.tolist(), offs.tolist()); t1 = time.perf_counter()
    d2 = ans.RansDecoder().decode_with_indexes(e2, i, rows, sizes.tolist(), offs.tolist()); t2 = time.perf_counter()
    assert e2 == enc
    print(f"reference  : encode {sym.size / (t1 - t0) / 1e6:6.1f} Msym/s  decode {sym.size / (t2 - t1) / 1e6:6.1f} Msym/s  (pybind11 list marshaling included, as its callers pay it)")
except Exception as e:
    print("reference extension not available here:", e)
