"""Generates tests/golden/hsic_tiny_codec.npz from the REFERENCE's own HSIC.compress (run in the build container only).

    python tests/golden/make_codec_goldens.py

The reference's MASIC.py is imported on CPU through oracle/ref_import.py and its compress() (MASIC.py:855-1158) is run on the
weights and inputs of tests/golden/hsic_tiny.npz.  Three things of the environment are substituted, none of them arithmetic of
the path:
  * `range_coder.RangeEncoder` (third-party, absent) by a recorder of the (symbol, cdf) pairs compress() hands to it -- the
    tables and symbols are what the fixture pins; the coder's bytes stay unpinned;
  * the hard-coded `.to('cuda:0')` of the symbol grid (:989, :1075) by a no-op (no GPU in the container);
  * `np.int` (removed from numpy) by `int`.
Stored: the bytes of the `.npz` header file the reference wrote (picture size, z string lengths, minmax, channel flags, the z
strings), and per view the coding tables [n][2*minmax+2] and symbols [n] in the reference's order (pixel raster, then non-zero
channel), plus minmax and the non-zero channel indices.  Data only: no reference source text is written anywhere."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402

R = ref_import.load()
if not hasattr(np, "int"):
    np.int = int

fx = np.load(os.path.join(HERE, "hsic_tiny.npz"))
N, M, K = (int(v) for v in fx["NMK"])
net = R.HSIC(N=N, M=M, K=K)
sd = net.state_dict()
for k in fx.files:
    if k.startswith("sd/"):
        sd[k[3:]] = torch.from_numpy(fx[k])
net.load_state_dict(sd)
net.eval()
net.update(force=True)

records = []


class Recorder:
    def __init__(self, path):
        self.path = path
        open(path, "wb").close()          # compress() stats the file afterwards (:1140)

    def encode(self, symbols, cdf):
        records.append((int(symbols[0]), np.asarray(cdf, dtype=np.int64)))

    def close(self):
        pass


R.RangeEncoder = Recorder
_to = torch.Tensor.to


def to_cpu(self, *a, **k):
    if a and isinstance(a[0], str) and a[0].startswith("cuda"):
        return self
    return _to(self, *a, **k)


torch.Tensor.to = to_cpu
x1, x2, hm = (torch.from_numpy(fx[k]) for k in ("x1", "x2", "h_matrix"))
tmp = tempfile.mkdtemp()
try:
    with torch.no_grad():
        net.compress(x1, x2, hm, "pair", output_path=tmp, device="cpu")
finally:
    torch.Tensor.to = _to
header = np.frombuffer(open(os.path.join(tmp, "pair.npz"), "rb").read(), dtype=np.uint8)

# split the records into the two views: header layout (:916-948) gives minmax and the channel flags
raw = header.tobytes()
n1, mm1 = (int(v) for v in np.frombuffer(raw[4:8], dtype=np.uint16))
flag1 = np.unpackbits(np.frombuffer(raw[8:8 + M // 8], dtype=np.uint8))
off = 8 + M // 8 + n1
n2, mm2 = (int(v) for v in np.frombuffer(raw[off:off + 4], dtype=np.uint16))
flag2 = np.unpackbits(np.frombuffer(raw[off + 4:off + 4 + M // 8], dtype=np.uint8))
h, w = x1.shape[-2] // 16, x1.shape[-1] // 16
c1 = int(flag1.sum()) * h * w
assert len(records) == c1 + int(flag2.sum()) * h * w, (len(records), c1)
out = {"header": header, "minmax": np.array([mm1, mm2]), "nz1": np.flatnonzero(flag1), "nz2": np.flatnonzero(flag2), "hw": np.array([h, w])}
for name, recs in (("y1", records[:c1]), ("y2", records[c1:])):
    out[name + "_sym"] = np.array([r[0] for r in recs], dtype=np.int64)
    out[name + "_cdf"] = np.stack([r[1] for r in recs])
np.savez_compressed(os.path.join(HERE, "hsic_tiny_codec.npz"), **out)
print("records", len(records), "minmax", mm1, mm2, "nonzero channels", int(flag1.sum()), int(flag2.sum()),
      "cdf totals", sorted(set(int(r[1][-1]) for r in records))[:6], "header bytes", header.size)
