"""Generates tests/golden/misc_api.npz from the REFERENCE implementation (run in the build container only): the API-surface classes
MASIC itself does not call -- GaussianConditional (entropy_models.py:433-562: update tables, build_indexes, eval forward) and GDN1
(layers/gdn.py:95-121).  Data only.

    python tests/golden/make_misc_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

ref_import.load()
from compressai.entropy_models import GaussianConditional  # noqa: E402  (the reference's: ref_import put its package in sys.modules)
from compressai.layers import GDN1  # noqa: E402

fx = {}
table = [0.11, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0]
gc = GaussianConditional(table)
gc.update()
fx["gc/scale_table"] = np.array(table, dtype=np.float32)
fx["gc/quantized_cdf"] = gc._quantized_cdf.numpy().astype(np.int32)
fx["gc/offset"] = gc._offset.numpy().astype(np.int32)
fx["gc/cdf_length"] = gc._cdf_length.numpy().astype(np.int32)
g = torch.Generator().manual_seed(4)
x = 6 * torch.randn(2, 5, 7, 9, generator=g)
scales = torch.rand(2, 5, 7, 9, generator=g) * 3            # part below the 0.11 bound
means = torch.randn(2, 5, 7, 9, generator=g)
gc.eval()
with torch.no_grad():
    y0, l0 = gc(x, scales)
    y1, l1 = gc(x, scales, means)
    idx = gc.build_indexes(scales)
fx.update({"gc/x": x.numpy(), "gc/scales": scales.numpy(), "gc/means": means.numpy(), "gc/y_nomeans": y0.numpy(), "gc/lik_nomeans": l0.numpy(),
           "gc/y_means": y1.numpy(), "gc/lik_means": l1.numpy(), "gc/indexes": idx.numpy().astype(np.int32)})
for inverse in (False, True):
    m = GDN1(6, inverse=inverse)
    with torch.no_grad():
        m.beta.mul_(1 + 0.2 * torch.rand(6, generator=g))
        m.gamma.add_(0.05 * torch.rand(6, 6, generator=g))
        xin = 2 * torch.randn(2, 6, 5, 11, generator=g)
        out = m(xin)
    tag = "gdn1_inv/" if inverse else "gdn1/"
    fx.update({tag + "beta": m.beta.detach().numpy(), tag + "gamma": m.gamma.detach().numpy(), tag + "x": xin.numpy(), tag + "y": out.numpy()})
np.savez_compressed(os.path.join(HERE, "misc_api.npz"), **fx)
print({k: v.shape for k, v in fx.items()})
