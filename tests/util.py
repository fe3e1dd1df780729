"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: 1e-4 relative on float32 outputs
RTOL = 1e-4


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def golden_state_dict(fx, template_sd):
    """Reference parameters stored in a fixture ('sd/<name>') over a template state dict's buffers."""
    sd = {k: v.clone() for k, v in template_sd.items()}
    for k in fx:
        if k.startswith("sd/"):
            sd[k[3:]] = torch.from_numpy(fx[k])
    return sd


def rel_err(a, b):
    """max |a-b| scaled by the largest magnitude of the reference tensor."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = float(b.abs().max())
    return float((a - b).abs().max()) / (scale if scale > 0 else 1.0)


def assert_close(a, b, name="", rtol=RTOL):
    assert tuple(a.shape) == tuple(b.shape), f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= rtol, f"{name}: relative error {e:.3e} > {rtol:.1e}"
    return e


def tie_zone(y, eps=1e-4):
    """latents within eps of a rounding boundary (SURVEY.md 7.3 policy (b))."""
    y = y.detach().double().cpu()
    return ((y - torch.floor(y)) - 0.5).abs() < eps


def assert_symbols(sym_hip, sym_ref, y_ref, name="", eps=1e-4):
    """Bit-exact except inside the declared tie zone |frac(y) - 1/2| < eps; returns #mismatches."""
    sym_hip = sym_hip.cpu().to(torch.int64)
    sym_ref = sym_ref.cpu().to(torch.int64)
    bad = sym_hip != sym_ref
    outside = bad & ~tie_zone(y_ref, eps)
    assert int(outside.sum()) == 0, f"{name}: {int(outside.sum())} symbol mismatches outside the tie zone"
    assert int((sym_hip - sym_ref).abs().max()) <= 1, f"{name}: symbol off by more than one"
    return int(bad.sum())
