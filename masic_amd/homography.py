"""3x3 homography preparation for the perspective-warp kernel.

kornia 0.5.0's warp_perspective (the reference's MASIC.py:638,644,781,821,833 calls) turns the pixel-space
homography M into  inverse(N_dst @ (M @ inverse(N_src)))  with float32 torch ops before sampling.  That
float32 chain is ill-conditioned in the perspective row (entries ~1e-5 next to translations of several
pixels): two correct float32 evaluations differ by up to ~1e-3 px at 192 px width and more at full
resolution, which shows at the bilinear mask borders.  To stay bit-compatible with the reference's CPU
path the default evaluates exactly that chain, with the same torch float32 calls, on the host (B x 9
floats; one small D2H/H2D per forward, before any kernel of the forward is launched).
`MASIC_WARP_MATRIX=device` selects the float64 device kernel (masic_warp_matrix) instead: no host
round trip, more accurate than the reference, not bit-compatible with it.
"""
import os

import torch

from . import ops


def _normal_transform_pixel(h, w):
    wd = 1e-14 if w == 1 else float(w - 1)
    hd = 1e-14 if h == 1 else float(h - 1)
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    t[0, 0] = t[0, 0] * 2.0 / wd
    t[1, 1] = t[1, 1] * 2.0 / hd
    return t.unsqueeze(0)


def warp_matrices(h_matrix, src_hw, dst_hw, want_inverse=False, device=False):
    """Returns the normalised sampling matrix for warp(., h_matrix) and, if asked, for
    warp(., inverse(h_matrix)) (the second warp of MASIC.py:644), as [B,3,3] float32 device tensors.
    device=True: the float64 device kernel (no host round trip, hence no stream synchronisation in front of the forward)."""
    if device or os.environ.get("MASIC_WARP_MATRIX", "host") == "device":
        fwd = ops.warp_matrix(h_matrix.contiguous(), src_hw, dst_hw)
        back = ops.warp_matrix(h_matrix.contiguous(), src_hw, dst_hw, invert_first=True) if want_inverse else None
        return fwd, back
    if not h_matrix.is_cuda:
        raise RuntimeError("masic_amd: h_matrix must be a CUDA (HIP) tensor -- the MI355X path has no CPU fallback")
    fwd, back = warp_matrices_host(h_matrix.detach().to("cpu", torch.float32), src_hw, dst_hw, want_inverse)
    return fwd.to(h_matrix.device), (back.to(h_matrix.device) if want_inverse else None)


def warp_matrices_host(m, src_hw, dst_hw, want_inverse=False):
    """The float32 chain itself on a CPU tensor [B,3,3]; returns CPU tensors (callers that pipeline the host step with
    device work -- masic_amd/graph.py -- do their own transfers)."""
    n_src = _normal_transform_pixel(*src_hw)
    n_dst = _normal_transform_pixel(*dst_hw)

    def chain(mat):
        return torch.inverse(n_dst @ (mat @ torch.inverse(n_src))).contiguous()

    return chain(m), (chain(torch.inverse(m)) if want_inverse else None)
