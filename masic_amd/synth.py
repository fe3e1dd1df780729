"""Deterministic synthetic weights and stereo inputs.

Everything comes from numpy's frozen legacy MT19937 stream (`numpy.random.RandomState`), so the
build container and the GPU box regenerate identical float32 tensors without depending on torch's
RNG.  Weights are kaiming-normal shaped, with per-layer gains chosen (once, by measurement on the
CPU oracle) so that the latents are non-degenerate: round(y) spans roughly +-15, z spans a few
units, sigma straddles the 0.11 bound and the mixture weights are not uniform.  With torch's default
initialisation every latent rounds to zero, which would make symbol parity vacuous.
"""
import math

import numpy as np
import torch

# multiplicative gains on top of kaiming-normal std = sqrt(2 / fan_in)
_GAINS = {
    "g_a_conv4": 6.0,            # spread of y
    "encode_hyper.4": 0.7,       # spread of z
    "pre_conv": 0.6,
    "g_s_conv1": 0.08,           # keep the IGDN chain of the synthesis transforms from blowing up
    "g_s_conv2": 0.5,
    "g_s_conv3": 0.5,
    "g_s_conv4": 0.5,
    "after_conv": 0.5,
    "gmm_sigma.4": 0.5,
    "gmm_means.4": 0.5,
    "gmm_weights.4": 1.5,
    "maskconv": 1.5,
    ".RB": 0.35,                 # CQE residual blocks: keep 9 stacked blocks per view at O(1)
}


def _gain(name):
    for k, g in _GAINS.items():
        if k in name:
            return g
    return 1.0


def synth_tensor(name, shape, rs):
    """One state-dict entry by name/shape. `rs`: numpy RandomState."""
    leaf = name.split(".")[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("beta"):                       # GDN beta (reparametrised): sqrt(1 + 2^-36) * (1 + jitter)
        v = np.sqrt(1.0 + 2.0 ** -36) * (1.0 + 0.1 * rs.uniform(-1, 1, size=shape))
    elif name.endswith("gamma"):                    # GDN gamma (reparametrised): sqrt(0.1*I + |noise| + 2^-36)
        C = shape[0]
        g = 0.1 * np.eye(C) + 0.02 * np.abs(rs.standard_normal(size=shape)) / math.sqrt(C / 3.0)
        v = np.sqrt(g + 2.0 ** -36)
    elif leaf == "weight" and len(shape) == 4:
        # fan_in as torch computes it: dim 1 * receptive field (also for ConvTranspose2d)
        fan_in = shape[1] * shape[2] * shape[3]
        v = rs.standard_normal(size=shape) * math.sqrt(2.0 / fan_in) * _gain(name)
    elif leaf == "bias":
        v = 0.05 * rs.standard_normal(size=shape)
        if "gmm_sigma.4" in name:
            v = v + 2.0
    elif "_matrices" in name:
        filters = (1, 3, 3, 3, 3, 1)
        i = int(leaf)
        init = math.log(math.expm1(1 / (10 ** 0.2) / filters[i + 1]))
        v = init + 0.2 * rs.standard_normal(size=shape)
    elif "_biases" in name:
        v = rs.uniform(-0.5, 0.5, size=shape)
    elif "_factors" in name:
        v = 0.3 * rs.standard_normal(size=shape)
    elif leaf == "quantiles":
        med = 0.4 * rs.standard_normal(size=(shape[0], 1))
        v = np.concatenate([med - 10.0, med, med + 10.0], axis=1).reshape(shape)
    else:
        raise KeyError(name)
    assert v.size == n
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape))


def synth_state_dict(reference_like_state_dict, seed=0):
    """Fills every float parameter of an HSIC-shaped state dict (buffers are left as constructed).
    Each tensor gets its own stream seeded by (seed, crc of its name): insertion order does not matter."""
    import zlib
    out = {}
    for name, t in reference_like_state_dict.items():
        leaf = name.split(".")[-1]
        is_param = (leaf in ("weight", "bias", "beta", "gamma", "quantiles")
                    or "_matrices" in name or "_biases" in name or "_factors" in name)
        if not is_param:
            out[name] = t.clone()
            continue
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
        out[name] = synth_tensor(name, tuple(t.shape), rs)
        if "context_prediction" in name and leaf == "weight":
            kh, kw = t.shape[-2:]
            out[name][:, :, kh // 2, kw // 2:] = 0      # what the reference's in-place masking leaves in checkpoints
            out[name][:, :, kh // 2 + 1:] = 0
    return out


def _smooth_field(rs, C, H, W, cell):
    """Band-limited field in [0,1): coarse uniform noise, bilinearly upsampled (numpy only)."""
    gh, gw = H // cell + 2, W // cell + 2
    coarse = rs.uniform(0, 1, size=(C, gh, gw))
    ys = (np.arange(H) + 0.5) / cell
    xs = (np.arange(W) + 0.5) / cell
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[None, :, None]; fx = (xs - x0)[None, None, :]
    a = coarse[:, y0][:, :, x0]; b = coarse[:, y0][:, :, x0 + 1]
    c = coarse[:, y0 + 1][:, :, x0]; d = coarse[:, y0 + 1][:, :, x0 + 1]
    return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)


def synth_inputs(B, H, W, seed=0):
    """Returns (x1, x2, h_matrix): float32 [B,3,H,W] x2 and [B,3,3].
    x1: smooth image content (two octaves); x2: x1 shifted by the integer translation of h_matrix plus
    mild noise, so the cross-view branch carries signal; h_matrix: near-identity homography with a few
    pixels of translation and perspective terms ~1e-5 (magnitudes of SURVEY.md section 8d)."""
    rs = np.random.RandomState(1234567 + seed)
    x1 = 0.75 * _smooth_field(rs, B * 3, H, W, 16) + 0.25 * _smooth_field(rs, B * 3, H, W, 4)
    x1 = x1.reshape(B, 3, H, W)
    hm = np.zeros((B, 3, 3))
    x2 = np.empty_like(x1)
    for b in range(B):
        tx = int(rs.randint(-int(0.03 * W) - 1, int(0.03 * W) + 2))
        ty = int(rs.randint(-int(0.01 * H) - 1, int(0.01 * H) + 2))
        a, a2 = rs.uniform(-0.02, 0.02, size=2)
        s1, s2 = rs.uniform(-0.01, 0.01, size=2)
        p1, p2 = rs.uniform(-2e-5, 2e-5, size=2)
        hm[b] = [[1 + a, s1, tx], [s2, 1 + a2, ty], [p1, p2, 1.0]]
        x2[b] = np.roll(x1[b], shift=(ty, tx), axis=(1, 2))
    x2 = np.clip(x2 + 0.02 * rs.standard_normal(size=x2.shape), 0.0, 1.0)
    f32 = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return f32(x1), f32(x2), f32(hm)


def synth_noise(B, N, M, H, W, seed=0):
    """The seven U(-1/2,1/2) draws of a training-mode forward, keyed as oracle.hsic_oracle.NOISE_KEYS, in the
    layouts the reference draws them (EB: (C,1,h*w*B); GMM: [B,M,h,w])."""
    rs = np.random.RandomState(7654321 + seed)
    h, w = H // 16, W // 16
    hz, wz = h // 4, w // 4
    u = lambda *s: torch.from_numpy(rs.uniform(-0.5, 0.5, size=s).astype(np.float32))
    return {"z1": u(N, 1, hz * wz * B), "y1_ctx": u(B, M, h, w), "y1": u(B, M, h, w), "z2": u(N, 1, hz * wz * B),
            "y2_ctx": u(B, M, h, w), "y1_warp": u(B, M, h, w), "y2": u(B, M, h, w)}
