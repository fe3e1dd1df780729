"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors that were produced by the
reference itself (tests/golden/make_goldens.py), host logic of the API mirror, and the C-ABI exports."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import hsic_oracle as O
from tests.util import GOLDEN, golden_state_dict, load_npz

import MASIC  # the product's mirror (coremasic/mywork on sys.path via conftest)
from masic_amd import synth


def _outputs_match(out, fx, prefix, tol):
    for k in ("x1_hat", "x2_hat", "y1_hat", "z1_hat", "x1_mask_R", "x1_mask_L"):
        d = float((out[k].detach() - torch.from_numpy(fx[prefix + k])).abs().max())
        assert d <= tol, (k, d)
    for k, v in out["likelihoods"].items():
        d = float((v.detach() - torch.from_numpy(fx[prefix + "lik_" + k])).abs().max())
        assert d <= tol, ("lik_" + k, d)


def test_pin_report_says_oracle_equals_reference():
    rep = json.load(open(os.path.join(GOLDEN, "pin_report.json")))
    for case in ("tiny_eval", "tiny_train", "small_eval", "c1_eval"):
        assert max(rep[case].values()) <= 1e-6, (case, rep[case])
    assert rep["tiny_train_grad_maxrel"] <= 1e-5
    assert rep["tiny_aux_loss"] <= 1e-4


def test_oracle_tiny_eval_and_symbols_vs_golden():
    fx = load_npz("hsic_tiny.npz")
    N, M, K = (int(v) for v in fx["NMK"])
    sd = golden_state_dict(fx, MASIC.HSIC(N, M, K).state_dict())
    x1, x2, H = (torch.from_numpy(fx[k]) for k in ("x1", "x2", "h_matrix"))
    with torch.no_grad():
        out = O.hsic_forward(sd, x1, x2, H, K=K, keep=True)
    _outputs_match(out, fx, "eval/", 1e-6)
    sym = O.symbols(out["_aux"], sd)
    for k in ("y1", "y2", "z1", "z2"):
        assert np.array_equal(sym[k].numpy(), fx["eval/sym_" + k]), k
    loss = O.rd_loss(out, x1, x2, float(fx["lmbda"]))
    for k in ("bpp_loss", "mse_loss", "loss", "psnr1", "psnr2"):
        assert abs(float(loss[k]) - float(fx["eval/loss_" + k])) <= 1e-5 * max(1.0, abs(float(fx["eval/loss_" + k]))), k


def test_oracle_tiny_train_and_gradients_vs_golden():
    fx = load_npz("hsic_tiny.npz")
    N, M, K = (int(v) for v in fx["NMK"])
    net = MASIC.HSIC(N, M, K)
    sd = golden_state_dict(fx, net.state_dict())
    pnames = {n for n, _ in net.named_parameters()}
    sd = {k: (v.requires_grad_(True) if k in pnames else v) for k, v in sd.items()}
    x1, x2, H = (torch.from_numpy(fx[k]) for k in ("x1", "x2", "h_matrix"))
    noise = {k: torch.from_numpy(fx["train/noise_" + k]) for k in O.NOISE_KEYS}
    out = O.hsic_forward(sd, x1, x2, H, K=K, training=True, noise=noise)
    _outputs_match(out, fx, "train/", 1e-6)
    O.rd_loss(out, x1, x2, float(fx["lmbda"]))["loss"].backward()
    worst = 0.0
    for k in fx:
        if not k.startswith("train/grad/"):
            continue
        g = torch.from_numpy(fx[k])
        og = sd[k[len("train/grad/"):]].grad
        og = torch.zeros_like(g) if og is None else og
        worst = max(worst, float((g - og).abs().max()) / (float(g.abs().max()) + 1e-30))
    assert worst <= 1e-5, worst
    aux = O.eb_aux_loss(sd, "entropy_bottleneck1") + O.eb_aux_loss(sd, "entropy_bottleneck2")
    assert abs(float(aux) - float(fx["train/aux_loss"])) <= 1e-4 * abs(float(fx["train/aux_loss"]))


def test_oracle_small_ragged_vs_golden():
    fx = load_npz("hsic_small.npz")
    N, M, K = (int(v) for v in fx["NMK"])
    B, H, W = (int(v) for v in fx["BHW"])
    seed = int(fx["seed"])
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=seed)
    x1, x2, hm = synth.synth_inputs(B, H, W, seed=seed)
    with torch.no_grad():
        out = O.hsic_forward(sd, x1, x2, hm, K=K, keep=True)
    _outputs_match(out, fx, "eval/", 1e-6)
    sym = O.symbols(out["_aux"], sd)
    for k in sym:
        assert np.array_equal(sym[k].numpy(), fx["eval/sym_" + k]), k


def test_oracle_config1_digest():
    """BASELINE config 1 (1x3x256x256, N=128, M=192, K=5): scalars, symbol hashes, sampled values."""
    dg = json.load(open(os.path.join(GOLDEN, "hsic_c1_digest.json")))
    N, M, K = dg["NMK"]
    B, H, W = dg["BHW"]
    sd = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=dg["seed"])
    x1, x2, hm = synth.synth_inputs(B, H, W, seed=dg["seed"])
    with torch.no_grad():
        out = O.hsic_forward(sd, x1, x2, hm, K=K, keep=True)
    sym = O.symbols(out["_aux"], sd)
    for k, h in dg["sym_sha256"].items():
        assert hashlib.sha256(sym[k].numpy().astype(np.int32).tobytes()).hexdigest() == h, k
    loss = O.rd_loss(out, x1, x2, dg["lmbda"])
    for k in ("bpp_loss", "mse_loss", "loss", "psnr1", "psnr2"):
        assert abs(float(loss[k]) - dg["loss"][k]) <= 1e-5 * max(1.0, abs(dg["loss"][k])), k
    flat = {k: out[k] for k in ("x1_hat", "x2_hat", "y1_hat", "z1_hat", "x1_mask_R", "x1_mask_L")}
    flat.update({"lik_" + k: v for k, v in out["likelihoods"].items()})
    for k, s in dg["samples"].items():
        got = flat[k].reshape(-1)[torch.tensor(s["index"])]
        assert float((got - torch.tensor(s["value"])).abs().max()) <= 1e-6 * max(1.0, s["absmax"]), k


def test_warp_identity_and_translation_properties():
    """Size-independent properties of the (parity-unpinned) warp restatement."""
    x = torch.rand(2, 3, 40, 56)
    eye = torch.eye(3).repeat(2, 1, 1)
    assert float((O.warp_perspective(x, eye, (40, 56)) - x).abs().max()) < 1e-4
    t = eye.clone()
    t[:, 0, 2] = 3.0   # dst(x) = src(x - 3)
    w = O.warp_perspective(x, t, (40, 56))
    assert float((w[..., 3:] - x[..., :-3]).abs().max()) < 1e-4
    assert float(w[..., :2].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ host logic
def test_state_dict_keys_match_reference():
    keys = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in MASIC.HSIC(128, 192, 5).state_dict().items()]
    assert mine == keys["HSIC_128_192_5"]
    en = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in MASIC.Independent_EN().state_dict().items()]
    assert en == keys["Independent_EN"]


def test_parameter_split_matches_reference():
    keys = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    net = MASIC.HSIC(16, 24, 3)
    assert len(list(net.parameters())) == keys["HSIC_parameters_count"] == 136
    assert len(list(net.aux_parameters())) == keys["HSIC_aux_parameters_count"] == 30
    aux_ids = {id(p) for p in net.aux_parameters()}
    assert all(id(p) not in aux_ids for p in net.parameters())
    assert len(list(net.named_parameters())) == 166


def test_module_surface():
    import compressai
    from compressai.entropy_models import EntropyBottleneck, GaussianMixtureConditional_gf
    from compressai.layers import GDN, MaskedConv2d
    from compressai.models import CompressionModel  # noqa: F401
    from compressai.models.utils import conv, deconv
    assert compressai.get_entropy_coder() == "ans" and "ans" in compressai.available_entropy_coders()
    with pytest.raises(ValueError):
        compressai.set_entropy_coder("nope")
    c, d = conv(3, 8), deconv(8, 3)
    assert isinstance(c, torch.nn.Conv2d) and c.padding == (2, 2) and c.stride == (2, 2)
    assert isinstance(d, torch.nn.ConvTranspose2d) and d.output_padding == (1, 1)
    m = MaskedConv2d(4, 8, kernel_size=5, padding=2, stride=1)
    assert int(m.mask[0, 0].sum()) == 12 and float(m.mask[0, 0, 2, 2]) == 0.0
    g = GDN(4)
    assert torch.allclose(g.gamma_reparam(g.gamma) if False else g.gamma ** 2 - 2.0 ** -36, 0.1 * torch.eye(4), atol=1e-6)
    eb = EntropyBottleneck(6)
    assert [tuple(p.shape) for p in eb._matrices] == [(6, 3, 1), (6, 3, 3), (6, 3, 3), (6, 3, 3), (6, 1, 3)]
    assert GaussianMixtureConditional_gf(K=5).K == 5
    for name in ("HSIC", "Independent_EN", "GMM_together", "mask", "RateDistortionLoss", "AverageMeter", "Encoder1",
                 "Decoder2", "mask2weights", "encode_hyper", "gmm_hyper_y1_same_resolution"):
        assert hasattr(MASIC, name), name


def test_cpu_tensors_fail_loudly():
    """No CPU fallback anywhere on the product path."""
    net = MASIC.HSIC(16, 24, 3).eval()
    x = torch.rand(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x, x, torch.eye(3).unsqueeze(0))
    from compressai.layers import GDN
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GDN(3)(x)


# ------------------------------------------------------------------------------------------ C ABI
def test_c_abi_exports_every_declared_symbol():
    from masic_amd import _lib
    header = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "masic_hip.h")).read()
    declared = set(re.findall(r"\b(masic_[a-z0-9_]+)\s*\(", header))
    declared -= {"masic_conv_desc_t"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/masic_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib.masic_version() == 1
    assert ctypes.sizeof(_lib.ConvDesc) == 22 * 4


def test_c_abi_argument_errors_are_reported_not_thrown():
    from masic_amd import _lib
    d = _lib.ConvDesc(B=1, Cin=3, Hi=8, Wi=8, in_ctot=3, in_coff=0, Cout=4, Ho=5, Wo=4, out_ctot=4, out_coff=0,
                      KH=5, KW=5, stride=2, pad=2)
    assert _lib.lib.masic_conv_packed_bytes(ctypes.byref(d)) == 0          # wrong Ho
    assert b"output size" in _lib.lib.masic_last_error()
    rc = _lib.lib.masic_gdn_fwd(None, None, None, None, 1, 3, 4, 4, 0, 1e-6, None)
    assert rc == -1 and b"null pointer" in _lib.lib.masic_last_error()


def test_oracle_cqe_vs_golden():
    """Independent_EN (CQE): oracle against the reference's outputs (tests/golden/cqe_small.npz)."""
    fx = load_npz("cqe_small.npz")
    B, H, W = (int(v) for v in fx["BHW"])
    seed = int(fx["seed"])
    sd = synth.synth_state_dict(MASIC.Independent_EN().state_dict(), seed=seed)
    xa, xb, hm = synth.synth_inputs(B, H, W, seed=seed)
    with torch.no_grad():
        out = O.independent_en_forward(sd, xa, xb, hm)
    for k in ("x1_hat", "x2_hat"):
        assert float((out[k] - torch.from_numpy(fx[k])).abs().max()) <= 1e-6 * float(np.abs(fx[k]).max()), k


def test_lazy_psnr_is_a_float_when_used():
    """masic_amd.loss.LazyPSNR: mse2psnr (newtrain_codec_real.py:62-65) evaluated when first used -- formatting, arithmetic, comparison
    and meters see a float; the (device) scalar is read exactly once."""
    import math
    import torch
    from masic_amd.loss import LazyPSNR

    class Counting:
        def __init__(self, v):
            self.v, self.reads = v, 0

        def __float__(self):
            self.reads += 1
            return self.v
    src = Counting(0.004)
    p = LazyPSNR(src)
    want = 10 * math.log10(1 / 0.004)
    assert src.reads == 0
    assert abs(float(p) - want) < 1e-12 and f"{p:.3f}" == f"{want:.3f}" and repr(p) == repr(want)
    assert p - 1.0 == want - 1.0 and 1.0 - p == 1.0 - want and p + p == 2 * want and 2 * p == 2 * want and p / 2 == want / 2
    assert (p > 20) == (want > 20) and abs(p) == abs(want) and -p == -want and p == want
    assert src.reads == 1
    total = 0.0
    total += p * 3                      # AverageMeter.update(val, n): sum += val * n
    assert total == want * 3
    assert abs(float(LazyPSNR(torch.tensor(0.25, dtype=torch.float64))) - 10 * math.log10(4.0)) < 1e-12
