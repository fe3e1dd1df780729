"""Generates tests/golden/cqe_train.npz from the REFERENCE implementation (run in the build container only).

    python tests/golden/make_cqe_goldens.py

The CQE training step of the reference (coremasic/mywork/newtrain_cqe_real.py:128-174): HSIC in eval mode, Independent_EN
in train mode, loss = lmbda * 255^2 * (MSE(x1_hat2, d1) + MSE(x2_hat2, d2)) on the outputs of Independent_EN, backward.
The reference's own MASIC.py / compressai are imported on CPU through oracle/ref_import.py (kornia warp restated there);
the criterion of the driver file is restated (that file imports cv2 / pytorch_msssim at module scope).  Data only.

Cases
(Gradients of tensors with more than 4096 entries are stored as 512 sampled entries + L2 norm + absmax.)
  standalone/   Independent_EN() on given decoded views: 2x3x32x48, inputs WITH requires_grad -- every parameter gradient
                (86 tensors), the gradients w.r.t. both input pictures, outputs and loss.
  chain/        the reference's full graph: HSIC(16,24,3) of hsic_tiny.npz in eval mode (grad enabled, as the reference runs
                it) -> Independent_EN -> loss -> backward on 1x3x64x64: Independent_EN's 86 gradients, outputs, loss, and the
                gradients that reach HSIC's two synthesis transforms (the analysis side gets exact zeros through round()).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import hsic_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402

R = ref_import.load()
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("masic_synth", os.path.join(ROOT, "masic_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

torch.set_num_threads(8)
LMBDA = 0.01
SEED_EN = 6


def cqe_loss(out, d1, d2):
    """newtrain_cqe_real.py:78-84, kind=0 (the MS-SSIM / PSNR entries are reporting only)."""
    mse = torch.nn.functional.mse_loss(out["x1_hat"], d1) + torch.nn.functional.mse_loss(out["x2_hat"], d2)
    return LMBDA * 255 ** 2 * mse


def put_grad(fx, key, g, rs):
    """Small tensors in full; large ones as L2 norm + 512 sampled entries (a fixture is data, and small)."""
    g = g.detach().numpy()
    if g.size <= 4096:
        fx[key] = g.copy()
        return
    idx = rs.randint(0, g.size, size=512)
    fx[key + "@idx"] = idx.astype(np.int64)
    fx[key + "@val"] = g.reshape(-1)[idx].copy()
    fx[key + "@norm"] = np.array(float(np.sqrt((g.astype(np.float64) ** 2).sum())))
    fx[key + "@absmax"] = np.array(float(np.abs(g).max()))


def build_en():
    en = R.Independent_EN()
    sd = synth.synth_state_dict(en.state_dict(), seed=SEED_EN)
    en.load_state_dict(sd)
    return en.train(), sd


def main():
    fx = {"lmbda": np.array(LMBDA), "seed_en": np.array(SEED_EN)}
    report = {}
    # ---------------------------------------------------------------- standalone
    en, sd = build_en()
    d1, d2, H = synth.synth_inputs(2, 32, 48, seed=7)
    rs = np.random.RandomState(77)
    xa = (d1 + torch.from_numpy((0.03 * rs.standard_normal(size=tuple(d1.shape))).astype(np.float32))).requires_grad_(True)
    xb = (d2 + torch.from_numpy((0.03 * rs.standard_normal(size=tuple(d2.shape))).astype(np.float32))).requires_grad_(True)
    out = en(xa, xb, H)
    loss = cqe_loss(out, d1, d2)
    loss.backward()
    fx.update({"standalone/d1": d1.numpy(), "standalone/d2": d2.numpy(), "standalone/h_matrix": H.numpy(),
               "standalone/x1_in": xa.detach().numpy(), "standalone/x2_in": xb.detach().numpy(),
               "standalone/x1_hat": out["x1_hat"].detach().numpy(), "standalone/x2_hat": out["x2_hat"].detach().numpy(),
               "standalone/loss": np.array(float(loss)),
               "standalone/gin/x1": xa.grad.numpy().copy(), "standalone/gin/x2": xb.grad.numpy().copy()})
    n = 0
    for name, p in en.named_parameters():
        assert p.grad is not None, name
        put_grad(fx, "standalone/grad/" + name, p.grad, rs)
        n += 1
    assert n == 86, n
    report["standalone_params_with_grad"] = n
    # oracle (autograd over the restatement) agrees with the reference
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xa2, xb2 = xa.detach().clone().requires_grad_(True), xb.detach().clone().requires_grad_(True)
    oo = O.independent_en_forward(sd_g, xa2, xb2, H)
    cqe_loss(oo, d1, d2).backward()
    worst = 0.0
    for name, p in en.named_parameters():
        g, og = p.grad, sd_g[name].grad
        worst = max(worst, float((g - og).abs().max()) / (float(g.abs().max()) + 1e-30))
    report["standalone_oracle_vs_reference_grad_maxrel"] = worst
    # float32 noise floor of each gradient: the same restatement evaluated in float64 (two correct float32 evaluations can
    # differ by this much; a LeakyReLU argument near 0 or a cancelling sum amplifies rounding) -- tests gate on
    # max(1e-4, 2 x floor)
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    xa64, xb64 = xa.detach().double().requires_grad_(True), xb.detach().double().requires_grad_(True)
    o64 = O.independent_en_forward(sd64, xa64, xb64, H.double())
    cqe_loss(o64, d1.double(), d2.double()).backward()
    for key, g32, g64 in (("x1", xa.grad, xa64.grad), ("x2", xb.grad, xb64.grad)):
        fx["standalone/f32_floor/gin/" + key] = np.array(float((g32.double() - g64).abs().max() / g64.abs().max()))
        report["standalone_f32_vs_f64_gin_" + key] = float(fx["standalone/f32_floor/gin/" + key])
    floors = {}
    for name, p in en.named_parameters():
        floors[name] = float((p.grad.double() - sd64[name].grad).abs().max() / sd64[name].grad.abs().max())
        fx["standalone/f32_floor/" + name] = np.array(floors[name])
    report["standalone_f32_vs_f64_grad_maxrel"] = max(floors.values())
    report["standalone_f32_vs_f64_grad_worst"] = max(floors, key=floors.get)
    report["standalone_oracle_vs_reference_gin_maxabs"] = float(max((xa.grad - xa2.grad).abs().max(), (xb.grad - xb2.grad).abs().max()))

    # ---------------------------------------------------------------- chain: the reference's full graph
    tiny = dict(np.load(os.path.join(HERE, "hsic_tiny.npz"), allow_pickle=False))
    N, M, K = (int(v) for v in tiny["NMK"])
    net = R.HSIC(N=N, M=M, K=K)
    hsd = synth.synth_state_dict(net.state_dict(), seed=1)
    for k in tiny:
        if k.startswith("sd/"):
            assert np.array_equal(hsd[k[3:]].numpy(), tiny[k]), k        # hsic_tiny.npz's weights ARE seed 1
    net.load_state_dict(hsd)
    net.eval()
    en, sd = build_en()
    x1, x2, Hm = (torch.from_numpy(tiny[k]) for k in ("x1", "x2", "h_matrix"))
    net.zero_grad()
    out1 = net(x1, x2, Hm)
    out2 = en(out1["x1_hat"], out1["x2_hat"], Hm)
    loss = cqe_loss(out2, x1, x2)
    loss.backward()
    fx.update({"chain/x1_hat": out2["x1_hat"].detach().numpy(), "chain/x2_hat": out2["x2_hat"].detach().numpy(),
               "chain/loss": np.array(float(loss))})
    for name, p in en.named_parameters():
        put_grad(fx, "chain/grad/" + name, p.grad, rs)
    zero_side, dec = 0.0, 0
    for name, p in net.named_parameters():
        if name.startswith(("decoder1.", "decoder2.")):
            assert p.grad is not None, name
            put_grad(fx, "chain/hsic_grad/" + name, p.grad, rs)
            dec += 1
        elif p.grad is not None:
            zero_side = max(zero_side, float(p.grad.abs().max()))
    report["chain_decoder_params_with_grad"] = dec
    # float32 floors for this case: decoder1 -> warp -> decoder2 -> Independent_EN -> loss restated in float64 on the
    # reference's own quantised latents (the part of the graph that carries gradient)
    with torch.no_grad():
        y2 = net.encoder2(O.warp_perspective(x1, Hm, x1.shape[-2:]), x2)
    h64 = {k: (v.double().clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in hsd.items()}
    e64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    x1h = O.decoder1(out1["y1_hat"].detach().double(), h64)
    x2h = O.decoder2(torch.round(y2).double(), O.warp_perspective(x1h, Hm.double(), x1.shape[-2:]), h64)
    cqe_loss(O.independent_en_forward(e64, x1h, x2h, Hm.double()), x1.double(), x2.double()).backward()
    fl = {}
    for name, p in en.named_parameters():
        fl[name] = float((p.grad.double() - e64[name].grad).abs().max() / e64[name].grad.abs().max())
        fx["chain/f32_floor/" + name] = np.array(fl[name])
    for name, p in net.named_parameters():
        if name.startswith(("decoder1.", "decoder2.")):
            fl[name] = float((p.grad.double() - h64[name].grad).abs().max() / h64[name].grad.abs().max())
            fx["chain/hsic_f32_floor/" + name] = np.array(fl[name])
    report["chain_f32_vs_f64_grad_maxrel"] = max(fl.values())
    report["chain_f32_vs_f64_grad_worst"] = max(fl, key=fl.get)
    report["chain_max_abs_grad_outside_the_decoders"] = zero_side      # round() has zero gradient: analysis / entropy side gets 0
    np.savez_compressed(os.path.join(HERE, "cqe_train.npz"), **fx)
    import json
    json.dump(report, open(os.path.join(HERE, "cqe_pin_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
