"""GPU parity tests (-m gpu) of the reference DRIVERS' loop shape on the product modules.

The drivers themselves (coremasic/mywork/newtrain_codec_real.py, newtrain_cqe_real.py) import cv2 / kornia at module scope
and cannot be imported; what an unchanged driver does to the model is restated here -- ONLY the caller side, in plain torch,
exactly as the driver writes it:
  * the criterion: `RateDistortionLoss(nn.Module)` built from `torch.log(likelihoods).sum()` and `nn.MSELoss()`
    (newtrain_codec_real.py:66-87; the CQE one, newtrain_cqe_real.py:69-96 kind=0, without its two ms_ssim report entries --
    pytorch_msssim is not in the image and they do not enter the loss);
  * the optimizers: plain `optim.Adam(net.parameters(), lr=1e-4)` / `optim.Adam(net.aux_parameters(), lr=1e-3)` (:434-435), not fused;
  * the step: `optimizer.zero_grad(); aux_optimizer.zero_grad(); out_net = model(d1, d2, h_matrix); out_criterion = criterion(out_net,
    d1, d2); out_criterion['loss'].backward(); optimizer.step(); aux_loss = model.aux_loss(); aux_loss.backward();
    aux_optimizer.step()` and the `.item()` reads of the log line (:135-161);
  * CQE: HSIC in eval mode WITH grad enabled, Independent_EN in train mode, Adam on model2 only, `model.aux_loss()` evaluated and not
    stepped (newtrain_cqe_real.py:128-174, :472).
Nothing of masic_amd.train / masic_amd.loss is used on the product side.  Checked against the gradient goldens the reference
produced (hsic_tiny.npz: 164 gradients at 1e-4; cqe_train.npz: 86) and against the same loop on the CPU oracle for two steps.
The only test plumbing is the injection of the recorded noise draws (goldens store the 7 tensors).

Also here: stale weight packs are impossible whoever steps the optimizer (fused Adam stepped by the caller, `.data` edits under
MASIC_PACK_VERIFY / invalidate_packs), and two models training in one process do not disturb each other's pack registry."""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.optim as optim

from oracle import hsic_oracle as O
from tests.util import assert_close, golden_state_dict, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ---------------------------------------------------------------- caller side, as the drivers write it
def mse2psnr(mse):
    return 10 * math.log10(1 / mse)


class RateDistortionLoss(nn.Module):
    """newtrain_codec_real.py:66-87."""

    def __init__(self, lmbda=1e-2):
        super().__init__()
        self.mse = nn.MSELoss()
        self.lmbda = lmbda

    def forward(self, output, target1, target2):
        N, _, H, W = target1.size()
        out = {}
        num_pixels = N * H * W
        out['bpp_loss'] = sum(
            (torch.log(likelihoods).sum() / (-math.log(2) * num_pixels))
            for likelihoods in output['likelihoods'].values())
        out['mse_loss'] = self.mse(output['x1_hat'], target1) + self.mse(output['x2_hat'], target2)
        out['loss'] = self.lmbda * 255 ** 2 * out['mse_loss'] + out['bpp_loss']
        out['psnr1'] = mse2psnr(self.mse(output['x1_hat'], target1))
        out['psnr2'] = mse2psnr(self.mse(output['x2_hat'], target2))
        return out


class DistortionLoss(nn.Module):
    """newtrain_cqe_real.py:69-96, kind=0 (minus the ms_ssim report entries)."""

    def __init__(self, lmbda=1e-2):
        super().__init__()
        self.mse = nn.MSELoss()
        self.lmbda = lmbda

    def forward(self, output, target1, target2, kind=0):
        out = {}
        out['mse_loss'] = self.mse(output['x1_hat'], target1) + self.mse(output['x2_hat'], target2)
        out['loss'] = self.lmbda * 255 ** 2 * out['mse_loss']
        out['psnr1'] = mse2psnr(self.mse(output['x1_hat'], target1))
        out['psnr2'] = mse2psnr(self.mse(output['x2_hat'], target2))
        return out


def driver_train_iteration(i, model, criterion, optimizer, aux_optimizer, d1, d2, h_matrix):
    """newtrain_codec_real.py:132-161 for one batch (h_matrix already computed and detached)."""
    h_matrix = h_matrix.detach()
    optimizer.zero_grad()
    aux_optimizer.zero_grad()

    out_net = model(d1, d2, h_matrix)

    out_criterion = criterion(out_net, d1, d2)
    out_criterion['loss'].backward()
    optimizer.step()

    aux_loss = model.aux_loss()
    aux_loss.backward()
    aux_optimizer.step()

    log = None
    if i % 10 == 0:
        log = (f'\tLoss: {out_criterion["loss"].item():.3f} |'
               f'\tMSE loss: {out_criterion["mse_loss"].item():.5f} |'
               f'\tBpp loss: {out_criterion["bpp_loss"].item():.2f} |'
               f'\tAux loss: {aux_loss.item():.2f}')
    return out_criterion, aux_loss, log


def driver_cqe_iteration(i, model, model2, criterion, optimizer, aux_optimizer, d1, d2, h_matrix):
    """newtrain_cqe_real.py:152-174."""
    h_matrix = h_matrix.detach()
    optimizer.zero_grad()
    aux_optimizer.zero_grad()

    out_net = model(d1, d2, h_matrix)

    out_net2 = model2(out_net['x1_hat'], out_net['x2_hat'], h_matrix)

    out_criterion = criterion(out_net2, d1, d2)
    out_criterion['loss'].backward()
    optimizer.step()

    aux_loss = model.aux_loss()
    return out_criterion, aux_loss, out_net2


class _Noise:
    """Feeds recorded noise tensors to the entropy models' draws, in order (test plumbing: the goldens store the draws)."""

    def __init__(self, tensors):
        from compressai.entropy_models import EntropyModel
        self.cls, self.queue = EntropyModel, list(tensors)

    def __enter__(self):
        self.orig = self.cls._get_noise_cached
        q = self.queue
        self.cls._get_noise_cached = lambda self_, x: q.pop(0).reshape(x.shape).contiguous()
        return self

    def __exit__(self, *exc):
        self.cls._get_noise_cached = self.orig
        return False


def _tiny():
    import MASIC
    fx = load_npz("hsic_tiny.npz")
    N, M, K = (int(v) for v in fx["NMK"])
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(golden_state_dict(fx, net.state_dict()))
    return fx, net.to(DEV), (N, M, K)


# ---------------------------------------------------------------- codec stage
def test_driver_loop_gradients_vs_reference_golden():
    """One iteration of the driver's loop on the product HSIC (train mode, plain-torch criterion, plain Adam): loss and all 164
    parameter gradients against the reference's (hsic_tiny.npz) at 1e-4, the aux loss and its two quantile gradients through
    `model.aux_loss().backward()`, the `.item()` reads of the log line, and that both optimizers moved what they own."""
    fx, net, _ = _tiny()
    net.train()
    d1, d2, hm = (torch.from_numpy(fx[k]).to(DEV) for k in ("x1", "x2", "h_matrix"))
    criterion = RateDistortionLoss(lmbda=float(fx["lmbda"]))
    optimizer = optim.Adam(net.parameters(), lr=1e-4)
    aux_optimizer = optim.Adam(net.aux_parameters(), lr=1e-3)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    grads, aux_grads, phase = {}, {}, ["main"]

    def grab(p, n):
        (grads if phase[0] == "main" else aux_grads)[n] = p.grad.detach().clone()
    hooks = [p.register_post_accumulate_grad_hook(lambda p, n=n: grab(p, n)) for n, p in net.named_parameters()]
    orig_step = optimizer.step

    def step_then_switch(*a, **k):          # gradients that arrive after optimizer.step() belong to the aux backward
        r = orig_step(*a, **k)
        phase[0] = "aux"
        return r
    optimizer.step = step_then_switch
    with _Noise([torch.from_numpy(fx["train/noise_" + k]).to(DEV) for k in O.NOISE_KEYS]) as nz:
        out_criterion, aux_loss, log = driver_train_iteration(0, net, criterion, optimizer, aux_optimizer, d1, d2, hm)
        assert not nz.queue
    for h in hooks:
        h.remove()
    assert log is not None and "Loss:" in log
    want = float(fx["train/loss_loss"])
    assert abs(out_criterion["loss"].item() - want) <= 1e-4 * abs(want), (out_criterion["loss"].item(), want)
    assert abs(aux_loss.item() - float(fx["train/aux_loss"])) <= 1e-4 * float(fx["train/aux_loss"])
    assert isinstance(out_criterion["psnr1"], float) and isinstance(out_criterion["psnr2"], float)
    worst, worst_name, n = 0.0, "", 0
    for key in fx:
        if not key.startswith("train/grad/"):
            continue
        name = key[len("train/grad/"):]
        ref = torch.from_numpy(fx[key])
        assert name in grads, name
        e = float((grads[name].cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        n += 1
        if e > worst:
            worst, worst_name = e, name
    print(f"driver loop (codec stage): {n} parameter gradients vs the reference, worst relative error {worst:.2e} ({worst_name})")
    assert n == 164 and worst <= 1e-4, (n, worst, worst_name)
    assert "entropy_bottleneck1.quantiles" not in grads           # no gradient from the main loss, as in the reference
    # the aux backward reaches the quantiles only; the main-loss gradients of the other bottleneck parameters are still there and the
    # aux Adam steps them (MASIC.py:85-94 -- the reference's parameter split)
    for nm in ("entropy_bottleneck1.quantiles", "entropy_bottleneck2.quantiles"):
        assert_close(aux_grads[nm], torch.from_numpy(fx["train/auxgrad/" + nm]), "aux:" + nm, 1e-4)
    assert set(aux_grads) == {"entropy_bottleneck1.quantiles", "entropy_bottleneck2.quantiles"}
    moved = {n: float((p.detach() - before[n]).abs().max()) for n, p in net.named_parameters()}
    main_names = {n for n, _ in net.named_parameters() if not n.startswith("entropy_bottleneck")}
    assert all(0 < moved[n] <= 1.01e-4 for n in main_names), [(n, moved[n]) for n in main_names if not 0 < moved[n] <= 1.01e-4][:3]
    assert all(0 < v <= 1.01e-3 for n, v in moved.items() if n not in main_names)


def test_driver_loop_two_iterations_vs_oracle_loop():
    """Two iterations of the driver's loop (plain Adam 1e-4 / 1e-3, `model.aux_loss().backward()`) on the product HSIC(16,24,3)
    against the same loop on the CPU oracle with torch autograd: losses per iteration at 2e-4, parameter updates per element."""
    import MASIC
    from masic_amd import synth
    N, M, K = 16, 24, 3
    lmbda = 0.01
    sd0 = synth.synth_state_dict(MASIC.HSIC(N, M, K).state_dict(), seed=33)
    x1, x2, hm = synth.synth_inputs(2, 64, 64, seed=33)
    noises = [synth.synth_noise(2, N, M, 64, 64, seed=45 + i) for i in range(2)]
    names = [n for n, _ in MASIC.HSIC(N, M, K).named_parameters()]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd0.items()}
    opt = optim.Adam([sd[n] for n in names if not n.startswith("entropy_bottleneck")], lr=1e-4)
    aopt = optim.Adam([sd[n] for n in names if n.startswith("entropy_bottleneck")], lr=1e-3)
    ref = []
    for it in range(2):
        opt.zero_grad(); aopt.zero_grad()
        with torch.no_grad():      # MaskedConv2d.forward zeroes the stored masked taps in place (layers.py:77)
            for cp in ("context_prediction1.weight", "context_prediction2.weight"):
                sd[cp].copy_(O.masked_weight(sd[cp].detach()))
        out = O.hsic_forward(sd, x1, x2, hm, K=K, training=True, noise=noises[it])
        loss = O.rd_loss(out, x1, x2, lmbda)["loss"]
        loss.backward()
        opt.step()
        a = O.eb_aux_loss(sd, "entropy_bottleneck1") + O.eb_aux_loss(sd, "entropy_bottleneck2")
        a.backward()
        aopt.step()
        ref.append((float(loss), float(a)))
    net = MASIC.HSIC(N, M, K)
    net.load_state_dict(sd0)
    net = net.to(DEV).train()
    criterion = RateDistortionLoss(lmbda=lmbda)
    optimizer = optim.Adam(net.parameters(), lr=1e-4)
    aux_optimizer = optim.Adam(net.aux_parameters(), lr=1e-3)
    d1, d2, h = x1.to(DEV), x2.to(DEV), hm.to(DEV)
    for it in range(2):
        with _Noise([noises[it][k].to(DEV) for k in O.NOISE_KEYS]):
            crit, aux, _ = driver_train_iteration(it, net, criterion, optimizer, aux_optimizer, d1, d2, h)
        assert abs(crit["loss"].item() - ref[it][0]) <= 2e-4 * abs(ref[it][0]), (it, crit["loss"].item(), ref[it])
        assert abs(aux.item() - ref[it][1]) <= 2e-4 * abs(ref[it][1])
    total_bad = total = 0
    worst_frac, worst_name = 0.0, ""
    for n, p in net.named_parameters():
        lr = 1e-3 if n.startswith("entropy_bottleneck") else 1e-4
        du_ref = (sd[n].detach() - sd0[n]).double()
        du = (p.detach().cpu() - sd0[n]).double()
        bad = (du - du_ref).abs() > 0.05 * lr
        total_bad += int(bad.sum()); total += bad.numel()
        frac = float(bad.double().mean())
        if frac > worst_frac:
            worst_frac, worst_name = frac, n
        assert float((du - du_ref).norm()) <= 0.5 * float(du_ref.norm()) + 1e-30, n
    print(f"driver loop, two iterations vs the oracle loop: {total_bad}/{total} elements off by > 5% of a step; worst tensor {worst_name} ({worst_frac:.2%})")
    assert total_bad <= 0.002 * total and worst_frac <= 0.05


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_driver_loop_reduces_the_loss_full_width(prec):
    """HSIC(128,192,5) from the reference's default init (MASIC.py:55-60), 2 x 128 x 128 band-limited pairs, 12 iterations of the
    driver's loop in both operand modes: the loss falls and every parameter the main optimizer owns has a finite gradient."""
    import MASIC
    from masic_amd import nn as mnn, synth
    torch.manual_seed(0)
    net = MASIC.HSIC(128, 192, 5).to(DEV).train()
    criterion = RateDistortionLoss(lmbda=0.01)
    optimizer = optim.Adam(net.parameters(), lr=1e-4)
    aux_optimizer = optim.Adam(net.aux_parameters(), lr=1e-3)
    d1, d2, hm = (t.to(DEV) for t in synth.synth_inputs(2, 128, 128, seed=5))
    mnn.set_precision(prec)
    try:
        losses, auxes = [], []
        for i in range(12):
            crit, aux, _ = driver_train_iteration(i, net, criterion, optimizer, aux_optimizer, d1, d2, hm)
            losses.append(crit["loss"].item())
            auxes.append(aux.item())
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    finally:
        mnn.set_precision("f32")
    print(f"driver loop [{prec}] 12 iterations from default init: loss {losses[0]:.3f} -> {losses[-1]:.3f}, aux {auxes[0]:.1f} -> {auxes[-1]:.1f}")
    assert losses[-1] < 0.9 * losses[0] and auxes[-1] < auxes[0], (losses, auxes)


# ---------------------------------------------------------------- CQE stage
def test_cqe_driver_loop_vs_reference_golden():
    """newtrain_cqe_real.py:128-174 as the driver writes it -- HSIC in eval mode with grad ENABLED (no no_grad around it),
    Independent_EN in train mode, plain criterion, plain Adam on model2 -- on hsic_tiny's weights: Independent_EN's 86 gradients
    against the reference's, the outputs, the loss; `model.aux_loss()` is evaluated and nothing of HSIC is stepped."""
    import MASIC
    from masic_amd import synth
    from tests.test_gpu_cqe import _check_grad
    fx = load_npz("cqe_train.npz")
    tiny, hsic, _ = _tiny()
    hsic.eval()
    net2 = MASIC.Independent_EN()
    net2.load_state_dict(synth.synth_state_dict(net2.state_dict(), seed=int(fx["seed_en"])))
    net2 = net2.to(DEV).train()
    d1, d2, hm = (torch.from_numpy(tiny[k]).to(DEV) for k in ("x1", "x2", "h_matrix"))
    criterion = DistortionLoss(lmbda=float(fx["lmbda"]))
    optimizer = optim.Adam(net2.parameters(), lr=1e-4)
    aux_optimizer = optim.Adam(hsic.aux_parameters(), lr=1e-3)
    grads = {}
    hooks = [p.register_post_accumulate_grad_hook(lambda p, n=n: grads.__setitem__(n, p.grad.detach().clone()))
             for n, p in net2.named_parameters()]
    hsic_before = {n: p.detach().clone() for n, p in hsic.named_parameters()}
    crit, aux_loss, out2 = driver_cqe_iteration(0, hsic, net2, criterion, optimizer, aux_optimizer, d1, d2, hm)
    for h in hooks:
        h.remove()
    for k in ("x1_hat", "x2_hat"):
        assert_close(out2[k], torch.from_numpy(fx["chain/" + k]), "cqe driver loop:" + k)
    assert abs(crit["loss"].item() - float(fx["chain/loss"])) <= 1e-4 * abs(float(fx["chain/loss"]))
    worst = [0.0, ""]
    for name, _ in net2.named_parameters():
        _check_grad(fx, "chain/grad/" + name, grads.get(name), worst, "chain/f32_floor/" + name)
    print(f"driver loop (CQE stage): 86 gradients vs the reference, worst error / tolerance {worst[0]:.2f} ({worst[1]})")
    assert len(grads) == 86 and worst[0] <= 1.0, worst
    assert math.isfinite(aux_loss.item())
    # what the reference's graph leaves behind in HSIC: gradients on the two synthesis transforms (round() blocks everything upstream),
    # never stepped -- the product records the same (HSIC.forward, eval mode with grad enabled)
    hs_worst, nd = [0.0, ""], 0
    for n, p in hsic.named_parameters():
        assert torch.equal(p.detach(), hsic_before[n]), n
        if n.startswith(("decoder1.", "decoder2.")):
            _check_grad(fx, "chain/hsic_grad/" + n, p.grad, hs_worst, "chain/hsic_f32_floor/" + n)
            nd += 1
        else:
            assert p.grad is None, n
    print(f"  gradients reaching HSIC's synthesis transforms: {nd}, worst error / tolerance {hs_worst[0]:.2f} ({hs_worst[1]})")
    assert nd == 32 and hs_worst[0] <= 1.0, hs_worst
    # a frozen codec (requires_grad_(False), the usual way to say so) records nothing and gives Independent_EN the same gradients
    for p_ in hsic.parameters():
        p_.requires_grad_(False)
    net2.zero_grad()
    grads2 = {}
    hooks = [p.register_post_accumulate_grad_hook(lambda p, n=n: grads2.__setitem__(n, p.grad.detach().clone())) for n, p in net2.named_parameters()]
    net3_out = net2(*(hsic(d1, d2, hm)[k] for k in ("x1_hat", "x2_hat")), hm)
    assert hsic(d1, d2, hm)["x1_hat"].grad_fn is None
    criterion(net3_out, d1, d2)["loss"].backward()
    for h in hooks:
        h.remove()
    assert len(grads2) == 86


# ---------------------------------------------------------------- stale packs are impossible
def _fresh_like(net, cls_args):
    import MASIC
    ref = MASIC.HSIC(*cls_args)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    return ref.to(DEV).eval()


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_fused_adam_stepped_by_the_caller_never_serves_stale_packs(prec):
    """A caller builds `Adam(..., fused=True)` and steps it directly (no masic_amd.train helper).  torch's fused step does not bump the
    parameters' version counters, which every weight-pack / table cache and GraphedHSIC's freshness check are keyed by; the global
    optimizer-step hook of masic_amd/fresh.py does.  After each step the eval forward AND a GraphedHSIC replay must equal a FRESH
    model built from state_dict() alone."""
    import MASIC
    from masic_amd import nn as mnn, synth
    from masic_amd.graph import GraphedHSIC
    args = (128, 192, 5) if prec == "bf16" else (32, 48, 3)
    net = MASIC.HSIC(*args)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=12))
    net = net.to(DEV)
    d1, d2, hm = (t.to(DEV) for t in synth.synth_inputs(2, 64, 128, seed=12))
    criterion = RateDistortionLoss(lmbda=0.01)
    optimizer = optim.Adam(net.parameters(), lr=1e-3, fused=True)
    aux_optimizer = optim.Adam(net.aux_parameters(), lr=1e-2, fused=True)
    mnn.set_precision(prec)
    try:
        net.eval()
        with torch.no_grad():
            net(d1, d2, hm)                                  # packs of the initial weights are cached now
        graphed = GraphedHSIC(net, d1, d2, hm)
        x_first = graphed(d1, d2, hm)["x1_hat"].clone()
        for it in range(2):
            net.train()
            v0 = net.encoder1.g_a_conv2.weight._version
            driver_train_iteration(1, net, criterion, optimizer, aux_optimizer, d1, d2, hm)
            assert net.encoder1.g_a_conv2.weight._version > v0, "the step hook must bump the version of fused-stepped parameters"
            net.eval()
            fresh = _fresh_like(net, args)
            with torch.no_grad():
                want = fresh(d1, d2, hm)
                got = net(d1, d2, hm)
            rep = graphed(d1, d2, hm)
            for k in ("x1_hat", "x2_hat", "y1_hat"):
                assert torch.equal(got[k], want[k]), (it, k, "eager forward ran on stale packs")
                assert torch.equal(rep[k], want[k]), (it, k, "graph replay ran on stale packs")
            for k in want["likelihoods"]:
                assert torch.equal(got["likelihoods"][k], want["likelihoods"][k]), (it, k)
                assert torch.equal(rep["likelihoods"][k], want["likelihoods"][k]), (it, k)
        assert not torch.equal(x_first, want["x1_hat"])      # the weights did move
    finally:
        mnn.set_precision("f32")


def test_data_writes_are_caught_by_verify_mode_and_by_invalidate():
    """`p.data.mul_()` is invisible to every version counter (torch's design).  MASIC_PACK_VERIFY / set_pack_verify adds a content
    fingerprint to every cache key: the next forward equals a fresh model's.  Without it, `invalidate_packs(model)` is the documented
    call, and the test also shows what it protects from: the unguarded forward still serves the old pack."""
    import MASIC
    from masic_amd import nn as mnn, synth
    args = (32, 48, 3)
    net = MASIC.HSIC(*args)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=14))
    net = net.to(DEV).eval()
    d1, d2, hm = (t.to(DEV) for t in synth.synth_inputs(1, 64, 64, seed=14))
    with torch.no_grad():
        base = net(d1, d2, hm)["x1_hat"].clone()
        net.decoder1.g_s_conv2.weight.data.mul_(1.5)
        net.entropy_bottleneck1._biases[0].data.add_(0.25)
        stale = net(d1, d2, hm)
        assert torch.equal(stale["x1_hat"], base)            # what a .data edit does without a guard
        mnn.invalidate_packs(net)
        want = _fresh_like(net, args)(d1, d2, hm)
        got = net(d1, d2, hm)
        assert torch.equal(got["x1_hat"], want["x1_hat"]) and torch.equal(got["likelihoods"]["z1"], want["likelihoods"]["z1"])
        assert not torch.equal(got["x1_hat"], base)
        mnn.set_pack_verify(True)
        try:
            net(d1, d2, hm)
            net.decoder1.g_s_conv3.weight.data.mul_(0.5)
            net.context_prediction1.weight.data.mul_(2.0)
            want = _fresh_like(net, args)(d1, d2, hm)
            got = net(d1, d2, hm)
            for k in ("x1_hat", "x2_hat"):
                assert torch.equal(got[k], want[k]), k
            assert torch.equal(got["likelihoods"]["y1"], want["likelihoods"]["y1"])
        finally:
            mnn.set_pack_verify(False)


def test_two_models_training_in_one_process_share_the_pack_registry():
    """ops.StreamPacks is a per-device registry of persistent weight packs whose batched refresh is decided by a stale-majority vote
    over the entries in use.  Two models stepped alternately (one through masic_amd.train.train_step = batched refresh, one through
    the driver's loop) must each see exactly their own current weights: after every step the eval forward of each equals a fresh copy."""
    import MASIC
    from masic_amd import nn as mnn, synth, train
    args = (128, 192, 5)
    nets = []
    for seed in (3, 4):
        n = MASIC.HSIC(*args)
        n.load_state_dict(synth.synth_state_dict(n.state_dict(), seed=seed))
        nets.append(n.to(DEV))
    batches = [tuple(t.to(DEV) for t in synth.synth_inputs(1, 64, 64, seed=s)) for s in (3, 4)]
    mnn.set_precision("bf16")
    try:
        oa, aa = train.make_optimizers(nets[0], lr=1e-3, aux_lr=1e-2)
        ob, ab = optim.Adam(nets[1].parameters(), lr=1e-3), optim.Adam(nets[1].aux_parameters(), lr=1e-2)
        crit = RateDistortionLoss(0.01)
        for it in range(3):
            nets[0].train(); nets[1].train()
            train.train_step(nets[0], oa, aa, *batches[0], 0.01)
            driver_train_iteration(1, nets[1], crit, ob, ab, *batches[1])
            if it == 1:
                train.train_step(nets[0], oa, aa, *batches[0], 0.01)       # uneven cadence: A steps twice, B once
            for n, b in zip(nets, batches):
                n.eval()
                with torch.no_grad():
                    want = _fresh_like(n, args)(*b)
                    got = n(*b)
                for k in ("x1_hat", "x2_hat"):
                    assert torch.equal(got[k], want[k]), (it, k)
                for k in want["likelihoods"]:
                    assert torch.equal(got["likelihoods"][k], want["likelihoods"][k]), (it, k)
    finally:
        mnn.set_precision("f32")


# ---------------------------------------------------------------- evaluation driver (test2_real.py)
def test_eval_driver_loop_vs_reference_golden():
    """test2_real.py:81-114 (criterion; its ms_ssim / lpips report entries need pytorch_msssim / lpips, which are not in the image and do not
    enter the loss) and :172-231 (test_epoch: eval mode, no_grad, AverageMeter updates with device tensors, model.aux_loss()) restated on the
    caller side, on the product HSIC with the reference's weights and inputs of hsic_tiny.npz: every scalar the driver prints -- loss, mse,
    bpp, bpp1 / bpp2, psnr1 / psnr2 -- against the reference's own (1e-4), after two passes over the same pair through the meters."""
    class EvalLoss(nn.Module):
        def __init__(self, lmbda=1e-2):
            super().__init__()
            self.mse = nn.MSELoss()
            self.lmbda = lmbda

        @staticmethod
        def mse2psnr(mse):                          # test2_real.py:66-69
            return 10 * math.log10(1 / mse)

        def forward(self, output, target1, target2):
            N, _, H, W = target1.size()
            out = {}
            num_pixels = N * H * W
            out['bpp_loss'] = sum((torch.log(likelihoods).sum() / (-math.log(2) * num_pixels)) for likelihoods in output['likelihoods'].values())
            out['mse_loss'] = self.mse(output['x1_hat'], target1) + self.mse(output['x2_hat'], target2)
            out['bpp1'] = (torch.log(output['likelihoods']['y1']).sum() / (-math.log(2) * num_pixels)) + (
                torch.log(output['likelihoods']['z1']).sum() / (-math.log(2) * num_pixels))
            out['bpp2'] = (torch.log(output['likelihoods']['y2']).sum() / (-math.log(2) * num_pixels)) + (
                torch.log(output['likelihoods']['z2']).sum() / (-math.log(2) * num_pixels))
            out['loss'] = self.lmbda * 255 ** 2 * out['mse_loss'] + out['bpp_loss']
            out['psnr1'] = self.mse2psnr(self.mse(output['x1_hat'], target1))
            out['psnr2'] = self.mse2psnr(self.mse(output['x2_hat'], target2))
            return out

    class AverageMeter:
        def __init__(self):
            self.val = 0; self.avg = 0; self.sum = 0; self.count = 0

        def update(self, val, n=1):
            self.val = val
            self.sum += val * n
            self.count += n
            self.avg = self.sum / self.count

    fx, net, _ = _tiny()
    d1, d2, hm = (torch.from_numpy(fx[k]).to(DEV) for k in ("x1", "x2", "h_matrix"))
    criterion = EvalLoss(lmbda=float(fx["lmbda"]))
    net.eval()
    meters = {k: AverageMeter() for k in ("loss", "bpp_loss", "mse_loss", "aux_loss", "psnr1", "psnr2", "bpp1", "bpp2")}
    with torch.no_grad():
        for _ in range(2):
            out_net = net(d1, d2, hm)
            out_criterion = criterion(out_net, d1, d2)
            meters["aux_loss"].update(net.aux_loss())
            for k in ("loss", "bpp_loss", "mse_loss", "psnr1", "psnr2", "bpp1", "bpp2"):
                meters[k].update(out_criterion[k])
    line = (f'\tLoss: {meters["loss"].avg:.3f} |\tMSE loss: {meters["mse_loss"].avg:.4f} |\tPSNR (dB): {(meters["psnr1"].avg + meters["psnr2"].avg) / 2:.3f} |'
            f'\tBpp loss: {meters["bpp_loss"].avg / 2:.4f} |\tBPP1: {meters["bpp1"].avg:.3f} |\tAux loss: {meters["aux_loss"].avg:.2f}')
    assert "Loss:" in line
    want = {"loss": "eval/loss_loss", "bpp_loss": "eval/loss_bpp_loss", "mse_loss": "eval/loss_mse_loss", "psnr1": "eval/loss_psnr1", "psnr2": "eval/loss_psnr2"}
    for k, g in want.items():
        a, b = float(meters[k].avg), float(fx[g])
        assert abs(a - b) <= 1e-4 * abs(b), (k, a, b)
    for k, (gy, gz) in {"bpp1": ("eval/loss_bpp_y1", "eval/loss_bpp_z1"), "bpp2": ("eval/loss_bpp_y2", "eval/loss_bpp_z2")}.items():
        a, b = float(meters[k].avg), float(fx[gy]) + float(fx[gz])
        assert abs(a - b) <= 1e-4 * abs(b), (k, a, b)
    for k in ("x1_hat", "x2_hat", "y1_hat", "z1_hat", "x1_mask_R", "x1_mask_L"):
        assert_close(out_net[k], torch.from_numpy(fx["eval/" + k]), "eval driver:" + k)


def test_cqe_eval_driver_loop_vs_reference_golden():
    """test3_real.py:186-194 (the CQE evaluation driver: `out_net = model(d1, d2, h)`, `out_net2 = model2(out_net['x1_hat'], out_net['x2_hat'], h)`,
    the distortion criterion on out_net2 and the rate criterion (`kind=1`) on out_net, under no_grad with both networks in eval mode) on the
    product modules with the reference's weights: Independent_EN's outputs and loss against cqe_train.npz's chain entries (Independent_EN has
    no mode-dependent layer), the rate against hsic_tiny.npz's eval goldens."""
    import MASIC
    from masic_amd import synth
    fx = load_npz("cqe_train.npz")
    tiny, hsic, _ = _tiny()
    hsic.eval()
    net2 = MASIC.Independent_EN()
    net2.load_state_dict(synth.synth_state_dict(net2.state_dict(), seed=int(fx["seed_en"])))
    net2 = net2.to(DEV).eval()
    d1, d2, hm = (torch.from_numpy(tiny[k]).to(DEV) for k in ("x1", "x2", "h_matrix"))
    criterion = DistortionLoss(lmbda=float(fx["lmbda"]))
    with torch.no_grad():
        out_net = hsic(d1, d2, hm)
        out_net2 = net2(out_net['x1_hat'], out_net['x2_hat'], hm)
        out_criterion = criterion(out_net2, d1, d2)
        N, _, H, W = d1.size()
        bpp = sum((torch.log(l).sum() / (-math.log(2) * N * H * W)) for l in out_net['likelihoods'].values())     # criterion(out_net, d1, d2, kind=1)
        aux = hsic.aux_loss()
    for k in ("x1_hat", "x2_hat"):
        assert_close(out_net2[k], torch.from_numpy(fx["chain/" + k]), "cqe eval driver:" + k)
    assert abs(out_criterion["loss"].item() - float(fx["chain/loss"])) <= 1e-4 * abs(float(fx["chain/loss"]))
    assert abs(bpp.item() - float(tiny["eval/loss_bpp_loss"])) <= 1e-4 * float(tiny["eval/loss_bpp_loss"])
    assert math.isfinite(aux.item()) and math.isfinite(out_criterion["psnr1"])
