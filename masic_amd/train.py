"""One optimisation step of the codec stage, as the reference's train_epoch does it
(coremasic/mywork/newtrain_codec_real.py:135-146): zero_grad x2 -> forward -> RD loss -> backward -> Adam step ->
aux loss -> backward -> aux Adam step.  Optimizers are torch's (plumbing); every tensor op of forward and backward
is a HIP launch.  With a GradientAllReducer the main-loss gradients are averaged across ranks before the step."""
import os

import torch

from .loss import distortion, rate_distortion


def make_optimizers(model, lr=1e-4, aux_lr=1e-3, fused=None):
    """newtrain_codec_real.py:434-435 (note: aux optimizer owns ALL entropy-bottleneck parameters, MASIC.py:85-94).
    fused (default: on for device parameters): torch's single-kernel Adam -- the same update; the default `foreach` form is ~25 small
    launches per optimizer per step, and a launch costs the device ~10 us here whatever it does."""
    params, aux = list(model.parameters()), list(model.aux_parameters())
    if fused is None:
        fused = all(p.is_cuda for p in params + aux)
    return (torch.optim.Adam(params, lr=lr, fused=fused), torch.optim.Adam(aux, lr=aux_lr, fused=fused))


def _step(optimizer):
    """optimizer.step().  torch's fused Adam updates the parameters WITHOUT bumping their version counters, which the weight-pack caches
    are keyed by: masic_amd/fresh.py registers a global optimizer-step hook that bumps them for every fused / capturable group, whoever
    calls step() -- nothing to do here, and nothing for a driver that steps its own optimizer."""
    optimizer.step()


_AUX_FUSED = os.environ.get("MASIC_AUX_FUSED", "1") != "0"      # 0: model.aux_loss().backward() through autograd (A/B timing)


def aux_backward(model):
    """aux_loss = model.aux_loss(); aux_loss.backward() (newtrain_codec_real.py:143-144) -- the sum of EntropyBottleneck.loss() over the
    model's bottlenecks and its gradient, which reaches the quantiles only (every density parameter is detached, reference
    entropy_models.py:345-348).  For device models with <= 4 bottlenecks: two launches that write loss and gradient directly
    (ops.entropy_bottleneck_aux_step; the gradient is accumulated into `quantiles.grad` as autograd would), otherwise autograd."""
    from compressai.entropy_models import EntropyBottleneck
    ebs = [m for m in model.modules() if isinstance(m, EntropyBottleneck)]
    if (_AUX_FUSED and 1 <= len(ebs) <= 4 and torch.is_grad_enabled()
            and all(m.quantiles.is_cuda and m.quantiles.requires_grad and m.quantiles.dtype == torch.float32 for m in ebs)):
        from . import ops
        loss, grads = ops.entropy_bottleneck_aux_step([m._table() for m in ebs], [m.quantiles for m in ebs], [m.tail_mass for m in ebs])
        for m, g in zip(ebs, grads):
            g = g.view_as(m.quantiles)
            if m.quantiles.grad is None:
                m.quantiles.grad = g
            else:
                m.quantiles.grad.add_(g)
        return loss
    aux_loss = model.aux_loss()
    aux_loss.backward()
    return aux_loss


def train_step(model, optimizer, aux_optimizer, d1, d2, h_matrix, lmbda, reducer=None):
    from . import ops
    with ops.batched_packs():       # (the step runs on one stream: the weight packs of all layers are refreshed by one launch)
        return _train_step(model, optimizer, aux_optimizer, d1, d2, h_matrix, lmbda, reducer)


def _train_step(model, optimizer, aux_optimizer, d1, d2, h_matrix, lmbda, reducer=None):
    optimizer.zero_grad()
    aux_optimizer.zero_grad()
    if reducer is not None:
        reducer.arm()
    out_net = model(d1, d2, h_matrix)
    out_criterion = rate_distortion(out_net, d1, d2, lmbda)
    out_criterion["loss"].backward()
    if reducer is not None:
        reducer.finish()
    _step(optimizer)
    aux_loss = aux_backward(model)
    _step(aux_optimizer)
    return out_criterion, aux_loss


def cqe_train_step(model, model2, optimizer, d1, d2, h_matrix, lmbda, reducer=None, reference_graph=False):
    """One step of the CQE stage (coremasic/mywork/newtrain_cqe_real.py:128-174): HSIC `model` in eval mode, Independent_EN
    `model2` in train mode, distortion-only criterion on model2's outputs, Adam on model2's parameters (`optimizer`, :472).

    By default HSIC runs under no_grad: its parameters are not in the stepped optimizer and round() blocks every gradient to
    the analysis side, so the parameter update is identical to the reference's without recording and
    back-propagating through the codec.  `reference_graph=True` records what the reference records -- the two synthesis
    transforms of HSIC as differentiable nodes (model.eval_autograd) -- for timing parity; gradients then also reach
    model.decoder1/2 (never stepped)."""
    if model.training or not model2.training:
        raise RuntimeError("cqe_train_step: HSIC must be in eval mode and Independent_EN in train mode (newtrain_cqe_real.py:130-131)")
    optimizer.zero_grad()
    if reducer is not None:
        reducer.arm()
    if reference_graph:
        prev = getattr(model, "eval_autograd", None)
        model.eval_autograd = True
        try:
            out_net = model(d1, d2, h_matrix)
        finally:
            model.eval_autograd = prev
    else:
        with torch.no_grad():
            out_net = model(d1, d2, h_matrix)
    from . import ops
    with ops.batched_packs():       # (Independent_EN's forward and backward run on one stream; HSIC's multi-stream forward above does not)
        out_net2 = model2(out_net["x1_hat"], out_net["x2_hat"], h_matrix)
        out_criterion = distortion(out_net2, d1, d2, lmbda)
        out_criterion["loss"].backward()
    if reducer is not None:
        reducer.finish()
    _step(optimizer)
    return out_criterion, out_net2
