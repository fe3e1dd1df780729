"""One optimisation step of the codec stage, as the reference's train_epoch does it
(coremasic/mywork/newtrain_codec_real.py:135-146): zero_grad x2 -> forward -> RD loss -> backward -> Adam step ->
aux loss -> backward -> aux Adam step.  Optimizers are torch's (plumbing); every tensor op of forward and backward
is a HIP launch.  With a GradientAllReducer the main-loss gradients are averaged across ranks before the step."""
import torch

from .loss import rate_distortion


def make_optimizers(model, lr=1e-4, aux_lr=1e-3):
    """newtrain_codec_real.py:434-435 (note: aux optimizer owns ALL entropy-bottleneck parameters, MASIC.py:85-94)."""
    return (torch.optim.Adam(model.parameters(), lr=lr), torch.optim.Adam(model.aux_parameters(), lr=aux_lr))


def train_step(model, optimizer, aux_optimizer, d1, d2, h_matrix, lmbda, reducer=None):
    optimizer.zero_grad()
    aux_optimizer.zero_grad()
    if reducer is not None:
        reducer.arm()
    out_net = model(d1, d2, h_matrix)
    out_criterion = rate_distortion(out_net, d1, d2, lmbda)
    out_criterion["loss"].backward()
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    aux_loss = model.aux_loss()
    aux_loss.backward()
    aux_optimizer.step()
    return out_criterion, aux_loss
