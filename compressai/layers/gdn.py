"""Generalized divisive normalisation (reference compressai/layers/gdn.py:41-121).

y_i = x_i / sqrt(beta_i + sum_j gamma_ij x_j^2)   (inverse: multiply by the square root).
Parameters are stored reparametrised exactly as in the reference (so checkpoints interchange);
the reparametrisation, the 1x1 contraction, rsqrt/sqrt and the product run in one HIP kernel
(masic_amd/csrc/gdn.hip)."""
import torch
import torch.nn as nn

from compressai.ops.parametrizers import NonNegativeParametrizer
from masic_amd import ops as _hip

__all__ = ["GDN", "GDN1"]


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_min = float(beta_min)
        self.beta_reparam = NonNegativeParametrizer(minimum=self.beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.gamma.requires_grad):
            from masic_amd.autograd import GdnFn
            return GdnFn.apply(x, self.beta, self.gamma, self.inverse, self.beta_min)
        from masic_amd import nn as _mnn
        return _hip.gdn(x, self.beta.detach(), self.gamma.detach(), inverse=self.inverse, beta_min=self.beta_min,
                        prec=_mnn._PRECISION)


class GDN1(GDN):
    """Simplified GDN (|x| instead of x^2; reference gdn.py:95-121). Not used by MASIC: no HIP kernel."""

    def forward(self, x):
        raise NotImplementedError("GDN1 is outside the MASIC hot path (SURVEY.md section 2 #3); no MI355X kernel is built for it")
