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
    """Simplified GDN: y_i = x_i / (beta_i + sum_j gamma_ij |x_j|) (reference gdn.py:95-121).  Not used by MASIC; inference-only
    float32 kernel (masic_gdn1_fwd) for the API surface."""

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.gamma.requires_grad):
            raise NotImplementedError("GDN1 has no backward on the HIP path (MASIC uses GDN); call it under torch.no_grad()")
        return _hip.gdn1(x.contiguous(), self.beta.detach(), self.gamma.detach().contiguous(), inverse=self.inverse, beta_min=self.beta_min)
