"""Non-negative reparametrisation used by GDN (reference compressai/ops/parametrizers.py:38-64).
`GDN.forward` applies it inside the HIP kernel; this module keeps the buffers (`pedestal`,
`lower_bound.bound`) that appear in checkpoints, `init()`, and a standalone forward."""
import torch
import torch.nn as nn

from .bound_ops import LowerBound


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        ped = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([ped]))
        self.lower_bound = LowerBound((self.minimum + ped) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        bounded = self.lower_bound(x)
        return bounded * bounded - self.pedestal
