"""LowerBound with the pass-through gradient rule (reference compressai/ops/bound_ops.py:36-80).

Inside the MASIC hot path the bound is fused into the GDN / EntropyBottleneck / GMM kernels; this
module is the standalone form (same buffer name `bound` so state dicts match)."""
import torch
import torch.nn as nn

from masic_amd import ops as _hip


class _LowerBoundHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x)
        ctx.bound = float(bound)
        return _hip.lower_bound(x.contiguous(), ctx.bound)

    @staticmethod
    def backward(ctx, grad):
        (x,) = ctx.saved_tensors
        return _hip.lower_bound_bwd(x.contiguous(), grad.contiguous(), ctx.bound), None


class LowerBound(nn.Module):
    """max(x, bound); d/dx is the identity where x >= bound or where the gradient pushes x upward."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundHip.apply(x, self._bound_value())

    def _bound_value(self):
        # cached host copy: avoids a device->host sync per call
        v = getattr(self, "_bound_host", None)
        if v is None:
            v = float(self.bound.detach().cpu().item())
            object.__setattr__(self, "_bound_host", v)
        return v
