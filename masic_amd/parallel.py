"""Data-parallel training support: one process per GPU, stereo pairs sharded across ranks, ONE exchange per step --
an averaging all-reduce of the gradients over RCCL/xGMI (backend "nccl" on ROCm), overlapped with backward.

The reference has no distributed code; this is the exchange step north_star adds.  Design for
MI355X: xGMI is point-to-point (7 links per GPU), so a few large flat buckets (default 32 MiB, ~5 for the 140 MB of
HSIC gradients) keep every link busy with long messages instead of many latency-bound small ones.  Buckets are
filled in reverse parameter order (the order backward produces gradients) and each bucket's all-reduce is launched
from a post-accumulate-grad hook as soon as its last gradient lands, so communication hides under the rest of
backward.  `HSIC.parameters()` hides the entropy-bottleneck parameters (MASIC.py:77-83), so the reducer registers
`named_parameters()` -- all 166 tensors; parameters that receive no gradient from the main loss (the two `quantiles`)
are skipped consistently on every rank, and the aux-loss backward (a function of the parameters only, identical on
every rank) needs no communication at all.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items units (stereo pairs) for `rank` of `world`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradientAllReducer:
    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, average=True, overlap=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        self.params = [p for _, p in reversed(list(module.named_parameters())) if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._pending = [len(b) for b in self.buckets]
        self._inflight = {}
        self._hooks = []
        self._armed = False
        if overlap and self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ---- overlap path: called by autograd when a parameter's gradient is complete
    def arm(self):
        """Call before the main-loss backward of each step."""
        self._pending = [len(b) for b in self.buckets]
        self._inflight = {}
        self._armed = True

    def _on_grad(self, p):
        if not self._armed:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def _launch(self, i):
        ps = [p for p in self.buckets[i] if p.grad is not None]
        if not ps or i in self._inflight:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self._inflight[i] = (ps, flat, work)

    def finish(self):
        """Call after backward: launches the buckets no hook completed (parameters without gradient), waits, and
        writes the averaged gradients back."""
        self._armed = False
        for i in range(len(self.buckets)):
            if i not in self._inflight:
                self._launch(i)
        for i, (ps, flat, work) in self._inflight.items():
            if work is not None:
                work.wait()
            if self.average and self.world > 1:
                flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._inflight = {}

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
