"""Data-parallel training support: one process per GPU, stereo pairs sharded across ranks, ONE exchange per step --
an averaging all-reduce of the gradients over RCCL/xGMI (backend "nccl" on ROCm), overlapped with backward.

The reference has no distributed code; this is the exchange step north_star adds.  Design for
MI355X: xGMI is point-to-point (7 links per GPU), so a few large flat buckets (default 32 MiB, ~5 for the 140 MB of
HSIC gradients) keep every link busy with long messages instead of many latency-bound small ones.

Buckets are PERSISTENT flat buffers and every parameter's `.grad` is a view into its bucket: backward accumulates straight
into the buffer the collective reduces in place -- no gather copy before the all-reduce and no scatter copy after it
(round 1 `torch.cat`-ed each bucket: 2 x 140 MB of extra device copies per step).  Buckets are filled in reverse
parameter order (the order backward produces gradients) and each bucket's all-reduce is launched from a
post-accumulate-grad hook as soon as its last gradient lands, so communication hides under the rest of backward.
`HSIC.parameters()` hides the entropy-bottleneck parameters (MASIC.py:77-83), so the reducer registers
`named_parameters()` -- all 166 tensors.  Parameters that receive no gradient from the main loss (the two `quantiles`:
`_quantize(..., 'noise')` ignores the medians) are learnt on the first step: they stop gating their bucket's launch and
their `.grad` is reset to None after the step, exactly what a single process sees; the aux-loss backward (a function of
the parameters only, identical on every rank) needs no communication at all.
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
        self.overlap = overlap
        self.params = [p for _, p in reversed(list(module.named_parameters())) if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            if cur and (cur[0].dtype != p.dtype or cur[0].device != p.device):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        # persistent flat gradient storage; element offsets are kept 64-byte aligned so every view starts on a cache line
        self.flat, self._views = [], {}
        for b in self.buckets:
            offs, n = [], 0
            align = max(1, 64 // b[0].element_size())
            for p in b:
                offs.append(n)
                n += (p.numel() + align - 1) // align * align
            flat = torch.zeros(n, dtype=b[0].dtype, device=b[0].device)
            self.flat.append(flat)
            for p, o in zip(b, offs):
                self._views[id(p)] = flat[o:o + p.numel()].view_as(p)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._expected = None            # ids of the parameters that receive a gradient (learnt on the first step)
        self._got = set()
        self._need = [set(id(p) for p in b) for b in self.buckets]      # per bucket: expected ids whose gradient has not landed yet
        self._inflight = {}
        self._events = [[] for _ in self.buckets]      # per bucket: events of the streams its gradients were produced on (see _on_grad)
        self._armed = False
        self._cancelled = False          # this step's overlap is off: every bucket is launched from finish()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def bucket_plan(self):
        """[(number of tensors, bytes)] per bucket, in launch order."""
        return [(len(b), f.numel() * f.element_size()) for b, f in zip(self.buckets, self.flat)]

    def arm(self):
        """Call after optimizer.zero_grad() and before the main-loss backward of each step: zeroes the buckets and (re)binds
        every parameter's .grad to its view, so that autograd accumulates into the buffers the collective reduces."""
        for flat in self.flat:
            flat.zero_()
        for p in self.params:
            p.grad = self._views[id(p)]
        if self._expected is None:
            self._need = [set(id(p) for p in b) for b in self.buckets]
        else:
            self._need = [set(id(p) for p in b if id(p) in self._expected) for b in self.buckets]
        self._got = set()
        self._inflight = {}
        self._events = [[] for _ in self.buckets]
        self._cancelled = False
        self._armed = True

    # ---- called by autograd when a parameter's gradient is complete
    def _on_grad(self, p):
        """Readiness is tracked PER PARAMETER: bucket i is launched when every expected parameter of it has fired exactly once.
        A parameter firing a second time (a second backward between arm() and finish(): gradient accumulation) or one outside the
        learnt set (another graph than the step before) switches this step's overlap off -- all buckets then go out from finish(),
        after the last backward.  If the bucket it belongs to is ALREADY being reduced, autograd has just accumulated into a buffer
        an asynchronous collective is reading and writing: nothing can repair that, so it raises (use overlap=False to accumulate
        gradients over several backwards)."""
        if not self._armed:
            return
        if p.grad is None or p.grad.data_ptr() != self._views[id(p)].data_ptr():
            # autograd replaced the tensor (cannot happen while .grad is bound before backward; kept as a hard check)
            raise RuntimeError("GradientAllReducer: a parameter's .grad is no longer the bucket view; call arm() after zero_grad()")
        pid = id(p)
        i = self._bucket_of[pid]
        repeat = pid in self._got
        unexpected = self._expected is not None and pid not in self._expected
        self._got.add(pid)
        if repeat or unexpected:
            if self._inflight.get(i) is not None:
                raise RuntimeError("GradientAllReducer(overlap=True): a gradient was accumulated into a bucket whose all-reduce is already in "
                                   "flight (" + ("a second backward between arm() and finish()" if repeat else "a parameter the previous "
                                   "steps' loss did not reach") + "); construct the reducer with overlap=False for gradient accumulation / "
                                   "changing graphs")
            self._cancelled = True
            return
        if p.is_cuda and self.overlap:
            # The training forward runs branches on side streams and autograd runs their backward there: the gradients of one bucket are
            # produced on several streams, and the hook that completes the bucket runs on ONE of them.  Every hook leaves an event on its
            # stream; the launch makes its stream wait for all of them before the collective reads the bucket.
            ev = torch.cuda.Event()
            ev.record()
            self._events[i].append(ev)
        self._need[i].discard(pid)
        if not self._need[i] and self.overlap and not self._cancelled:
            self._launch(i)

    def _launch(self, i):
        if i in self._inflight:
            return
        work = None
        if self._events[i]:
            cur = torch.cuda.current_stream(self.flat[i].device)
            for ev in self._events[i]:
                cur.wait_event(ev)
            self._events[i] = []
        if self.world > 1:
            work = dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight[i] = work

    def finish(self):
        """Call after backward: launches the buckets no hook completed, waits, averages in place, and resets to None the
        .grad of parameters the loss did not reach (as a single process would see them)."""
        self._armed = False
        for i in range(len(self.buckets)):
            self._launch(i)
        for i, work in self._inflight.items():
            if work is not None:
                work.wait()
            if self.average and self.world > 1:
                self.flat[i].div_(self.world)
        self._inflight = {}
        if self._expected is None and not self._cancelled:
            self._expected = set(self._got)
        elif self._cancelled or self._got != self._expected:
            # the graph changed (another loss, accumulation): relearn on the next step.  This step is correct all the same: a bucket is
            # only ever launched early when all of ITS expected gradients have landed once and nothing unexpected has fired before, and a
            # gradient landing in an in-flight bucket raises in _on_grad
            self._expected = None
        for p in self.params:
            if id(p) not in self._got:
                p.grad = None

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
