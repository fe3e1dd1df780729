"""HIP-graph replay of the inference forward.

An eval-mode `HSIC.forward` is ~110 launches of 5-300 us on three streams (coremasic/mywork/MASIC.py: _forward_eval); issued eagerly from Python the host
side costs ~15 % of the step.  Everything in it is capturable -- kernels go to torch's current stream, outputs come from
torch's caching allocator (graph-private pool during capture), weight packs are cached per weight version, the side
streams fork and join through events -- except the host-side float32 evaluation of the 3x3 sampling matrices
(masic_amd/homography.py).  That step is pipelined with the device: the homography is fetched on a copy stream (or
passed as a CPU tensor), the chain runs on the host while the previous replay is still executing, and the results are
uploaded on the copy stream; the main stream only waits for that upload before copying them into the graph's static
buffers and replaying.
"""
import torch

from . import ops
from .homography import warp_matrices_host


class GraphedHSIC:
    """Callable with the signature of `HSIC.forward`; outputs are static tensors that the next call overwrites.
    `h_matrix` may be a CPU tensor or a device tensor (read on a separate copy stream after an event recorded on the
    caller's stream, so a homography just produced there is complete when it is read)."""

    def __init__(self, net, x1, x2, h_matrix, warmup=2, on_stale="recapture"):
        """on_stale: what a call does when a parameter or buffer of `net` has changed since the capture (optimizer step,
        load_state_dict, .to()): "recapture" (default) or "raise".  The captured kernels hold raw pointers to the weight packs
        and tables derived from the parameters; those tensors are kept alive by this object, and they are never replayed
        against parameters they were not derived from."""
        if net.training:
            raise RuntimeError("GraphedHSIC captures the eval-mode forward")
        if on_stale not in ("recapture", "raise"):
            raise ValueError("on_stale must be 'recapture' or 'raise'")
        self.net = net
        self.on_stale = on_stale
        self.warmup = warmup
        dev = x1.device
        self.x1, self.x2 = x1.clone(), x2.clone()
        H, W = x1.shape[-2:]
        self.hw = (H, W)
        B = x1.shape[0]
        self.h = torch.zeros((B, 3, 3), dtype=torch.float32, device=dev)
        self.mfb = torch.zeros((2, B, 3, 3), dtype=torch.float32, device=dev)     # both sampling matrices: one upload
        self.mf, self.mb = self.mfb[0], self.mfb[1]
        # look-ahead mode (__call__(..., next_h_matrix=)): everything on the caller's stream, double-buffered pinned memory
        self.la_h = [torch.empty((B, 3, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.la_m = [torch.empty((2, B, 3, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.la_parity = 0
        self._la_last = [None, None]     # event behind the last upload from la_m[q]
        self._ahead = None
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.h_pinned = torch.empty((B, 3, 3), dtype=torch.float32).pin_memory()
        self.m_pinned = torch.empty((2, B, 3, 3), dtype=torch.float32).pin_memory()
        self.m_stage = [torch.empty((2, B, 3, 3), dtype=torch.float32, device=dev) for _ in range(2)]   # double-buffered upload
        self.stage_free = [torch.cuda.Event(), torch.cuda.Event()]     # main stream has consumed stage buffer p
        for ev in self.stage_free:
            ev.record(torch.cuda.current_stream())
        self.parity = 0
        self._prepare(h_matrix)
        self._capture()

    def _capture(self):
        from . import nn as _mnn
        net = self.net
        if net.training:
            raise RuntimeError("GraphedHSIC captures the eval-mode forward; the model was put in training mode")
        torch.cuda.current_stream().synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):                 # packs weights, warms the allocator
                net(self.x1, self.x2, self.h, warp_matrices=(self.mf, self.mb))
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = net(self.x1, self.x2, self.h, warp_matrices=(self.mf, self.mb))
        # what the captured launches point at: every weight pack / table the forward used, and what they were derived from
        self._packs = _mnn.cached_packs(net)
        self._sources = _mnn.pack_sources(net)
        self._signature = _mnn.pack_signature(self._sources)
        self._precision = (_mnn.get_precision(), ops.get_warp_align_corners())     # settings baked into the captured launches

    def _check_fresh(self):
        from . import nn as _mnn
        if (_mnn.pack_signature(self._sources) == self._signature and (_mnn.get_precision(), ops.get_warp_align_corners()) == self._precision
                and not self.net.training):
            return
        if self.on_stale == "raise":
            raise RuntimeError("GraphedHSIC: the model's parameters / buffers, its mode, the operand precision or the warp convention changed since the "
                               "capture; build a new GraphedHSIC (or pass on_stale='recapture')")
        self._capture()

    def _stage(self, h_matrix, after=None):
        """Host chain for a homography and its upload into a staging buffer; returns (buffer index, upload event).  A device
        h_matrix is read on the copy stream after `after` (default: an event recorded now on the caller's stream), so a homography
        just produced there is complete when it is read; the host waits for that read."""
        cs = self.copy_stream
        if h_matrix.is_cuda:
            if after is None:
                after = torch.cuda.Event()
                after.record(torch.cuda.current_stream())
            cs.wait_event(after)
            with torch.cuda.stream(cs):
                self.h_pinned.copy_(h_matrix.detach().to(torch.float32), non_blocking=True)
            cs.synchronize()
            m = self.h_pinned
        else:
            m = h_matrix.detach().to(torch.float32)
        mf, mb = warp_matrices_host(m, self.hw, self.hw, want_inverse=True)
        self.m_pinned[0].copy_(mf)
        self.m_pinned[1].copy_(mb)
        p = self.parity
        self.parity ^= 1
        stage = self.m_stage[p]
        cs.wait_event(self.stage_free[p])                 # the call before last has copied this stage buffer out (bounds the host's lead)
        with torch.cuda.stream(cs):
            stage.copy_(self.m_pinned, non_blocking=True)
            up = torch.cuda.Event()
            up.record(cs)
        cs.synchronize()                                  # the pinned buffers are reused by the next call
        return p, up

    def _commit(self, staged):
        """The caller's stream takes the staged sampling matrices into the graph's static buffers."""
        p, up = staged
        cur = torch.cuda.current_stream()
        cur.wait_event(up)
        self.mf.copy_(self.m_stage[p][0])
        self.mb.copy_(self.m_stage[p][1])
        self.stage_free[p].record(cur)
        # self.h is only passed through: with precomputed sampling matrices the forward never reads the homography itself

    def _prepare(self, h_matrix):
        self._commit(self._stage(h_matrix))

    @property
    def inputs(self):
        """The graph's static input buffers (x1, x2): a producer that writes the next batch straight into them (an upload,
        a decoder) and passes them back to __call__ saves the device-to-device copy of 2 x B x 3 x H x W floats per step."""
        return self.x1, self.x2

    def __call__(self, x1, x2, h_matrix, next_h_matrix=None):
        """next_h_matrix: the homography of the NEXT call, if the caller has it already (a loader running one batch ahead).  Its
        device -> host read is then enqueued in FRONT of this call's replay, the host evaluates the float32 chain while the replay
        executes, and the next call uploads the finished matrices right behind it -- all on the caller's stream (eager cross-stream
        dependencies are expensive on this ROCm), with double-buffered pinned memory.  Without it every call first
        waits for the previous replay (the read of a device h_matrix must be ordered behind whatever produced it), evaluates the
        chain with the device idle (~0.3 ms of a 2.3 ms step at 8 x 512 x 512) and only then replays."""
        self._check_fresh()
        ahead, self._ahead = self._ahead, None
        if ahead is not None and ahead[0] is h_matrix and ahead[1] == h_matrix._version:
            self.mfb.copy_(ahead[2], non_blocking=True)              # pinned -> static buffers, behind the previous replay
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._la_last[ahead[3]] = ev
        else:
            self._prepare(h_matrix)
        if x1 is not self.x1:
            self.x1.copy_(x1)
        if x2 is not self.x2:
            self.x2.copy_(x2)
        if next_h_matrix is None:
            self.graph.replay()
            return self.out
        q = self.la_parity
        self.la_parity ^= 1
        read = None
        if next_h_matrix.is_cuda:
            self.la_h[q].copy_(next_h_matrix.detach().to(torch.float32), non_blocking=True)      # in front of the replay
            read = torch.cuda.Event()
            read.record(torch.cuda.current_stream())
        self.graph.replay()
        if read is not None:
            read.synchronize()           # = the previous replay has finished: also bounds the host's lead, and la_m[q] (uploaded two calls ago) is free
            m = self.la_h[q]
        else:
            if self._la_last[q] is not None:
                self._la_last[q].synchronize()   # the upload from la_m[q] two calls ago has executed: the buffer may be rewritten
            m = next_h_matrix.detach().to(torch.float32)
        mf, mb = warp_matrices_host(m, self.hw, self.hw, want_inverse=True)
        self.la_m[q][0].copy_(mf)
        self.la_m[q][1].copy_(mb)
        self._ahead = (next_h_matrix, next_h_matrix._version, self.la_m[q], q)
        return self.out


class GraphedTrainStep:
    """One optimisation step of the codec stage -- masic_amd.train.train_step, i.e. newtrain_codec_real.py:135-146: zero_grad x2,
    forward, RD loss, backward, Adam step, aux loss, backward, aux Adam step -- captured as ONE HIP graph and replayed per batch.

    What makes the step capturable: no host synchronisation inside it (masic_amd/loss.py: LazyPSNR; device-side loss-gradient
    scalars; sampling matrices from the device kernel -- reduced operand precision only, the float32 parity path evaluates them on
    the host), torch's Adam in its `capturable` form, every buffer from torch's caching allocator (graph-private pool during capture),
    and zero fills as KERNEL nodes: the library's hipMemsetAsync calls became memset nodes, and the first memset node of a graph
    captured by torch.cuda.graph loses its ordering from the second replay on when the graph is replayed on torch's default stream
    (ROCm 7.2 / torch 2.10; reduced to 25 lines without this library in tools/memset_node_repro.py) -- weight
    gradients read back with garbage, different tensors from run to run (masic_amd/csrc/common.h: masic_zero_async).

    Measured (tools/train_prof.py, TRAIN_PROF_GRAPH=1): the replay is NOT faster than the eager, synchronisation-free step on this
    ROCm -- 20.8 vs 19.9 ms at 8 x 512 x 512, 11.1 vs 11.3 ms at 1 x 512 x 512 -- because the ~880 nodes of a step cost the device
    ~12 us each whether they come from a graph or from the launch queue; the lever for small batches is the node count, not the host.
    The class is kept as the capturable form of the step (and as the regression test for the memset-node finding); the eager
    train_step stays the default.

    The graph owns its optimizers (Adam, capturable); `optimizer` / `aux_optimizer` expose them (state_dict for checkpoints).  Inputs
    are copied into static buffers; the returned criterion dict and aux loss are static tensors overwritten by the next call."""

    def __init__(self, model, d1, d2, h_matrix, lmbda, lr=1e-4, aux_lr=1e-3, warmup=3):
        from . import nn as _mnn
        from .train import train_step
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures the training-mode step: call model.train() first")
        if not _mnn.reduced_precision():
            raise RuntimeError("GraphedTrainStep needs the bf16 / fp8 operand mode (masic_amd.nn.set_precision): the float32 parity path "
                               "evaluates the sampling matrices on the host, which a captured step cannot do")
        self.model, self.lmbda = model, lmbda
        self.optimizer = torch.optim.Adam(list(model.parameters()), lr=lr, capturable=True, fused=True)
        self.aux_optimizer = torch.optim.Adam(list(model.aux_parameters()), lr=aux_lr, capturable=True, fused=True)
        self.d1, self.d2, self.h = d1.clone(), d2.clone(), h_matrix.clone()
        self._params = [p for _, p in model.named_parameters()]
        self._precision = _mnn.get_precision()
        # warm-up on a side stream (allocator pools, weight packs, one-time kernel attributes), then put parameters, buffers and
        # optimizer state back: the warm-up steps are not part of the training run
        saved = {k: v.clone() for k, v in model.state_dict().items()}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(model, self.optimizer, self.aux_optimizer, self.d1, self.d2, self.h, lmbda)
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(saved[k])
                for opt in (self.optimizer, self.aux_optimizer):
                    for st in opt.state.values():
                        for t in st.values():
                            if torch.is_tensor(t):
                                t.zero_()
        cur.wait_stream(side)
        self.optimizer.zero_grad(set_to_none=True)
        self.aux_optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._crit, self._aux = train_step(model, self.optimizer, self.aux_optimizer, self.d1, self.d2, self.h, lmbda)
        self._touch()

    def _touch(self):
        """The replay changes every parameter without Python seeing it: bump the version counters so that the per-version caches of
        weight packs / tables (masic_amd/nn.py) are rebuilt by the next EAGER use of the model instead of serving the packs of the
        capture.  (The graph itself re-packs inside every replay.)"""
        torch._C._increment_version(self._params)       # the list form: one call (per tensor it iterates over the tensor's rows)

    @property
    def inputs(self):
        return self.d1, self.d2, self.h

    def __call__(self, d1, d2, h_matrix):
        from . import nn as _mnn
        from .loss import LazyPSNR
        if not self.model.training or _mnn.get_precision() != self._precision:
            raise RuntimeError("GraphedTrainStep: the model left training mode or the operand precision changed since the capture")
        if d1 is not self.d1:
            self.d1.copy_(d1)
        if d2 is not self.d2:
            self.d2.copy_(d2)
        if h_matrix is not self.h:
            self.h.copy_(h_matrix)
        self.graph.replay()
        self._touch()
        crit = dict(self._crit)
        crit["psnr1"], crit["psnr2"] = LazyPSNR(crit["mse1"]), LazyPSNR(crit["mse2"])      # fresh readers of this step's static scalars
        return crit, self._aux
