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
        self.mf = torch.zeros((B, 3, 3), dtype=torch.float32, device=dev)
        self.mb = torch.zeros((B, 3, 3), dtype=torch.float32, device=dev)
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

    def _prepare(self, h_matrix):
        """Host chain for this call's homography and its upload; the main stream is made to wait for the upload only."""
        cs = self.copy_stream
        if h_matrix.is_cuda:
            produced = torch.cuda.Event()
            produced.record(torch.cuda.current_stream())     # the homography may just have been produced on the caller's stream
            cs.wait_event(produced)
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
        cur = torch.cuda.current_stream()
        cur.wait_event(up)
        self.mf.copy_(stage[0])
        self.mb.copy_(stage[1])
        self.stage_free[p].record(cur)
        # self.h is only passed through: with precomputed sampling matrices the forward never reads the homography itself

    @property
    def inputs(self):
        """The graph's static input buffers (x1, x2): a producer that writes the next batch straight into them (an upload,
        a decoder) and passes them back to __call__ saves the device-to-device copy of 2 x B x 3 x H x W floats per step."""
        return self.x1, self.x2

    def __call__(self, x1, x2, h_matrix):
        self._check_fresh()
        self._prepare(h_matrix)
        if x1 is not self.x1:
            self.x1.copy_(x1)
        if x2 is not self.x2:
            self.x2.copy_(x2)
        self.graph.replay()
        return self.out
