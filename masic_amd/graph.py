"""HIP-graph replay of the inference forward.

An eval-mode `HSIC.forward` is ~130 dependent launches of 20-300 us each; issued eagerly from Python the host side
costs about a tenth of the step.  Everything in it is capturable -- kernels go to torch's current stream, outputs
come from torch's caching allocator (graph-private pool during capture), weight packs are cached per weight version --
except the host-side float32 evaluation of the 3x3 sampling matrices (masic_amd/homography.py), which therefore runs
before each replay and is copied into static device tensors together with the images.
"""
import torch

from .homography import warp_matrices


class GraphedHSIC:
    """Callable with the signature of `HSIC.forward`; outputs are static tensors that the next call overwrites."""

    def __init__(self, net, x1, x2, h_matrix, warmup=2):
        if net.training:
            raise RuntimeError("GraphedHSIC captures the eval-mode forward")
        self.net = net
        self.x1, self.x2, self.h = x1.clone(), x2.clone(), h_matrix.clone()
        H, W = x1.shape[-2:]
        self.hw = (H, W)
        mf, mb = warp_matrices(self.h, self.hw, self.hw, want_inverse=True)
        self.mf, self.mb = mf.clone(), mb.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # packs weights, warms the allocator
                net(self.x1, self.x2, self.h, warp_matrices=(self.mf, self.mb))
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = net(self.x1, self.x2, self.h, warp_matrices=(self.mf, self.mb))

    def __call__(self, x1, x2, h_matrix):
        mf, mb = warp_matrices(h_matrix, self.hw, self.hw, want_inverse=True)    # host float32 chain (sync on h_matrix only)
        self.x1.copy_(x1); self.x2.copy_(x2); self.h.copy_(h_matrix)
        self.mf.copy_(mf); self.mb.copy_(mb)
        self.graph.replay()
        return self.out
