"""Host cost of the pieces of one launch through the ctypes boundary (no device synchronisation inside the timed loops)."""
import sys, os, time, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from masic_amd import ops, _lib
x = torch.randn(64, device="cuda"); y = torch.empty_like(x)
def t(name, fn, n=20000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:44s} {dt:7.2f} us")
t("ops._stream()", ops._stream)
t("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
t("torch.empty_like(x)", lambda: torch.empty_like(x))
t("torch.empty(n, int16)", lambda: torch.empty(4096, dtype=torch.int16, device=x.device))
t("ops._p(x) x3", lambda: (ops._p(x), ops._p(y), ops._p(None)))
t("x.is_contiguous(); x.shape", lambda: (x.is_contiguous(), x.shape))
st = ops._stream()
t("raw lib.masic_elementwise (launch only)", lambda: _lib.lib.masic_elementwise(ops._p(x), None, ops._p(y), 64, int(ops.EW_SQUARE), 0.0, 0.0, st), 5000)
t("ops.elementwise(EW_SQUARE, x)", lambda: ops.elementwise(ops.EW_SQUARE, x), 5000)
d = ops.make_conv_desc(2, 32, 16, 16, 32, 3, 3, 1, 1)
t("make_conv_desc", lambda: ops.make_conv_desc(2, 32, 16, 16, 32, 3, 3, 1, 1))
t("torch add (aten) x + y", lambda: x + y, 5000)
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a
    @staticmethod
    def backward(ctx, g): return g
xr = x.clone().requires_grad_(True)
t("autograd.Function.apply (identity)", lambda: F.apply(xr), 5000)
