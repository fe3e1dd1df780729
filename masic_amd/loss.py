"""Stereo rate-distortion criterion on the HIP reductions.

Reference: coremasic/mywork/newtrain_codec_real.py:66-87 (`RateDistortionLoss.forward(output, target1, target2)`):
  bpp_loss = sum_{y1,y2,z1,z2} sum(log(lik)) / (-ln2 * B*H*W);  mse_loss = MSE(x1_hat,x1) + MSE(x2_hat,x2);
  loss = lmbda * 255^2 * mse_loss + bpp_loss;  psnr_i = 10*log10(1/mse_i).
The four log-sums and two squared-error sums run as deterministic two-stage device reductions
(masic_amd/csrc/misc.hip) with a float64 final stage; the scalars stay on the device until asked for.
"""
import math

from . import ops


class RateDistortionLoss:
    """Callable with the reference's signature: criterion(out_net, d1, d2) -> dict."""

    def __init__(self, lmbda=1e-2):
        self.lmbda = lmbda

    def __call__(self, output, target1, target2):
        return rate_distortion(output, target1, target2, self.lmbda)


class LazyPSNR:
    """mse2psnr(mse) = 10 log10(1 / mse) (newtrain_codec_real.py:62-65) of a device scalar, as a float-like value that is read back --
    and makes the host wait for the stream -- when it is first USED (formatted, compared, added to a meter), not when the criterion
    returns: the reference's `math.log10(1 / mse)` on a device tensor stalls the host once per step right after the forward, and the
    device then idles until the host has caught up with the backward's launches."""
    __slots__ = ("_mse", "_v")

    def __init__(self, mse):
        self._mse, self._v = mse, None

    def __float__(self):
        if self._v is None:
            self._v = 10 * math.log10(1 / float(self._mse))
            self._mse = None
        return self._v

    def __repr__(self):
        return repr(float(self))

    def __format__(self, spec):
        return format(float(self), spec)

    def __add__(self, o): return float(self) + o
    def __radd__(self, o): return o + float(self)
    def __sub__(self, o): return float(self) - o
    def __rsub__(self, o): return o - float(self)
    def __mul__(self, o): return float(self) * o
    def __rmul__(self, o): return o * float(self)
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __neg__(self): return -float(self)
    def __abs__(self): return abs(float(self))
    def __lt__(self, o): return float(self) < o
    def __le__(self, o): return float(self) <= o
    def __gt__(self, o): return float(self) > o
    def __ge__(self, o): return float(self) >= o
    def __eq__(self, o): return float(self) == o
    def __hash__(self): return hash(float(self))


def rate_distortion(output, target1, target2, lmbda):
    import torch
    liks = output["likelihoods"]
    if torch.is_grad_enabled() and any(t.requires_grad for t in (output["x1_hat"], output["x2_hat"], *liks.values())):
        from .autograd import RateDistortionFn
        loss, mse1, mse2, bpp, *per = RateDistortionFn.apply(lmbda, target1, target2, output["x1_hat"], output["x2_hat"], *liks.values())
        out = {"bpp_loss": bpp.float(), "mse_loss": (mse1 + mse2).float(), "loss": loss, "mse1": mse1, "mse2": mse2}
        out.update({"bpp_" + k: v for k, v in zip(liks.keys(), per)})
        out["psnr1"], out["psnr2"] = LazyPSNR(mse1), LazyPSNR(mse2)
        return out
    B, _, H, W = target1.shape
    num_pixels = B * H * W
    per = {k: ops.sum_log(v.contiguous()) / (-math.log(2) * num_pixels) for k, v in output["likelihoods"].items()}
    bpp = sum(per.values())
    mse1 = ops.sse(output["x1_hat"].contiguous(), target1.contiguous()) / target1.numel()
    mse2 = ops.sse(output["x2_hat"].contiguous(), target2.contiguous()) / target2.numel()
    mse = mse1 + mse2
    out = {"bpp_loss": bpp.float(), "mse_loss": mse.float(), "loss": (lmbda * 255 ** 2 * mse + bpp).float(),
           "mse1": mse1, "mse2": mse2}
    out.update({"bpp_" + k: v for k, v in per.items()})
    out["psnr1"] = LazyPSNR(mse1)
    out["psnr2"] = LazyPSNR(mse2)
    return out


def distortion(output, target1, target2, lmbda):
    """The criterion of the CQE stage (coremasic/mywork/newtrain_cqe_real.py:66-96, kind=0): loss = lmbda * 255^2 *
    (MSE(x1_hat, d1) + MSE(x2_hat, d2)) on the outputs of Independent_EN, plus psnr1 / psnr2 (the reference's MS-SSIM entries
    are reporting only and need pytorch_msssim, which is outside the path)."""
    import torch
    x1_hat, x2_hat = output["x1_hat"], output["x2_hat"]
    if torch.is_grad_enabled() and (x1_hat.requires_grad or x2_hat.requires_grad):
        from .autograd import RateDistortionFn
        loss, mse1, mse2, _ = RateDistortionFn.apply(lmbda, target1, target2, x1_hat, x2_hat)      # no likelihood terms: distortion only
        mse = mse1 + mse2
    else:
        mse1 = ops.sse(x1_hat.detach().contiguous(), target1.contiguous()) / target1.numel()
        mse2 = ops.sse(x2_hat.detach().contiguous(), target2.contiguous()) / target2.numel()
        mse = mse1 + mse2
        loss = (lmbda * 255 ** 2 * mse).float()
    return {"mse_loss": mse.float(), "loss": loss, "mse1": mse1, "mse2": mse2,
            "psnr1": LazyPSNR(mse1), "psnr2": LazyPSNR(mse2)}
