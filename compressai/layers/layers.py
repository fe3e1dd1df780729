"""Conv building blocks used by MASIC (reference compressai/layers/layers.py:52-95,160-190).
Blocks that only the upstream cheng2020/ssf models use (AttentionBlock, ResidualBlockWithStride,
ResidualBlockUpsample, subpel_conv3x3, QReLU) are outside the hot path and not provided."""
import torch
import torch.nn as nn

from masic_amd import ops as _hip
from masic_amd.fresh import stamp as _stamp
from masic_amd.nn import Conv2d

__all__ = ["MaskedConv2d", "ResidualBlock", "conv3x3", "conv1x1"]


class MaskedConv2d(Conv2d):
    """PixelCNN-style masked convolution (type 'A' hides the current and all later positions).

    As in the reference (layers.py:77) the stored weight is multiplied by the mask in place on every
    forward, so masked taps read exactly 0 in state dicts; the kernel then contracts only the live
    taps (12 of 25 for the 5x5 context model)."""

    def __init__(self, *args, mask_type="A", **kwargs):
        super().__init__(*args, **kwargs)
        if mask_type not in ("A", "B"):
            raise ValueError(f'Invalid "mask_type" value "{mask_type}"')
        if mask_type != "A":
            raise NotImplementedError("only mask type 'A' is on the MASIC path")
        self.masked_conv = True
        mask = torch.ones_like(self.weight.data)
        _, _, kh, kw = mask.shape
        mask[:, :, kh // 2, kw // 2:] = 0
        mask[:, :, kh // 2 + 1:] = 0
        self.register_buffer("mask", mask)

    def zero_masked_taps(self):
        """The reference multiplies weight.data by the mask in place on every forward (layers.py:77); once a weight version is
        masked the product is idempotent, so it is skipped until the parameter changes (optimizer step, load_state_dict)."""
        key = (_stamp(self.weight), self.weight.data_ptr())
        if self.__dict__.get("_masked_version") != key:
            _hip.mul_inplace(self.weight.data, self.mask)
            self.__dict__["_masked_version"] = (_stamp(self.weight), self.weight.data_ptr())

    def run(self, x, **kw):
        self.zero_masked_taps()
        return super().run(x, **kw)


def conv3x3(in_ch, out_ch, stride=1):
    return Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def conv1x1(in_ch, out_ch, stride=1):
    return Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


class ResidualBlock(nn.Module):
    """conv3x3 -> LeakyReLU -> conv3x3 -> LeakyReLU, plus identity (1x1-projected if widths differ).
    Used by the CQE network (MASIC.py:149-164)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x, extra_identity=None):
        """out = leaky(conv2(leaky(conv1(x)))) + identity [+ extra_identity]; activations and both adds run in the
        conv epilogues (`extra_identity` lets Enhancement_Block fold its own skip into the last block)."""
        identity = x if self.skip is None else self.skip.run(x)
        t = self.conv1.run(x, act=_hip.ACT_LEAKY)
        return self.conv2.run(t, act=_hip.ACT_LEAKY, res1=identity, res2=extra_identity)

    def f16k_supported(self, B, H, W):
        return self.skip is None and self.conv1.f16k_supported(B, H, W) and self.conv2.f16k_supported(B, H, W)

    def forward_f16k(self, x16, B, H, W, extra16=None, out16=None, out_ctot=None, out_coff=0):
        """The same block on F16K activations (bf16 operands, masic_amd/csrc/conv_f16k.hip): both LeakyReLUs and both adds in the
        epilogues; the result may go straight into a channel slice of a wider F16K buffer (the reference's torch.cat target)."""
        C = self.conv1.in_channels
        t = self.conv1.run_f16k_res(x16, B, H, W, act=_hip.ACT_LEAKY)
        return self.conv2.run_f16k_res(t, B, H, W, act=_hip.ACT_LEAKY, res1=x16, res2=extra16, res_ctot=C, out16=out16, out_ctot=out_ctot, out_coff=out_coff)
