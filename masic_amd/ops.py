"""Tensor-level entry points of the HIP path: each function checks shapes on the host, allocates
its outputs with torch (device memory is torch's; the library never allocates) and launches the
C-ABI kernel on torch's current HIP stream.  No function here computes on the CPU and none falls
back to ATen ops: CPU tensors raise.
"""
import ctypes

import torch

from . import _lib
from .fresh import stamp as _stamp
from ._lib import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SOFTMAX_C, INOP_ABS, INOP_NONE, INOP_ROUND,  # noqa: F401
                   ConvDesc, check, lib)

LIK_BOUND = 1e-9
SCALE_BOUND = 0.11

import os as _os

# GDN fused into the epilogue of a bf16-operand convolution: 1 = one bf16 product gamma^ x x^2 (error ~2^-10 of the result, the order
# of the bf16 rounding the result gets when it is stored; the default), 3 = the bf16 hi/lo split product (~2^-16).
_FUSED_GDN_PRODUCTS = 3 if _os.environ.get("MASIC_GDN_X3", "0") == "1" else 1


def set_fused_gdn_products(n):
    global _FUSED_GDN_PRODUCTS
    if n not in (1, 3):
        raise ValueError("fused GDN: 1 or 3 bf16 products")
    _FUSED_GDN_PRODUCTS = n


def _gdn_flags(inverse, products=None):
    """the gdn_inverse argument of the fused kernels: bit 0 = inverse GDN, bit 1 = three-product contraction"""
    return int(bool(inverse)) | (2 if (products or _FUSED_GDN_PRODUCTS) == 3 else 0)


def _dev(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"masic_amd: {name} must be a CUDA (HIP) tensor -- the MI355X path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"masic_amd: {name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"masic_amd: {name} must be contiguous (NCHW)")
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device


def zeros(shape, dtype=torch.float32, device=None):
    """A zero tensor by an explicit fill kernel.  (torch.zeros is a fill kernel as well on this build -- rocprofv3 shows FillFunctor for
    zeros / zero_ / fill_(0) -- so this is equivalent; what must not be used for fills inside a captured step is hipMemsetAsync,
    masic_amd/csrc/common.h: masic_zero_async.)"""
    return torch.empty(shape, dtype=dtype, device=device).fill_(0)


def _stream():
    # the current HIP stream's handle; torch.cuda.current_stream() builds a Stream object through three layers of Python (8 us of
    # the ~13 us a launch costs the host, and a training step makes ~1000 of them), the raw accessor is one C call
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------------------------- conv
def conv_out_hw(Hi, Wi, KH, KW, stride, pad, transposed):
    if transposed:
        return ((Hi - 1) * stride - 2 * pad + KH + stride - 1, (Wi - 1) * stride - 2 * pad + KW + stride - 1)
    return ((Hi + 2 * pad - KH) // stride + 1, (Wi + 2 * pad - KW) // stride + 1)


# ---- host cost.  A training step makes ~620 calls through the C ABI of which ~290 launch nothing: "is this shape supported", "how many
# bytes does the pack / workspace take" -- 2-3 us each through ctypes, and with the side streams of the training forward the step is bound
# by the host's launch rate.  Descriptors are therefore interned (one immutable ConvDesc per argument tuple: nothing in the package writes
# to a descriptor after it is made), the answers to descriptor queries are kept on the descriptor object, and the pure-integer size
# queries are lru-cached.
_DESC_CACHE = {}


def make_conv_desc(*args, **kw):
    key = (args, tuple(sorted(kw.items()))) if kw else args
    d = _DESC_CACHE.get(key)
    if d is None:
        if len(_DESC_CACHE) > 4096:
            _DESC_CACHE.clear()
        d = _DESC_CACHE[key] = _make_conv_desc(*args, **kw)
    return d


def _desc_query(name, desc):
    """lib.<name>(&desc) for a query without side effects, answered once per descriptor object."""
    memo = desc.__dict__.get("_memo")
    if memo is None:
        memo = desc.__dict__["_memo"] = {}
    v = memo.get(name)
    if v is None:
        v = memo[name] = getattr(lib, name)(ctypes.byref(desc))
    return v


import functools as _functools


@_functools.lru_cache(maxsize=4096)
def _int_query(name, *args):
    """lib.<name>(ints...) for a pure size query."""
    return getattr(lib, name)(*args)


def _make_conv_desc(B, Cin, Hi, Wi, Cout, KH, KW, stride, pad, transposed=False, masked=False,
                    in_ctot=None, in_coff=0, out_ctot=None, out_coff=0, in_op=INOP_NONE, act=ACT_NONE,
                    gate_ctot=0, gate_c=0, prec=_lib.PREC_F32):
    Ho, Wo = conv_out_hw(Hi, Wi, KH, KW, stride, pad, transposed)
    return ConvDesc(B=B, Cin=Cin, Hi=Hi, Wi=Wi, in_ctot=in_ctot if in_ctot is not None else Cin, in_coff=in_coff,
                    Cout=Cout, Ho=Ho, Wo=Wo, out_ctot=out_ctot if out_ctot is not None else Cout, out_coff=out_coff,
                    KH=KH, KW=KW, stride=stride, pad=pad, transposed=int(transposed), masked=int(masked),
                    in_op=in_op, act=act, gate_ctot=gate_ctot, gate_c=gate_c, prec=prec)


def pack_conv_weight(weight, desc):
    """Re-lays a Conv2d / ConvTranspose2d weight out as [phase-tap][ci][co] (include/masic_hip.h)."""
    _dev(weight, "weight")
    nbytes = _desc_query("masic_conv_packed_bytes", desc)
    if nbytes == 0:
        check(-1, "conv_packed_bytes")
    packed = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=weight.device)
    check(lib.masic_conv_pack_weight(_p(weight), _p(packed), ctypes.byref(desc), _stream()), "conv_pack_weight")
    return packed


class KernelTimer:
    """HIP-event timing of conv launches on torch's current stream (the stream the kernels are launched on),
    keyed by kernel symbol; used by bench.py for the live roofline figure."""
    def __init__(self, only=None):
        self.records = []
        self.only = only      # bracket launches of this kernel symbol only (events between every launch cost ~8% of a step)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for variant, launches, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(variant, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += launches
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def conv_algorithmic_work(desc):
    """(flops, bytes) of one conv layer: dense MACs incl. masked taps; activations in + out + weights."""
    taps = desc.KH * desc.KW
    if desc.transposed:
        macs = desc.B * desc.Cin * desc.Cout * taps * desc.Hi * desc.Wi
    else:
        macs = desc.B * desc.Cout * desc.Cin * taps * desc.Ho * desc.Wo
    nbytes = 4 * (desc.B * desc.Cin * desc.Hi * desc.Wi + desc.B * desc.Cout * desc.Ho * desc.Wo + desc.Cin * desc.Cout * taps)
    return 2.0 * macs, float(nbytes)


def conv2d(x, packed, bias, desc, out=None, gate=None, res1=None, res2=None):
    if _timer is not None:
        n = ctypes.c_int(0)
        lib.masic_conv_variant(ctypes.byref(desc), ctypes.byref(n))
        buf = ctypes.create_string_buffer(96)
        lib.masic_conv_kernel_name(ctypes.byref(desc), buf, 96)
        variant = buf.value.decode()
        if _timer.only is not None and variant != _timer.only:
            return _conv2d(x, packed, bias, desc, out, gate, res1, res2)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        y = _conv2d(x, packed, bias, desc, out, gate, res1, res2)
        e1.record()
        flops, nbytes = conv_algorithmic_work(desc)
        _timer.records.append((variant, n.value, flops, nbytes, e0, e1))
        return y
    return _conv2d(x, packed, bias, desc, out, gate, res1, res2)


def _conv2d(x, packed, bias, desc, out=None, gate=None, res1=None, res2=None):
    _dev(x, "conv input")
    if x.dim() != 4 or x.shape[0] != desc.B or x.shape[1] != desc.in_ctot or x.shape[2] != desc.Hi or x.shape[3] != desc.Wi:
        raise RuntimeError(f"masic_amd.conv2d: input {tuple(x.shape)} does not match descriptor "
                           f"[{desc.B},{desc.in_ctot},{desc.Hi},{desc.Wi}]")
    if out is None:
        out = torch.empty((desc.B, desc.out_ctot, desc.Ho, desc.Wo), dtype=torch.float32, device=x.device)
    else:
        _dev(out, "conv output")
        if tuple(out.shape) != (desc.B, desc.out_ctot, desc.Ho, desc.Wo):
            raise RuntimeError(f"masic_amd.conv2d: output buffer {tuple(out.shape)} does not match descriptor")
    if gate is not None:
        _dev(gate, "gate")
        if tuple(gate.shape) != (desc.B, desc.gate_ctot, desc.Ho, desc.Wo):
            raise RuntimeError(f"masic_amd.conv2d: gate {tuple(gate.shape)} does not match descriptor")
    if bias is not None:
        _dev(bias, "bias")
    for r in (res1, res2):
        if r is not None:
            _dev(r, "residual")
            if tuple(r.shape) != (desc.B, desc.Cout, desc.Ho, desc.Wo):
                raise RuntimeError(f"masic_amd.conv2d: residual {tuple(r.shape)} does not match the output")
    check(lib.masic_conv2d_fwd_ex(_p(x), _p(packed), _p(bias), _p(gate), _p(res1), _p(res2), _p(out), ctypes.byref(desc), _stream()),
          "conv2d_fwd")
    return out


# --------------------------------------------------------------------------------------------- GDN
def gdn(x, beta, gamma, inverse=False, beta_min=1e-6, prec=_lib.PREC_F32):
    _dev(x, "gdn input"); _dev(beta, "beta"); _dev(gamma, "gamma")
    B, C, H, W = x.shape
    if beta.numel() != C or tuple(gamma.shape) != (C, C):
        raise RuntimeError("masic_amd.gdn: parameter shapes do not match the input channels")
    y = torch.empty_like(x)
    check(lib.masic_gdn_fwd_ex(_p(x), _p(beta), _p(gamma), _p(y), B, C, H, W, int(inverse), float(beta_min), int(prec), _stream()), "gdn_fwd")
    return y


def gdn1(x, beta, gamma, inverse=False, beta_min=1e-6):
    """Simplified GDN (|x| in place of x^2, no square root; compressai/layers/gdn.py:95-121), float32."""
    _dev(x, "gdn1 input"); _dev(beta, "beta"); _dev(gamma, "gamma")
    B, C, H, W = x.shape
    if beta.numel() != C or tuple(gamma.shape) != (C, C):
        raise RuntimeError("masic_amd.gdn1: parameter shapes do not match the input channels")
    y = torch.empty_like(x)
    check(lib.masic_gdn1_fwd(_p(x), _p(beta), _p(gamma), _p(y), B, C, H, W, int(inverse), float(beta_min), _stream()), "gdn1_fwd")
    return y


def gdn_f16k(x, beta, gamma, inverse=False, beta_min=1e-6):
    """GDN of a 128-channel float32 NCHW tensor -> F16K bf16 buffer (input of conv2d_f16k)."""
    _dev(x, "gdn input"); _dev(beta, "beta"); _dev(gamma, "gamma")
    B, C, H, W = x.shape
    y = torch.empty(B * C * H * W, dtype=torch.int16, device=x.device)
    check(lib.masic_gdn_fwd_f16k(_p(x), _p(beta), _p(gamma), _p(y), B, C, H, W, int(inverse), float(beta_min), _stream()), "gdn_fwd_f16k")
    return y


# --------------------------------------------------------------------------------------------- entropy
def quantize(x, mode, noise=None, out=None, out_coff=0, gate=None, gate_c=0):
    """mode: 'dequantize' (round), 'noise' (x + noise) or 'copy' (identity; for gated writes into a concat slice)."""
    _dev(x, "quantize input")
    B, C, H, W = x.shape
    m = {"dequantize": 0, "noise": 1, "copy": 3}[mode]
    if m == 1:
        _dev(noise, "noise")
        if noise.numel() != x.numel():
            raise RuntimeError("masic_amd.quantize: noise size mismatch")
    if out is None:
        out = torch.empty_like(x)
    else:
        _dev(out, "quantize output")
        if out.shape[0] != B or out.shape[2] != H or out.shape[3] != W:
            raise RuntimeError("masic_amd.quantize: output buffer shape mismatch")
    gate_ctot = 0
    if gate is not None:
        _dev(gate, "gate")
        if gate.shape[0] != B or gate.shape[2] != H or gate.shape[3] != W:
            raise RuntimeError("masic_amd.quantize: gate shape mismatch")
        gate_ctot = gate.shape[1]
    check(lib.masic_quantize_fwd(_p(x), _p(noise), _p(gate), _p(out), B, C, H, W, out.shape[1], out_coff,
                                 gate_ctot, gate_c, m, _stream()), "quantize_fwd")
    return out


def symbols(x, medians=None):
    _dev(x, "symbols input")
    B, C, H, W = x.shape
    if medians is not None:
        _dev(medians, "medians")
        if medians.numel() != C:
            raise RuntimeError("masic_amd.symbols: medians size mismatch")
    sym = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    check(lib.masic_symbols_fwd(_p(x), _p(medians), _p(sym), B, C, H, W, _stream()), "symbols_fwd")
    return sym


def eb_param_table(matrices, biases, factors):
    """[C,58] table: matrices (3,9,9,9,3) | biases (3,3,3,3,1) | factors (3,3,3,3) per channel."""
    C = matrices[0].shape[0]
    parts = [m.reshape(C, -1) for m in matrices] + [b.reshape(C, -1) for b in biases] + [f.reshape(C, -1) for f in factors]
    table = torch.cat(parts, dim=1).contiguous()
    if table.shape[1] != _lib.EB_PARAMS_PER_CHANNEL:
        raise RuntimeError("masic_amd: EntropyBottleneck filters must be (3,3,3,3)")
    return table


def eb_table_split(g_table, shapes):
    """The gradient of eb_param_table's output split into gradients of its inputs (shapes: their shapes, [C, ...] each): one launch,
    the results are contiguous views of one buffer (masic_eb_table_split)."""
    _dev(g_table, "g_table")
    C, ncol = g_table.shape
    widths = [int(torch.Size(s[1:]).numel()) for s in shapes]
    if sum(widths) != ncol or any(s[0] != C for s in shapes):
        raise RuntimeError("masic_amd.eb_table_split: shapes do not tile the table")
    flat = torch.empty(C * ncol, dtype=torch.float32, device=g_table.device)
    w = (ctypes.c_int * len(widths))(*widths)
    check(lib.masic_eb_table_split(_p(g_table.contiguous()), _p(flat), C, w, len(widths), _stream()), "eb_table_split")
    outs, off = [], 0
    for s, wd in zip(shapes, widths):
        outs.append(flat[off:off + C * wd].view(s))
        off += C * wd
    return outs


def entropy_bottleneck(z, table, medians, training=False, noise=None, lik_bound=LIK_BOUND):
    _dev(z, "z"); _dev(table, "EB table"); _dev(medians, "medians")
    B, C, H, W = z.shape
    if tuple(table.shape) != (C, _lib.EB_PARAMS_PER_CHANNEL) or medians.numel() != C:
        raise RuntimeError("masic_amd.entropy_bottleneck: parameter table does not match channels")
    if training:
        _dev(noise, "noise")
        if noise.numel() != z.numel():
            raise RuntimeError("masic_amd.entropy_bottleneck: noise size mismatch")
    z_hat = torch.empty_like(z)
    lik = torch.empty_like(z)
    check(lib.masic_entropy_bottleneck_fwd(_p(z), _p(table), _p(medians), _p(noise), _p(z_hat), _p(lik),
                                           B, C, H, W, int(training), float(lik_bound), _stream()), "entropy_bottleneck_fwd")
    return z_hat, lik


def entropy_bottleneck_auxloss(table, quantiles, tail_mass=1e-9):
    _dev(table, "EB table"); _dev(quantiles, "quantiles")
    C = table.shape[0]
    out = torch.empty(1, dtype=torch.float32, device=table.device)
    check(lib.masic_entropy_bottleneck_auxloss(_p(table), _p(quantiles), _p(out), C, float(tail_mass), _stream()), "eb_auxloss")
    return out[0]


def gmm_likelihood(y, sigma, mu, wts, K, training=False, noise=None, weights_are_logits=False,
                   want_weights=False, scale_bound=SCALE_BOUND, lik_bound=LIK_BOUND):
    _dev(y, "y"); _dev(sigma, "sigma"); _dev(mu, "mu"); _dev(wts, "weights")
    B, M, H, W = y.shape
    for t, n in ((sigma, "sigma"), (mu, "mu"), (wts, "weights")):
        if tuple(t.shape) != (B, K * M, H, W):
            raise RuntimeError(f"masic_amd.gmm_likelihood: {n} {tuple(t.shape)} != {(B, K * M, H, W)}")
    if training == 2:            # y is the quantised latent: likelihood only (y_hat = y, nothing written)
        noise = None
    elif training:
        _dev(noise, "noise")
        if noise.numel() != y.numel():
            raise RuntimeError("masic_amd.gmm_likelihood: noise size mismatch")
    y_hat = torch.empty_like(y) if training != 2 else None
    lik = torch.empty_like(y)
    wout = torch.empty_like(wts) if (weights_are_logits and want_weights) else None
    check(lib.masic_gmm_likelihood_fwd(_p(y), _p(noise), _p(sigma), _p(mu), _p(wts), _p(y_hat), _p(lik), _p(wout),
                                       B, M, K, H, W, int(training), int(weights_are_logits),
                                       float(scale_bound), float(lik_bound), _stream()), "gmm_likelihood_fwd")
    if training == 2:
        y_hat = y
    return (y_hat, lik, wout) if want_weights else (y_hat, lik)


def softmax_k(x, K):
    _dev(x, "softmax input")
    B, KM, H, W = x.shape
    y = torch.empty_like(x)
    check(lib.masic_softmax_k_fwd(_p(x), _p(y), B, KM // K, K, H * W, _stream()), "softmax_k_fwd")
    return y


# --------------------------------------------------------------------------------------------- warp
def warp_matrix(M, src_hw, dst_hw, invert_first=False):
    _dev(M, "homography")
    B = M.shape[0]
    if tuple(M.shape) != (B, 3, 3):
        raise RuntimeError("masic_amd.warp_matrix: homography must be [B,3,3]")
    out = torch.empty_like(M)
    check(lib.masic_warp_matrix(_p(M), _p(out), B, src_hw[0], src_hw[1], dst_hw[0], dst_hw[1], int(invert_first), _stream()), "warp_matrix")
    return out


def pair_prep(img_u8, start_hw, crop_hw, homopic=256, patch_xy=(0, 0), homopatch=128):
    """One view of a dataset item (compressai/datasets/utils.py:207-285) from the decoded uint8 RGB picture [H,W,3] on the
    device -> (float32 [3,ph,pw] crop / 255, float32 [1,homopatch,homopatch] grey patch of the 256 x 256 resized crop)."""
    if not (isinstance(img_u8, torch.Tensor) and img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 3
            and img_u8.shape[2] == 3 and img_u8.is_contiguous()):
        raise RuntimeError("masic_amd.pair_prep: the picture must be a contiguous uint8 [H,W,3] CUDA (HIP) tensor")
    H, W = img_u8.shape[:2]
    ph, pw = crop_hw
    pic = torch.empty((3, ph, pw), dtype=torch.float32, device=img_u8.device)
    patch = torch.empty((1, homopatch, homopatch), dtype=torch.float32, device=img_u8.device)
    check(lib.masic_pair_prep(_p(img_u8), H, W, int(start_hw[0]), int(start_hw[1]), int(ph), int(pw), _p(pic), int(homopic),
                              int(patch_xy[0]), int(patch_xy[1]), int(homopatch), _p(patch), _stream()), "pair_prep")
    return pic, patch


def homography_from_corners(corners, delta, ori_hw, patch_hw):
    """h_matrix [B,3,3] of udh/udh/model.py:100-111 + h_adjust (newtrain_codec_real.py:49-59, :129) from the patch corners
    [B,4,2] and the predicted offsets [B,4,2]; ori_hw = picture size, patch_hw = the size the homography net saw."""
    _dev(corners, "corners"); _dev(delta, "delta")
    B = corners.shape[0]
    if tuple(corners.shape) != (B, 4, 2) or tuple(delta.shape) != (B, 4, 2):
        raise RuntimeError("masic_amd.homography_from_corners: corners and delta must be [B,4,2]")
    out = torch.empty((B, 3, 3), dtype=torch.float32, device=corners.device)
    check(lib.masic_homography_from_corners(_p(corners.contiguous()), _p(delta.contiguous()), _p(out), B, float(ori_hw[0]) / float(patch_hw[0]),
                                            float(ori_hw[1]) / float(patch_hw[1]), _stream()), "homography_from_corners")
    return out


def set_warp_align_corners(flag):
    """grid_sample convention of every warp of the library: True = kornia 0.5.0's default (the reference's pinned version, and the
    default here), False = kornia <= 0.4.1's.  Process-wide (include/masic_hip.h: masic_set_warp_align_corners); captured HIP graphs
    notice the change (masic_amd/graph.py)."""
    lib.masic_set_warp_align_corners(1 if flag else 0)


def get_warp_align_corners():
    return bool(lib.masic_get_warp_align_corners())


def warp_perspective(src, minv_norm, dsize, ones_like=None, out=None, out_coff=0):
    """src None: warp an all-ones [B,1,H,W] image whose size is given by ones_like=(B,H,W)."""
    _dev(minv_norm, "warp matrix")
    if src is not None:
        _dev(src, "warp source")
        B, C, Hs, Ws = src.shape
    else:
        B, Hs, Ws = ones_like
        C = 1
    if tuple(minv_norm.shape) != (B, 3, 3):
        raise RuntimeError("masic_amd.warp_perspective: matrix batch mismatch")
    Hd, Wd = dsize
    if out is None:
        out = torch.empty((B, C, Hd, Wd), dtype=torch.float32, device=minv_norm.device)
    else:
        _dev(out, "warp output")
    check(lib.masic_warp_perspective_fwd(_p(src), _p(minv_norm), _p(out), B, C, Hs, Ws, Hd, Wd, out.shape[1], out_coff, _stream()),
          "warp_perspective_fwd")
    return out


# --------------------------------------------------------------------------------------------- misc
def mul_inplace(x, m):
    _dev(x, "x"); _dev(m, "mask")
    if x.numel() != m.numel():
        raise RuntimeError("masic_amd.mul_inplace: size mismatch")
    check(lib.masic_mul_inplace(_p(x), _p(m), x.numel(), _stream()), "mul_inplace")
    return x


def lower_bound(x, bound):
    _dev(x, "x")
    y = torch.empty_like(x)
    check(lib.masic_lower_bound_fwd(_p(x), _p(y), float(bound), x.numel(), _stream()), "lower_bound_fwd")
    return y


def lower_bound_bwd(x, g, bound):
    _dev(x, "x"); _dev(g, "grad")
    gx = torch.empty_like(x)
    check(lib.masic_lower_bound_bwd(_p(x), _p(g), _p(gx), float(bound), x.numel(), _stream()), "lower_bound_bwd")
    return gx


def copy_view(x, out, out_coff):
    _dev(x, "x"); _dev(out, "out")
    B, C, H, W = x.shape
    check(lib.masic_copy_view(_p(x), _p(out), B, C, H * W, out.shape[1], out_coff, _stream()), "copy_view")
    return out


_ws = {}


def _workspace(device):
    key = (device.type, device.index, int(_stream().value or 0))
    if key not in _ws:
        _ws[key] = torch.empty(lib.masic_reduce_workspace_bytes() // 8, dtype=torch.float64, device=device)
    return _ws[key]


def sum_log(x):
    """float64 device scalar: sum(log(x))."""
    _dev(x, "x")
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    check(lib.masic_sum_log(_p(x), x.numel(), _p(out), _p(_workspace(x.device)), _stream()), "sum_log")
    return out[0]


def sse(a, b):
    """float64 device scalar: sum((a-b)^2)."""
    _dev(a, "a"); _dev(b, "b")
    if a.numel() != b.numel():
        raise RuntimeError("masic_amd.sse: size mismatch")
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    check(lib.masic_sse(_p(a), _p(b), a.numel(), _p(out), _p(_workspace(a.device)), _stream()), "sse")
    return out[0]


_RD_WS = {}


def _rd_ptrs(liks):
    n = len(liks)
    arr = (ctypes.c_void_p * max(n, 1))(*[l.data_ptr() for l in liks])
    cnt = (ctypes.c_size_t * max(n, 1))(*[l.numel() for l in liks])
    return arr, cnt


def rd_loss(x1_hat, x1, x2_hat, x2, liks, cb, cm):
    """The rate-distortion criterion in two launches (masic_rd_loss): returns (loss float32 0-dim, mse1, mse2, bpp, [per-tensor bpp]) --
    float64 0-dim device tensors; loss = cm (mse1 + mse2) + bpp, bpp = cb sum_k sum log liks[k].  Up to four likelihood tensors."""
    for t in (x1_hat, x1, x2_hat, x2, *liks):
        _dev(t, "tensor")
    if not (x1_hat.numel() == x1.numel() == x2_hat.numel() == x2.numel()) or len(liks) > 4:
        raise RuntimeError("masic_amd.rd_loss: pictures of one size and at most four likelihood tensors")
    dev = x1.device
    key = (dev, int(_stream().value or 0))
    ws = _RD_WS.get(key)
    if ws is None:
        ws = _RD_WS[key] = torch.empty(lib.masic_rd_loss_workspace_bytes() // 8, dtype=torch.float64, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    vals = [torch.empty((), dtype=torch.float64, device=dev) for _ in range(3 + len(liks))]
    arr, cnt = _rd_ptrs(liks)
    per = (ctypes.c_void_p * max(len(liks), 1))(*[v.data_ptr() for v in vals[3:]])
    check(lib.masic_rd_loss(_p(x1_hat), _p(x1), _p(x2_hat), _p(x2), x1.numel(), arr, cnt, len(liks), float(cb), float(cm),
                            _p(loss), _p(vals[0]), _p(vals[1]), _p(vals[2]), per, _p(ws), _stream()), "rd_loss")
    return loss, vals[0], vals[1], vals[2], vals[3:]


def rd_loss_bwd(x1_hat, x1, x2_hat, x2, liks, s_pic, s_lik, g):
    """Gradients of rd_loss's loss times the float32 device scalar g, one launch: (g_x1_hat, g_x2_hat, [g_lik])."""
    if g.dtype != torch.float32 or g.numel() != 1 or not g.is_cuda:
        raise RuntimeError("masic_amd.rd_loss_bwd: g must be a float32 device scalar")
    g1, g2 = torch.empty_like(x1_hat), torch.empty_like(x2_hat)
    gl = [torch.empty_like(l) for l in liks]
    arr, cnt = _rd_ptrs(liks)
    outs = (ctypes.c_void_p * max(len(liks), 1))(*[t.data_ptr() for t in gl])
    check(lib.masic_rd_loss_bwd(_p(x1_hat), _p(x1), _p(x2_hat), _p(x2), x1.numel(), arr, cnt, len(liks), float(s_pic), float(s_lik),
                                _p(g), _p(g1), _p(g2), outs, _stream()), "rd_loss_bwd")
    return g1, g2, gl


# --------------------------------------------------------------------------------------------- backward kernels
EW_ACT_BWD, EW_ABS_BWD, EW_SQUARE, EW_ABS, EW_AXPY, EW_RECIP_SCALE, EW_DIFF_SCALE, EW_MUL, EW_REPARAM, EW_REPARAM_BWD, EW_ADD = range(11)


def elementwise(op, a, b=None, s0=0.0, s1=0.0):
    _dev(a, "a")
    if b is not None:
        _dev(b, "b")
        if b.numel() != a.numel():
            raise RuntimeError("masic_amd.elementwise: size mismatch")
    y = torch.empty_like(a)
    check(lib.masic_elementwise(_p(a), _p(b), _p(y), a.numel(), int(op), float(s0), float(s1), _stream()), "elementwise")
    return y


def channel_sum(x, C=None, coff=0):
    _dev(x, "x")
    B, ctot = x.shape[0], x.shape[1]
    C = ctot if C is None else C
    HW = x.numel() // (B * ctot)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(_int_query("masic_channel_sum_workspace_bytes", C) // 8, dtype=torch.float64, device=x.device)
    check(lib.masic_channel_sum(_p(x), _p(out), _p(ws), B, C, HW, ctot, coff, _stream()), "channel_sum")
    return out


def slice_copy(x, coff, C):
    _dev(x, "x")
    B, ctot, H, W = x.shape
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    check(lib.masic_slice_copy(_p(x), _p(y), B, C, H * W, ctot, coff, _stream()), "slice_copy")
    return y


def gate_bwd(g, x, gate, gate_c):
    _dev(g, "g"); _dev(x, "x"); _dev(gate, "gate")
    B, C, H, W = x.shape
    gx = torch.empty_like(x)
    ggate = zeros(gate.shape, gate.dtype, gate.device)
    check(lib.masic_gate_bwd(_p(g), _p(x), _p(gate), _p(gx), _p(ggate), B, C, H * W, gate.shape[1], gate_c, _stream()), "gate_bwd")
    return gx, ggate


def softmax_k_bwd(g, y, K):
    _dev(g, "g"); _dev(y, "y")
    B, KM, H, W = y.shape
    gx = torch.empty_like(y)
    check(lib.masic_softmax_k_bwd(_p(g), _p(y), _p(gx), B, KM // K, K, H * W, _stream()), "softmax_k_bwd")
    return gx


def gdn_bwd_pre(x, nrm, g, inverse):
    s = torch.empty_like(x)
    t = torch.empty_like(x)
    check(lib.masic_gdn_bwd_pre(_p(_dev(x)), _p(_dev(nrm)), _p(_dev(g)), _p(s), _p(t), x.numel(), int(inverse), _stream()), "gdn_bwd_pre")
    return s, t


def gdn_bwd_post(x, s, u):
    dx = torch.empty_like(x)
    check(lib.masic_gdn_bwd_post(_p(_dev(x)), _p(_dev(s)), _p(_dev(u)), _p(dx), x.numel(), _stream()), "gdn_bwd_post")
    return dx


_GDN_BWD_WS = {}


_GDN_BWD_SMALL_WS = {}


def gdn_bwd_small(x, g, beta, gamma, inverse=False, beta_min=1e-6):
    """(dx, d beta, d gamma) of GDN / inverse GDN with C <= 4 channels in one pass (float32; masic_gdn_bwd_small)."""
    _dev(x, "x"); _dev(g, "g")
    B, C, H, W = x.shape
    if g.shape != x.shape or not (x.is_contiguous() and g.is_contiguous()):
        raise RuntimeError("masic_amd.gdn_bwd_small: x and g must be contiguous and of one shape")
    key = (x.device, int(_stream().value or 0))
    ws = _GDN_BWD_SMALL_WS.get(key)
    if ws is None:
        ws = _GDN_BWD_SMALL_WS[key] = torch.empty(lib.masic_gdn_bwd_small_workspace_bytes(), dtype=torch.uint8, device=x.device)
    gx = torch.empty_like(x)
    g_beta = torch.empty(C, dtype=torch.float32, device=x.device)
    g_gamma = torch.empty(C, C, dtype=torch.float32, device=x.device)
    check(lib.masic_gdn_bwd_small(_p(x), _p(g), _p(_dev(beta.contiguous())), _p(_dev(gamma.contiguous())), _p(gx), _p(g_beta),
                                  _p(g_gamma), _p(ws), B, C, H, W, int(inverse), float(beta_min), _stream()), "gdn_bwd_small")
    return gx, g_beta, g_gamma


def gdn_bwd_fused(x, g, beta, gamma, inverse=False, beta_min=1e-6):
    """(dx, d beta, d gamma) of GDN / inverse GDN at C = 128 in one kernel (bf16 operands, float32 accumulate)."""
    _dev(x, "x"); _dev(g, "g")
    B, C, H, W = x.shape
    if g.shape != x.shape or not (x.is_contiguous() and g.is_contiguous()):
        raise RuntimeError("masic_amd.gdn_bwd_fused: x and g must be contiguous and of one shape")
    key = (x.device, int(_stream().value or 0))        # per stream: two streams' kernels must not share the partial sums
    ws = _GDN_BWD_WS.get(key)
    if ws is None:
        ws = _GDN_BWD_WS[key] = torch.empty(lib.masic_gdn_bwd_fused_workspace_bytes(), dtype=torch.uint8, device=x.device)
    gx = torch.empty_like(x)
    g_beta = torch.empty(C, dtype=torch.float32, device=x.device)
    g_gamma = torch.empty(C, C, dtype=torch.float32, device=x.device)
    check(lib.masic_gdn_bwd_fused(_p(x), _p(g), _p(_dev(beta.contiguous())), _p(_dev(gamma.contiguous())), _p(gx), _p(g_beta),
                                  _p(g_gamma), _p(ws), B, C, H, W, int(inverse), float(beta_min), _stream()), "gdn_bwd_fused")
    return gx, g_beta, g_gamma


def gdn_bwd_fused_ex(x, g, shape, beta, gamma, inverse=False, beta_min=1e-6, want_nchw=True, want_f16k=False, want_sum=True, want_b16=False):
    """gdn_bwd_fused with F16K operands (masic_gdn_bwd_fused_ex2): x and g are float32 NCHW tensors or F16K int16 buffers of
    shape = (B, 128, H, W); returns (dx float32 NCHW | None, dx F16K | None, channel sums of dx [128] | None, d beta, d gamma).
    want_b16: the NCHW result is bf16 (torch.bfloat16, returned in the first slot) -- the operand of conv2d_wgrad_b16."""
    B, C, H, W = shape
    for t, name in ((x, "x"), (g, "g")):
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError(f"masic_amd.gdn_bwd_fused_ex: {name} must be a contiguous device tensor")
        if t.dtype == torch.int16:
            if t.numel() != B * C * H * W:
                raise RuntimeError(f"masic_amd.gdn_bwd_fused_ex: F16K buffer {name} does not hold {shape}")
        elif t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"masic_amd.gdn_bwd_fused_ex: {name} must be float32 {shape} or an F16K int16 buffer")
    if not (want_nchw or want_f16k or want_b16):
        raise RuntimeError("masic_amd.gdn_bwd_fused_ex: no output requested")
    dev = x.device
    key = (dev, int(_stream().value or 0))
    ws = _GDN_BWD_WS.get(key)
    if ws is None:
        ws = _GDN_BWD_WS[key] = torch.empty(lib.masic_gdn_bwd_fused_workspace_bytes(), dtype=torch.uint8, device=dev)
    gx = torch.empty(shape, dtype=torch.float32, device=dev) if (want_nchw and not want_b16) else None
    gxb = torch.empty(shape, dtype=torch.bfloat16, device=dev) if want_b16 else None
    gx16 = torch.empty(B * C * H * W, dtype=torch.int16, device=dev) if want_f16k else None
    g_sum = torch.empty(C, dtype=torch.float32, device=dev) if want_sum else None
    g_beta = torch.empty(C, dtype=torch.float32, device=dev)
    g_gamma = torch.empty(C, C, dtype=torch.float32, device=dev)
    x16, g16 = x.dtype == torch.int16, g.dtype == torch.int16
    check(lib.masic_gdn_bwd_fused_ex2(_p(None if x16 else x), _p(x if x16 else None), _p(None if g16 else g), _p(g if g16 else None),
                                      _p(_dev(beta.contiguous())), _p(_dev(gamma.contiguous())), _p(gx), _p(gx16), _p(gxb), _p(g_sum), _p(g_beta), _p(g_gamma),
                                      _p(ws), B, C, H, W, int(inverse), float(beta_min), _stream()), "gdn_bwd_fused_ex")
    return (gxb if want_b16 else gx), gx16, g_sum, g_beta, g_gamma


def conv2d_wgrad_b16_supported(desc):
    """True if conv2d_wgrad takes bf16 NCHW operands for this layer (the 5x5 stride-2 layers of the bf16 mode)."""
    return bool(_desc_query("masic_conv2d_wgrad_bf16in_supported", desc))


def conv2d_wgrad(x, dy, desc, weight_shape):
    """dW of the layer `desc` describes; x, dy float32 NCHW -- or BOTH torch.bfloat16 NCHW where conv2d_wgrad_b16_supported(desc)."""
    b16 = isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16
    if b16:
        if not (x.is_cuda and x.is_contiguous() and isinstance(dy, torch.Tensor) and dy.is_cuda and dy.is_contiguous()):
            raise RuntimeError("masic_amd.conv2d_wgrad: contiguous device tensors only")
    else:
        _dev(x, "x"); _dev(dy, "dy")
    if b16 != (dy.dtype == torch.bfloat16) or (b16 and not conv2d_wgrad_b16_supported(desc)):
        raise RuntimeError("masic_amd.conv2d_wgrad: bf16 operands need both tensors in bf16 and a layer conv2d_wgrad_b16_supported")
    if tuple(dy.shape) != (desc.B, desc.Cout, desc.Ho, desc.Wo):
        raise RuntimeError(f"masic_amd.conv2d_wgrad: dy {tuple(dy.shape)} does not match the descriptor")
    if tuple(x.shape) != (desc.B, desc.in_ctot, desc.Hi, desc.Wi):
        raise RuntimeError(f"masic_amd.conv2d_wgrad: x {tuple(x.shape)} does not match the descriptor")
    dw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    nbytes = _desc_query("masic_conv2d_wgrad_workspace_bytes", desc)
    if desc.Cin * desc.Cout * desc.KH * desc.KW != dw.numel() or nbytes < dw.numel() * 4:
        raise RuntimeError("masic_amd.conv2d_wgrad: weight shape does not match the descriptor")
    ws = _clean_workspace(x.device, nbytes // 4)       # (>= the weight: the few-channel 5x5 kernel keeps per-workgroup partials in it)
    try:
        fn = lib.masic_conv2d_wgrad_bf16in if b16 else lib.masic_conv2d_wgrad_ws
        check(fn(_p(x), _p(dy), _p(dw), _p(ws), ctypes.byref(desc), 1, _stream()), "conv2d_wgrad")
    except Exception:
        _drop_workspace(x.device, nbytes // 4)         # an error part-way may have left it dirty
        raise
    return dw


_CLEAN_WS = {}


def _clean_workspace(device, numel):
    """Persistent float32 workspace of the weight-gradient kernels, zeroed ONCE: the kernels accumulate into it with float atomics and
    their last pass puts the zeros back (masic_conv2d_wgrad_ws), so a step's ~80 weight gradients need no fill launch each.  One buffer
    per (device, stream, size): launches on one stream are ordered, two streams never share a buffer."""
    key = (device, int(_stream().value or 0), int(numel))
    ws = _CLEAN_WS.get(key)
    if ws is None:
        ws = _CLEAN_WS[key] = zeros(numel, torch.float32, device)
    return ws


def _drop_workspace(device, numel):
    _CLEAN_WS.pop((device, int(_stream().value or 0), int(numel)), None)


def gmm_likelihood_bwd(y_hat, sigma, mu, wts, g_lik, g_yhat, K, weights_are_logits, scale_bound=SCALE_BOUND, lik_bound=LIK_BOUND):
    for t in (y_hat, sigma, mu, wts, g_lik):
        _dev(t)
    B, M, H, W = y_hat.shape
    g_y = torch.empty_like(y_hat)
    g_s, g_m, g_w = torch.empty_like(sigma), torch.empty_like(mu), torch.empty_like(wts)
    check(lib.masic_gmm_likelihood_bwd(_p(y_hat), _p(sigma), _p(mu), _p(wts), _p(g_lik), _p(g_yhat), _p(g_y), _p(g_s), _p(g_m), _p(g_w),
                                       B, M, K, H, W, int(weights_are_logits), float(scale_bound), float(lik_bound), _stream()),
          "gmm_likelihood_bwd")
    return g_y, g_s, g_m, g_w


def entropy_bottleneck_bwd(z_hat, table, g_lik, g_zhat, lik_bound=LIK_BOUND):
    _dev(z_hat); _dev(table); _dev(g_lik)
    B, C, H, W = z_hat.shape
    g_z = torch.empty_like(z_hat)
    g_t = torch.empty_like(table)
    check(lib.masic_entropy_bottleneck_bwd(_p(z_hat), _p(table), _p(g_lik), _p(g_zhat), _p(g_z), _p(g_t), B, C, H, W,
                                           float(lik_bound), _stream()), "entropy_bottleneck_bwd")
    return g_z, g_t


def entropy_bottleneck_aux_step(tables, quantiles, tail_masses):
    """Loss and d loss / d quantiles of sum_e EntropyBottleneck_e.loss() in two launches (masic_entropy_bottleneck_aux_step):
    returns (total loss, float32 0-dim; [gradient like quantiles[e]])."""
    n = len(tables)
    if not 1 <= n <= 4 or len(quantiles) != n or len(tail_masses) != n:
        raise RuntimeError("masic_amd.entropy_bottleneck_aux_step: 1..4 bottlenecks")
    qs = [_dev(q.detach().contiguous(), "quantiles") for q in quantiles]
    for t, q in zip(tables, qs):
        _dev(t, "EB table")
        if q.numel() != 3 * t.shape[0]:
            raise RuntimeError("masic_amd.entropy_bottleneck_aux_step: quantiles do not match the table")
    dev = tables[0].device
    gq = [torch.empty_like(q) for q in qs]
    cmax = max(t.shape[0] for t in tables)
    ws = torch.empty(n * 3 * cmax, dtype=torch.float32, device=dev)
    loss = torch.empty(1 + n, dtype=torch.float32, device=dev)
    vp = ctypes.c_void_p * n
    check(lib.masic_entropy_bottleneck_aux_step(vp(*[t.data_ptr() for t in tables]), vp(*[q.data_ptr() for q in qs]), vp(*[g.data_ptr() for g in gq]),
                                                (ctypes.c_int * n)(*[t.shape[0] for t in tables]), (ctypes.c_double * n)(*[float(m) for m in tail_masses]),
                                                n, _p(loss), _p(ws), ws.numel(), _stream()), "entropy_bottleneck_aux_step")
    return loss[0], gq


def entropy_bottleneck_auxloss_bwd(table, quantiles, gout, tail_mass=1e-9):
    _dev(table); _dev(quantiles)
    g_q = torch.empty_like(quantiles)
    check(lib.masic_entropy_bottleneck_auxloss_bwd(_p(table), _p(quantiles), _p(g_q), table.shape[0], float(tail_mass), float(gout), _stream()),
          "entropy_bottleneck_auxloss_bwd")
    return g_q


_WARP_BWD_GATHER = _os.environ.get("MASIC_WARP_BWD_GATHER", "1") != "0"      # 0: the scatter form (float atomics) always (A/B timing)


def warp_perspective_bwd(g_dst, minv_norm, src_shape, want_flag=False):
    """d loss / d src of warp_perspective.  Default: the gather form (masic_warp_perspective_bwd_gather: no atomics, reproducible, no zero
    fill), which falls back to the scatter form on the device when the homography is outside its bounds; want_flag: also return the int32
    device flag that says it did."""
    _dev(g_dst); _dev(minv_norm)
    B, C, Hs, Ws = src_shape
    Hd, Wd = g_dst.shape[-2:]
    if _WARP_BWD_GATHER and min(Hs, Ws, Hd, Wd) > 1:
        g_src = torch.empty(src_shape, dtype=torch.float32, device=g_dst.device)
        flag = torch.empty(1, dtype=torch.int32, device=g_dst.device).fill_(0)
        check(lib.masic_warp_perspective_bwd_gather(_p(g_dst), _p(minv_norm), _p(g_src), _p(flag), B, C, Hs, Ws, Hd, Wd, _stream()),
              "warp_perspective_bwd_gather")
        return (g_src, flag) if want_flag else g_src
    g_src = zeros(src_shape, torch.float32, g_dst.device)
    check(lib.masic_warp_perspective_bwd(_p(g_dst), _p(minv_norm), _p(g_src), B, C, Hs, Ws, Hd, Wd, _stream()), "warp_perspective_bwd")
    return (g_src, None) if want_flag else g_src


# --------------------------------------------------------------------------------------------- bf16 1x1 GEMM stacks
def nchw_to_f16k(x, C=None, coff=0, in_op=INOP_NONE):
    """float32 NCHW (channel view) -> F16K bf16 [B][ceil16(C)/16][HW][16] (include/masic_hip.h); in_op: |x| / round(x) on the way."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    C = ctot if C is None else C
    y = torch.empty(_int_query("masic_f16k_bytes", B, C, H * W) // 2, dtype=torch.int16, device=x.device)
    check(lib.masic_nchw_to_f16k_op(_p(x), _p(y), B, C, H * W, ctot, coff, int(in_op), _stream()), "nchw_to_f16k")
    return y


def pack_gemm1x1_weight(weight, Cin, Cout, transposed):
    _dev(weight, "weight")
    wp = torch.empty(_int_query("masic_gemm1x1_packed_bytes", Cin, Cout) // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_gemm1x1_pack_weight(_p(weight), _p(wp), Cin, Cout, int(transposed), _stream()), "gemm1x1_pack_weight")
    return wp


def gemm1x1_bf16(x_f16k, wp, bias, B, Cin, Cout, H, W, act, out_nchw=None, out_coff=0, want_nchw=False):
    """y = act(W x + b) on F16K activations; returns F16K (int16 buffer) or, with want_nchw / out_nchw, float32 NCHW."""
    HW = H * W
    y16 = y32 = None
    out_ctot = 0
    if want_nchw or out_nchw is not None:
        y32 = out_nchw if out_nchw is not None else torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_f16k.device)
        out_ctot = y32.shape[1]
    else:
        y16 = torch.empty(_int_query("masic_f16k_bytes", B, Cout, HW) // 2, dtype=torch.int16, device=x_f16k.device)
    check(lib.masic_gemm1x1_bf16_fwd(_p(x_f16k), _p(wp), _p(bias), _p(y16), _p(y32), B, Cin, Cout, HW, out_ctot, out_coff, int(act), _stream()),
          "gemm1x1_bf16_fwd")
    return y32 if y32 is not None else y16


# --------------------------------------------------------------------------------------------- F16K convolutions
def f16k_to_nchw(y16, B, C, H, W):
    """F16K int16 buffer -> float32 NCHW (torch ops; used by tests and by consumers outside the bf16 chain)."""
    c16 = (C + 15) // 16
    t = y16.view(torch.bfloat16).view(B, c16, H * W, 16).permute(0, 1, 3, 2).reshape(B, c16 * 16, H, W)
    return t[:, :C].float().contiguous()


def conv_f16k_supported(desc):
    return bool(_desc_query("masic_conv_f16k_supported", desc))


def pack_conv_f16k_weight(weight, desc, persistent=False):
    """The weight as conv_f16k's slab stream.  persistent: `weight` is (a detached view of) a long-lived parameter -- the pack is
    registered with StreamPacks below and refreshed together with all the other registered packs in ONE launch when the parameters
    change (a training step), instead of one launch per layer and orientation."""
    _dev(weight, "weight")
    if persistent and _PACK_MULTI:
        return _stream_packs(weight.device).get(weight, desc)
    nbytes = _desc_query("masic_conv_f16k_packed_bytes", desc)
    if nbytes == 0:
        check(-1, "conv_f16k_packed_bytes")
    packed = torch.empty(nbytes // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_conv_f16k_pack_weight(_p(weight), _p(packed), ctypes.byref(desc), _stream()), "conv_f16k_pack_weight")
    return packed


_PACK_MULTI = _os.environ.get("MASIC_PACK_MULTI", "1") != "0"      # 0: one pack launch per weight and orientation (A/B timing)
_DESC_KEY = ("B", "Cin", "Hi", "Wi", "in_ctot", "Cout", "KH", "KW", "stride", "pad", "transposed", "masked", "prec")


class StreamPacks:
    """Registry of the conv_f16k weight packs of long-lived parameters on one device.  An entry = (parameter storage, layer geometry)
    -> a persistent packed buffer + the parameter version it was packed from.  `get` returns the buffer; when it finds the entry stale
    and most other entries stale too (the first pack request after an optimizer step), every registered pack is refreshed by ONE launch
    (masic_conv_f16k_pack_jobs_run on a job table kept in device memory); a lone stale entry (a masked layer re-zeroing its taps, a
    layer seen for the first time, any request under no_grad) is packed by itself, on the requesting stream, as without the registry."""

    def __init__(self, device):
        self.device = device
        self.entries = {}          # key -> [weight, desc, out, version, job bytes, nb, tick of the last request]
        self.tick = 0              # counts batched refreshes; entries not requested for four of them (their model is gone) are dropped
        self.table = None          # device uint8: the jobs of all entries, in dict order
        self.max_nb = 1
        self.job_bytes = lib.masic_conv_f16k_pack_job_bytes()

    def _register(self, key, weight, desc):
        nbytes = _desc_query("masic_conv_f16k_packed_bytes", desc)
        if nbytes == 0:
            check(-1, "conv_f16k_packed_bytes")
        out = torch.empty(nbytes // 2, dtype=torch.int16, device=weight.device)
        d = ConvDesc()
        ctypes.pointer(d)[0] = desc
        job = ctypes.create_string_buffer(self.job_bytes)
        nb = lib.masic_conv_f16k_pack_job(_p(weight), _p(out), ctypes.byref(d), job)
        if nb < 1:
            check(nb if nb < 0 else -1, "conv_f16k_pack_job")
        if len(self.entries) >= 1024:      # many models and no training step in between (a test session): forget the oldest half
            for k in sorted(self.entries, key=lambda k: self.entries[k][6])[:512]:
                del self.entries[k]
        e = self.entries[key] = [weight, d, out, None, job.raw, nb, self.tick]
        self.table = None
        return e

    def get(self, weight, desc):
        key = (weight.data_ptr(),) + tuple(getattr(desc, f) for f in _DESC_KEY)
        e = self.entries.get(key)
        if e is None:
            e = self._register(key, weight, desc)
        e[6] = self.tick
        if e[3] == _stamp(weight):
            return e[2]
        # "most entries are stale" = the parameters were stepped; judged on the entries in use (requested since the refresh before last):
        # the leftovers of a model that is gone must not keep the majority fresh
        active = [x for x in self.entries.values() if x[6] >= self.tick - 1]
        stale = sum(1 for x in active if x[3] != _stamp(x[0]))
        # (batched only inside `with batched_packs():` -- the training steps of masic_amd/train.py, one stream: a refresh rewrites every
        # registered buffer, and the multi-stream eval forward must not have buffers rewritten or first filled by another stream than the
        # one about to read them)
        capturing = torch.cuda.is_current_stream_capturing()      # (no host -> device copy of a new job table inside a graph capture)
        if stale >= 4 and 2 * stale >= len(active) and _BATCHED_PACKS[0] > 0 and not (capturing and self.table is None):
            self.tick += 1
            dead = [] if capturing else [k for k, x in self.entries.items() if x[6] < self.tick - 4]
            for k in dead:                 # (the registry holds the only reference that keeps such a weight and its pack alive)
                del self.entries[k]
            if dead:
                self.table = None
            if self.table is None:
                raw = b"".join(x[4] for x in self.entries.values())
                self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
                self.max_nb = max(x[5] for x in self.entries.values())
            check(lib.masic_conv_f16k_pack_jobs_run(_p(self.table), len(self.entries), self.max_nb, _stream()), "conv_f16k_pack_jobs_run")
            for x in self.entries.values():
                x[3] = _stamp(x[0])
        else:
            check(lib.masic_conv_f16k_pack_weight(_p(e[0]), _p(e[2]), ctypes.byref(e[1]), _stream()), "conv_f16k_pack_weight")
            e[3] = _stamp(weight)
        return e[2]


_STREAM_PACKS = {}
_BATCHED_PACKS = [0]


class batched_packs:
    """Context of a single-stream training step: stale weight packs are refreshed together (StreamPacks)."""

    def __enter__(self):
        _BATCHED_PACKS[0] += 1

    def __exit__(self, *exc):
        _BATCHED_PACKS[0] -= 1
        return False


def _stream_packs(device):
    sp = _STREAM_PACKS.get(device)
    if sp is None:
        sp = _STREAM_PACKS[device] = StreamPacks(device)
    return sp


def pack_gdn_f16k(beta, gamma, beta_min=1e-6):
    """A 128-channel GDN's stored parameters in the fragment order of conv_f16k's fused epilogue."""
    _dev(beta, "beta"); _dev(gamma, "gamma")
    packed = torch.empty(_int_query("masic_gdn_f16k_packed_bytes") // 2, dtype=torch.int16, device=beta.device)
    check(lib.masic_gdn_pack_f16k(_p(beta), _p(gamma), _p(packed), beta.numel(), float(beta_min), _stream()), "gdn_pack_f16k")
    return packed


def conv2d_f16k(x16, packed, bias, desc, out_nchw=None, want_nchw=False, gate=None, gdn=None, out16=None):
    """Conv on an F16K input [B][in_ctot/16][Hi*Wi][16]; returns float32 NCHW (want_nchw / out_nchw) or an F16K buffer of
    desc.out_ctot channels (a fresh one holds exactly ceil16(Cout) channels; `out16`: an existing buffer of desc.out_ctot channels,
    written at desc.out_coff -- a slice of a concat buffer, optionally gated)."""
    if x16.dtype != torch.int16 or not x16.is_cuda:
        raise RuntimeError("masic_amd.conv2d_f16k: input must be an F16K int16 CUDA buffer")
    if x16.numel() != desc.B * desc.in_ctot * desc.Hi * desc.Wi:
        raise RuntimeError(f"masic_amd.conv2d_f16k: input buffer of {x16.numel()} elements does not match descriptor "
                           f"[{desc.B},{desc.in_ctot},{desc.Hi},{desc.Wi}]")
    if bias is not None:
        _dev(bias, "bias")
    y32 = y16 = None
    if want_nchw or out_nchw is not None:
        y32 = out_nchw if out_nchw is not None else torch.empty((desc.B, desc.out_ctot, desc.Ho, desc.Wo), dtype=torch.float32, device=x16.device)
        if tuple(y32.shape) != (desc.B, desc.out_ctot, desc.Ho, desc.Wo):
            raise RuntimeError(f"masic_amd.conv2d_f16k: output buffer {tuple(y32.shape)} does not match descriptor")
    elif out16 is not None:
        if out16.dtype != torch.int16 or out16.numel() != desc.B * desc.out_ctot * desc.Ho * desc.Wo:
            raise RuntimeError("masic_amd.conv2d_f16k: out16 does not match the descriptor's output view")
        y16 = out16
    else:
        y16 = torch.empty(desc.B * desc.out_ctot * desc.Ho * desc.Wo, dtype=torch.int16, device=x16.device)
    if gate is not None and tuple(gate.shape) != (desc.B, desc.gate_ctot, desc.Ho, desc.Wo):
        raise RuntimeError("masic_amd.conv2d_f16k: gate does not match descriptor")
    timed = None
    if _timer is not None:
        buf = ctypes.create_string_buffer(96)
        lib.masic_conv_f16k_kernel_name(ctypes.byref(desc), int(gdn is not None), buf, 96)
        variant = buf.value.decode()
        if _timer.only is None or variant == _timer.only:
            timed = (variant, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timed[1].record()
    gp, ginv = (None, 0) if gdn is None else (gdn[0], _gdn_flags(gdn[1]))       # gdn = (pack_gdn_f16k(...), inverse)
    check(lib.masic_conv_f16k_gdn_fwd(_p(x16), _p(packed), _p(bias), _p(gate), _p(gp), ginv, _p(y32), _p(y16), ctypes.byref(desc), _stream()),
          "conv_f16k_fwd")
    if timed is not None:
        timed[2].record()
        flops, nbytes = conv_algorithmic_work(desc)
        _timer.records.append((timed[0], 1, flops, nbytes, timed[1], timed[2]))
    return y32 if y32 is not None else y16


def pack_conv_a_weight(weight):
    _dev(weight, "weight")
    if tuple(weight.shape) != (128, 3, 5, 5):
        raise RuntimeError("masic_amd.pack_conv_a_weight: the first-layer kernel is built for Conv2d(3, 128, 5, stride 2)")
    packed = torch.empty(_int_query("masic_conv_a_packed_bytes") // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_conv_a_pack_weight(_p(weight), _p(packed), _stream()), "conv_a_pack_weight")
    return packed


def conv_a_gdn_f16k(x, packed, bias, gdn, in_coff=0):
    """GDN(Conv2d(3 -> 128, k5, s2, p2)(x[:, in_coff:in_coff+3]) + bias) -> (F16K buffer, Ho, Wo); gdn = (pack_gdn_f16k(..), inverse)."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B * 128 * Ho * Wo, dtype=torch.int16, device=x.device)
    check(lib.masic_conv_a_gdn_fwd(_p(x), _p(packed), _p(bias), _p(gdn[0]), _gdn_flags(gdn[1]), _p(y), B, H, W, ctot, in_coff, _stream()), "conv_a_gdn_fwd")
    return y, Ho, Wo


def deconv_s2_as_conv_weight(weight, bias):
    """ConvTranspose2d(Cin -> C, k5, s2, p2, output_padding 1) as the equivalent Conv2d(Cin -> 4C, k3, s1, p1) followed by a
    2x2 depth-to-space: row (4c + phase) = W_t[:, c, ph + 2(2-u), pw + 2(2-v)] with phase = 2 ph + pw, zero where that index
    exceeds 4; rows padded to 32.  Returns (weight [32, Cin, 3, 3], bias [32])."""
    Cin, C, KH, KW = weight.shape
    if (KH, KW) != (5, 5) or 4 * C > 32:
        raise RuntimeError("masic_amd.deconv_s2_as_conv_weight: needs a 5x5 kernel and at most 8 output channels")
    w = torch.zeros((32, Cin, 3, 3), dtype=torch.float32, device=weight.device)
    b = torch.zeros(32, dtype=torch.float32, device=weight.device)
    for ph in range(2):
        for pw in range(2):
            rows = slice(ph * 2 + pw, 4 * C, 4)
            if bias is not None:
                b[rows] = bias
            for u in range(3):
                for v in range(3):
                    kh, kw = ph + 2 * (2 - u), pw + 2 * (2 - v)
                    if kh <= 4 and kw <= 4:
                        w[rows, :, u, v] = weight[:, :, kh, kw].t()
    return w, b


def conv2d_f16k_d2s(x16, packed, bias32, desc, C, out=None, out_coff=0):
    """The depth-to-space form of a stride-2 transposed convolution (see deconv_s2_as_conv_weight) on an F16K input;
    returns float32 NCHW [B, C, 2Hi, 2Wi] (or writes channels out_coff.. of `out`)."""
    if out is None:
        out = torch.empty((desc.B, C, 2 * desc.Hi, 2 * desc.Wi), dtype=torch.float32, device=x16.device)
    check(lib.masic_conv_f16k_d2s_fwd(_p(x16), _p(packed), _p(bias32), _p(out), C, out.shape[1], out_coff, ctypes.byref(desc), _stream()),
          "conv_f16k_d2s_fwd")
    return out


def conv5s1_pair(xa, xb, packed, bias, gdn_in=None, gdn_out=None, beta_min=1e-6):
    """Conv2d / ConvTranspose2d(6 -> 3, k5, s1, p2) on [xa | xb] (3 + 3 channels, no concat buffer) with an optional 3-channel
    (I)GDN on xa while staging (gdn_in = (beta, gamma, inverse)) and / or on the result (gdn_out)."""
    _dev(xa, "xa"); _dev(xb, "xb")
    if xa.shape != xb.shape or xa.shape[1] != 3:
        raise RuntimeError("masic_amd.conv5s1_pair: both sources must be [B, 3, H, W]")
    B, _, H, W = xa.shape
    y = torch.empty((B, 3, H, W), dtype=torch.float32, device=xa.device)
    gi = (None, None, 0) if gdn_in is None else gdn_in
    go = (None, None, 0) if gdn_out is None else gdn_out
    check(lib.masic_conv5s1_pair_fwd(_p(xa), _p(xb), _p(packed), _p(bias), _p(gi[0]), _p(gi[1]), int(gi[2]), _p(go[0]), _p(go[1]), int(go[2]),
                                     float(beta_min), _p(y), B, H, W, _stream()), "conv5s1_pair_fwd")
    return y


def pack_gemm_f16k_weights(jobs):
    """jobs: [(weight, Cin, Cout, transposed), ...] (<= 18) -> the packs pack_gemm_f16k_weight would make, in ONE launch."""
    n = len(jobs)
    outs = [torch.empty(_int_query("masic_gemm_f16k_packed_bytes", ci, co) // 2, dtype=torch.int16, device=w.device) for w, ci, co, _ in jobs]
    for w, _, _, _ in jobs:
        _dev(w, "weight")
    W = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _, _, _ in jobs])
    P = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    I = lambda k: (ctypes.c_int * n)(*[int(j[k]) for j in jobs])
    check(lib.masic_gemm_f16k_pack_weights(W, P, I(1), I(2), I(3), n, _stream()), "gemm_f16k_pack_weights")
    return outs


def pack_gemm_f16k_weight(weight, Cin, Cout, transposed):
    _dev(weight, "weight")
    wp = torch.empty(_int_query("masic_gemm_f16k_packed_bytes", Cin, Cout) // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_gemm_f16k_pack_weight(_p(weight), _p(wp), Cin, Cout, int(transposed), _stream()), "gemm_f16k_pack_weight")
    return wp


def gemm_f16k(x_f16k, wp, bias, B, Cin, Cout, H, W, act, out_nchw=None, out_coff=0, want_nchw=False):
    """DMA-staged form of gemm1x1_bf16 (same arguments and results); needs Cin % 16 == 0 and Cout % 32 == 0."""
    HW = H * W
    y16 = y32 = None
    if want_nchw or out_nchw is not None:
        y32 = out_nchw if out_nchw is not None else torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_f16k.device)
        out_ctot = y32.shape[1]
    else:
        out_ctot = (Cout + 15) // 16 * 16
        y16 = torch.empty(B * out_ctot * HW, dtype=torch.int16, device=x_f16k.device)
    check(lib.masic_gemm_f16k_fwd(_p(x_f16k), _p(wp), _p(bias), _p(y16), _p(y32), B, Cin, Cout, HW, out_ctot, out_coff, int(act), _stream()),
          "gemm_f16k_fwd")
    return y32 if y32 is not None else y16


def gemm_f16k_group(layers, B, H, W):
    """Up to three independent 1x1 layers over the same B x H x W pixels in ONE launch (masic_gemm_f16k_group_fwd): layer i of the
    three entropy-parameter stacks of a GMM head.  layers: dicts with x, wp, bias, Cin, Cout, act and out = "f16k" | "nchw" | "f8k";
    fp8-operand layers (all or none) add ws (dequantisation scales) and, for out = "f8k", out_scale.  Returns the outputs in order;
    each equals what gemm_f16k / gemm_f8k return for that layer, bit for bit."""
    HW = H * W
    arr = (_lib.GemmGroup * len(layers))()
    outs = []
    for g, L in zip(arr, layers):
        x, Cout, out = L["x"], L["Cout"], L.get("out", "f16k")
        y16 = y8 = y32 = None
        if out == "nchw":
            y32 = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
            out_ctot = Cout
        elif out == "f8k":
            out_ctot = (Cout + 31) // 32 * 32
            y8 = torch.empty(B * out_ctot * HW, dtype=torch.uint8, device=x.device)
        else:
            out_ctot = (Cout + 31) // 32 * 32 if L.get("ws") is not None else (Cout + 15) // 16 * 16
            y16 = torch.empty(B * out_ctot * HW, dtype=torch.int16, device=x.device)
        g.x, g.w_packed, g.wscale, g.bias = _p(x), _p(L["wp"]), _p(L.get("ws")), _p(L["bias"])
        g.y_f16k, g.y_f8k, g.y_nchw = _p(y16), _p(y8), _p(y32)
        g.out_inv_scale = 1.0 / float(L["out_scale"]) if out == "f8k" else 0.0
        g.Cin, g.Cout, g.out_ctot, g.out_coff, g.act = L["Cin"], Cout, out_ctot, 0, int(L["act"])
        outs.append(y32 if y32 is not None else (y8 if y8 is not None else y16))
    check(lib.masic_gemm_f16k_group_fwd(arr, len(layers), B, HW, _stream()), "gemm_f16k_group_fwd")
    return outs


def pack_skinny_ctx_weight(weight):
    """Fragment pack of a 5x5 type-A masked convolution's weight [Cout, Cin, 5, 5] (masked taps zero) for skinny_group(ctx=True)."""
    _dev(weight, "weight")
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (5, 5):
        raise RuntimeError("masic_amd.pack_skinny_ctx_weight: 5x5 kernels only")
    out = torch.empty(_int_query("masic_skinny_ctx_packed_bytes", Cin, Cout) // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_skinny_ctx_pack_weight(_p(weight.contiguous()), _p(out), Cin, Cout, _stream()), "skinny_ctx_pack_weight")
    return out


def skinny_group(layers, pix, step, list_stride, npix, h, w, ctx=False, gate=None, gate_c=0):
    """masic_skinny_group_fwd: up to three 1x1 layers (or one masked 5x5 convolution, ctx=True) on the pixels pix[step * list_stride + i],
    i < npix, of one image.  layers: dicts with x (F16K), wp, bias, Cin, Cout, act, out_ctot, out_coff and the FULL-SIZE output buffer
    y16 (F16K) or y32 (float32 [out_ctot, h, w]); pix / step: int32 device tensors (step may be None)."""
    arr = (_lib.GemmGroup * len(layers))()
    for g, L in zip(arr, layers):
        g.x, g.w_packed, g.wscale, g.bias = _p(L["x"]), _p(L["wp"]), None, _p(L["bias"])
        g.y_f16k, g.y_f8k, g.y_nchw = _p(L.get("y16")), None, _p(L.get("y32"))
        g.out_inv_scale = 0.0
        g.Cin, g.Cout, g.out_ctot, g.out_coff, g.act = L["Cin"], L["Cout"], L["out_ctot"], L.get("out_coff", 0), int(L["act"])
    check(lib.masic_skinny_group_fwd(arr, len(layers), 1 if ctx else 0, _p(pix), _p(step), int(list_stride), int(npix), int(h), int(w),
                                     _p(gate), int(gate_c), _stream()), "skinny_group_fwd")


# --------------------------------------------------------------------------------------------- fp8 operand path (F8K)
def f8k_to_nchw(y8, B, C, H, W, scale):
    """F8K uint8 buffer -> float32 NCHW (torch ops; tests only)."""
    c32 = (C + 31) // 32
    t = y8.cpu().view(torch.float8_e4m3fn).float().view(B, c32, H * W, 32).permute(0, 1, 3, 2).reshape(B, c32 * 32, H, W)
    return (t[:, :C] * scale).contiguous().to(y8.device)


def nchw_to_f8k(x, scale, C=None, coff=0, in_op=INOP_NONE):
    """float32 NCHW (channel view) -> F8K fp8 [B][ceil32(C)/32][HW][32] holding fp8(x / scale) (saturating)."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    C = ctot if C is None else C
    y = torch.empty(_int_query("masic_f8k_bytes", B, C, H * W), dtype=torch.uint8, device=x.device)
    check(lib.masic_nchw_to_f8k(_p(x), _p(y), B, C, H * W, ctot, coff, int(in_op), 1.0 / float(scale), _stream()), "nchw_to_f8k")
    return y


def absmax(t, into=None):
    """Device float32 scalar max(|t|) (float32 tensors, or F16K int16 buffers read as bf16); `into`: running maximum."""
    if not t.is_cuda:
        raise RuntimeError("masic_amd.absmax: device tensors only")
    out = into if into is not None else torch.zeros(1, dtype=torch.float32, device=t.device)
    bf16 = t.dtype in (torch.int16, torch.bfloat16)
    if not bf16 and t.dtype != torch.float32:
        raise RuntimeError("masic_amd.absmax: float32 or bf16 (F16K) data")
    check(lib.masic_absmax(_p(t), t.numel(), int(bf16), _p(out), _stream()), "absmax")
    return out


def pack_conv_f8k_weight(weight, desc):
    """-> (fp8 slab stream, per-output-channel weight scales [Cout]) for masic_conv_f8k_fwd (desc.prec = PREC_FP8)."""
    _dev(weight, "weight")
    nbytes = _desc_query("masic_conv_f16k_packed_bytes", desc)
    if nbytes == 0:
        check(-1, "conv_f8k_packed_bytes")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    ws = torch.empty(desc.Cout, dtype=torch.float32, device=weight.device)
    check(lib.masic_conv_f8k_pack_weight(_p(weight), _p(packed), _p(ws), ctypes.byref(desc), _stream()), "conv_f8k_pack_weight")
    return packed, ws


def conv2d_f8k(x, packed, wscale, bias, desc, out="f16k", out_scale=None, out_nchw=None, gate=None, gdn=None):
    """Conv on an F8K (desc.prec = PREC_FP8, `wscale` = weight scales x input scale) or F16K (PREC_BF16, wscale None) input.
    out: "nchw" (float32, or the channel view `out_nchw`), "f16k" (bf16) or "f8k" (fp8 of y / out_scale)."""
    if not x.is_cuda or x.dtype not in (torch.uint8, torch.int16):
        raise RuntimeError("masic_amd.conv2d_f8k: input must be an F8K uint8 / F16K int16 CUDA buffer")
    f8 = desc.prec == _lib.PREC_FP8
    if (x.dtype == torch.uint8) != f8:
        raise RuntimeError("masic_amd.conv2d_f8k: input format does not match desc.prec")
    per_rec = 32 if f8 else 16
    if x.numel() != desc.B * desc.in_ctot * desc.Hi * desc.Wi:
        raise RuntimeError(f"masic_amd.conv2d_f8k: input buffer of {x.numel()} elements does not match the descriptor (records of {per_rec} channels)")
    if f8:
        _dev(wscale, "wscale")
        if wscale.numel() != desc.Cout:
            raise RuntimeError("masic_amd.conv2d_f8k: wscale must have Cout entries")
    if bias is not None:
        _dev(bias, "bias")
    y32 = y16 = y8 = None
    inv = 0.0
    if out == "nchw" or out_nchw is not None:
        y32 = out_nchw if out_nchw is not None else torch.empty((desc.B, desc.out_ctot, desc.Ho, desc.Wo), dtype=torch.float32, device=x.device)
        if tuple(y32.shape) != (desc.B, desc.out_ctot, desc.Ho, desc.Wo):
            raise RuntimeError("masic_amd.conv2d_f8k: output buffer does not match the descriptor")
    elif out == "f8k":
        if not out_scale or out_scale <= 0:
            raise RuntimeError("masic_amd.conv2d_f8k: an fp8 output needs out_scale > 0")
        y8 = torch.empty(desc.B * desc.out_ctot * desc.Ho * desc.Wo, dtype=torch.uint8, device=x.device)
        inv = 1.0 / float(out_scale)
    else:
        y16 = torch.empty(desc.B * desc.out_ctot * desc.Ho * desc.Wo, dtype=torch.int16, device=x.device)
    gp, ginv = (None, 0) if gdn is None else (gdn[0], _gdn_flags(gdn[1]))
    timed = None
    if _timer is not None:
        buf = ctypes.create_string_buffer(96)
        lib.masic_conv_f16k_kernel_name(ctypes.byref(desc), int(gdn is not None), buf, 96)
        variant = buf.value.decode()
        if _timer.only is None or variant == _timer.only:
            timed = (variant, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timed[1].record()
    check(lib.masic_conv_f8k_fwd(_p(x), _p(packed), _p(wscale) if f8 else None, _p(bias), _p(gate), _p(gp), ginv, _p(y32), _p(y16), _p(y8),
                                 float(inv), ctypes.byref(desc), _stream()), "conv_f8k_fwd")
    if timed is not None:
        timed[2].record()
        flops, nbytes = conv_algorithmic_work(desc)
        _timer.records.append((timed[0], 1, flops, nbytes, timed[1], timed[2]))
    return y32 if y32 is not None else (y8 if y8 is not None else y16)


def conv_a_gdn_f8k(x, packed, bias, gdn, out_scale, in_coff=0):
    """conv_a_gdn_f16k with the result stored as F8K fp8(y / out_scale)."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B * 128 * Ho * Wo, dtype=torch.uint8, device=x.device)
    check(lib.masic_conv_a_gdn_fwd_ex(_p(x), _p(packed), _p(bias), _p(gdn[0]), _gdn_flags(gdn[1]), None, _p(y), 1.0 / float(out_scale), B, H, W, ctot, in_coff,
                                      _stream()), "conv_a_gdn_fwd_ex")
    return y, Ho, Wo


def pack_gemm_f8k_weight(weight, Cin, Cout, transposed):
    _dev(weight, "weight")
    wp = torch.empty(_int_query("masic_gemm_f8k_packed_bytes", Cin, Cout), dtype=torch.uint8, device=weight.device)
    ws = torch.empty(Cout, dtype=torch.float32, device=weight.device)
    check(lib.masic_gemm_f8k_pack_weight(_p(weight), _p(wp), _p(ws), Cin, Cout, int(transposed), _stream()), "gemm_f8k_pack_weight")
    return wp, ws


def gemm_f8k(x_f8k, wp, wscale, bias, B, Cin, Cout, H, W, act, out="f16k", out_scale=None, out_nchw=None, out_coff=0):
    """y = act(W x + b) on F8K activations with fp8 weights; out: "f16k" (bf16), "f8k" (fp8 of y / out_scale) or "nchw"."""
    HW = H * W
    y32 = y16 = y8 = None
    inv = 0.0
    if out == "nchw" or out_nchw is not None:
        y32 = out_nchw if out_nchw is not None else torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_f8k.device)
        out_ctot = y32.shape[1]
    elif out == "f8k":
        out_ctot = (Cout + 31) // 32 * 32
        y8 = torch.empty(B * out_ctot * HW, dtype=torch.uint8, device=x_f8k.device)
        inv = 1.0 / float(out_scale)
    else:
        out_ctot = (Cout + 15) // 16 * 16
        y16 = torch.empty(B * out_ctot * HW, dtype=torch.int16, device=x_f8k.device)
    check(lib.masic_gemm_f8k_fwd(_p(x_f8k), _p(wp), _p(_dev(wscale, "wscale")), _p(bias), _p(y16), _p(y8), _p(y32), float(inv), B, Cin, Cout, HW,
                                 out_ctot, out_coff, int(act), _stream()), "gemm_f8k_fwd")
    return y32 if y32 is not None else (y8 if y8 is not None else y16)


# --------------------------------------------------------------------------------------------- F16K chains of Independent_EN
def f16k_empty(B, C, H, W, device):
    """Uninitialised F16K buffer of C (multiple of 16) channels."""
    if C % 16:
        raise RuntimeError("masic_amd.f16k_empty: channels must be a multiple of 16")
    return torch.empty(B * C * H * W, dtype=torch.int16, device=device)


def f16k_gate(src16, B, C, H, W, dst16, dst_ctot, dst_coff, gate=None, gate_c=0, minv=None):
    """dst[:, dst_coff:dst_coff+C] = (warp(src, minv) if minv is not None else src) * (gate[:, gate_c] if gate is not None else 1)."""
    if src16.dtype != torch.int16 or dst16.dtype != torch.int16 or src16.numel() != B * C * H * W or dst16.numel() != B * dst_ctot * H * W:
        raise RuntimeError("masic_amd.f16k_gate: F16K buffer sizes do not match (B, C, H, W)")
    gct = 0
    if gate is not None:
        _dev(gate, "gate")
        if gate.shape[0] != B or tuple(gate.shape[-2:]) != (H, W):
            raise RuntimeError("masic_amd.f16k_gate: gate shape mismatch")
        gct = gate.shape[1]
    if minv is not None:
        _dev(minv, "warp matrix")
        if tuple(minv.shape) != (B, 3, 3):
            raise RuntimeError("masic_amd.f16k_gate: matrix batch mismatch")
    check(lib.masic_f16k_gate(_p(src16), _p(gate), _p(minv), _p(dst16), B, C, H, W, dst_ctot, dst_coff, gct, gate_c, _stream()), "f16k_gate")
    return dst16


def nchw_to_f16k_view(x, dst16, dst_ctot, dst_coff, C=None, coff=0, in_op=INOP_NONE, gate=None, gate_c=0):
    """float32 NCHW (channel view) -> channels dst_coff.. of the F16K buffer dst16 (dst_ctot channels); in_op: |x| / round(x) on the
    way, gate: multiply by gate[:, gate_c] before the rounding to bf16."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    C = ctot if C is None else C
    if dst16.dtype != torch.int16 or dst16.numel() != B * dst_ctot * H * W:
        raise RuntimeError("masic_amd.nchw_to_f16k_view: destination buffer size mismatch")
    if gate is not None and (_dev(gate, "gate").dim() != 4 or gate.shape[0] != B or tuple(gate.shape[2:]) != (H, W)):
        raise RuntimeError("masic_amd.nchw_to_f16k_view: gate does not match the tensor")
    check(lib.masic_nchw_to_f16k_view_op(_p(x), _p(dst16), B, C, H * W, ctot, coff, dst_ctot, dst_coff, int(in_op), _p(gate),
                                         0 if gate is None else gate.shape[1], gate_c, _stream()), "nchw_to_f16k_view")
    return dst16


def f16k_to_nchw_dev(x16, B, C, H, W, src_ctot=None, src_coff=0, bf16=False):
    """F16K channel slice -> float32 NCHW, or bf16 NCHW (bf16=True: exact, no rounding -- F16K holds bf16) (HIP kernel; `f16k_to_nchw`
    above is the torch-op form the tests use as a checker)."""
    src_ctot = C if src_ctot is None else src_ctot
    if x16.dtype != torch.int16 or x16.numel() != B * src_ctot * H * W:
        raise RuntimeError("masic_amd.f16k_to_nchw_dev: source buffer size mismatch")
    y = torch.empty((B, C, H, W), dtype=torch.bfloat16 if bf16 else torch.float32, device=x16.device)
    fn = lib.masic_f16k_to_nchw_bf16 if bf16 else lib.masic_f16k_to_nchw
    check(fn(_p(x16), _p(y), B, C, H * W, src_ctot, src_coff, C, 0, _stream()), "f16k_to_nchw")
    return y


def conv2d_f16k_res(x16, packed, bias, desc, y16=None, res1=None, res2=None, res_ctot=0, mask=None, mask_slope=0.0, y_pre=None):
    """F16K in -> F16K out (channel view desc.out_ctot / out_coff of `y16`, or a fresh buffer) + residual F16K tensors.
    Training-step pieces (masic_conv_f16k_res_ex_fwd): `mask` multiplies by act'(mask) before the adds (input gradients), `y_pre`
    receives the value before the adds; both are F16K tensors of res_ctot channels like the residuals."""
    if x16.dtype != torch.int16 or x16.numel() != desc.B * desc.in_ctot * desc.Hi * desc.Wi:
        raise RuntimeError("masic_amd.conv2d_f16k_res: input buffer does not match the descriptor")
    n_out = desc.B * desc.out_ctot * desc.Ho * desc.Wo
    if y16 is None:
        y16 = torch.empty(n_out, dtype=torch.int16, device=x16.device)
    elif y16.dtype != torch.int16 or y16.numel() != n_out:
        raise RuntimeError("masic_amd.conv2d_f16k_res: output buffer does not match the descriptor")
    for r in (res1, res2, mask, y_pre):
        if r is not None and (r.dtype != torch.int16 or r.numel() != desc.B * res_ctot * desc.Ho * desc.Wo):
            raise RuntimeError("masic_amd.conv2d_f16k_res: residual / mask / pre buffer does not match (B, res_ctot, Ho, Wo)")
    timed = None
    if _timer is not None:
        buf = ctypes.create_string_buffer(96)
        lib.masic_conv_f16k_kernel_name(ctypes.byref(desc), 0, buf, 96)
        variant = buf.value.decode()
        if _timer.only is None or variant == _timer.only:
            timed = (variant, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timed[1].record()
    if mask is None and y_pre is None:
        check(lib.masic_conv_f16k_res_fwd(_p(x16), _p(packed), _p(bias), _p(res1), _p(res2), int(res_ctot), _p(y16), ctypes.byref(desc), _stream()),
              "conv_f16k_res_fwd")
    else:
        check(lib.masic_conv_f16k_res_ex_fwd(_p(x16), _p(packed), _p(bias), _p(res1), _p(res2), int(res_ctot), _p(mask), float(mask_slope), _p(y_pre),
                                             _p(y16), ctypes.byref(desc), _stream()), "conv_f16k_res_ex_fwd")
    if timed is not None:
        timed[2].record()
        flops, nbytes = conv_algorithmic_work(desc)
        _timer.records.append((timed[0], 1, flops, nbytes, timed[1], timed[2]))
    return y16


def mask2weights_en(mask, params):
    """mask2weights_EN (Kw = 2) in one launch: mask [B,1,H,W] float32 -> gates [B,2,H,W]; params = (w1, b1, ..., w4, b4) of the four
    3x3 convolutions 1 -> 2 -> 4 -> 4 -> 2 (csrc/m2w.hip)."""
    _dev(mask, "mask")
    B, c, H, W = mask.shape
    shapes = [(2, 1), (4, 2), (4, 4), (2, 4)]
    if c != 1 or len(params) != 8 or any(tuple(params[2 * i].shape) != (co, ci, 3, 3) or tuple(params[2 * i + 1].shape) != (co,) for i, (co, ci) in enumerate(shapes)):
        raise RuntimeError("masic_amd.mask2weights_en: a [B,1,H,W] mask and the 1 -> 2 -> 4 -> 4 -> 2 stack's weights / biases expected")
    ps = [_dev(p.detach().contiguous(), "parameter") for p in params]
    out = torch.empty((B, 2, H, W), dtype=torch.float32, device=mask.device)
    check(lib.masic_mask2weights_en_fwd(_p(mask.contiguous()), *[_p(p) for p in ps], _p(out), B, H, W, _stream()), "mask2weights_en_fwd")
    return out


def conv3x3_resident_supported(B, Cin, Cout, H, W):
    return bool(lib.masic_conv3x3_resident_supported(int(B), int(Cin), int(Cout), int(H), int(W)))


def pack_conv3x3_resident_weight(weight, transposed=False):
    """Conv2d(Cin -> Cout, 3x3) weight [Cout, Cin, 3, 3] -> the LDS-resident fragment slabs of masic_conv3x3_resident_fwd;
    transposed: the slabs of the INPUT gradient of the layer this weight belongs to (a Cout -> Cin convolution)."""
    _dev(weight, "weight")
    co, ci = weight.shape[:2]
    cin, cout = (co, ci) if transposed else (ci, co)
    nbytes = _int_query("masic_conv3x3_resident_packed_bytes", cin, cout)
    if tuple(weight.shape[2:]) != (3, 3) or nbytes == 0:
        raise RuntimeError("masic_amd.pack_conv3x3_resident_weight: a 3x3 weight with a resident configuration (Cout 32 | 64, Cin <= Cout) expected")
    wp = torch.empty(nbytes // 2, dtype=torch.int16, device=weight.device)
    check(lib.masic_conv3x3_resident_pack_weight(_p(weight.contiguous()), _p(wp), cin, cout, int(transposed), _stream()), "conv3x3_resident_pack_weight")
    return wp


def conv3x3_resident(x16, packed, bias, B, Cin, Cout, H, W, act=ACT_NONE, y16=None, out_ctot=None, out_coff=0, in_ctot=None, in_coff=0,
                     res1=None, res2=None, res_ctot=0, mask=None, mask_slope=0.0, y_pre=None, mask2=None, mask2_slope=0.0, sum_of=0):
    """y = act(conv3x3(x) + bias) * act'(mask) + res1 + res2 on F16K buffers with LDS-resident weights (csrc/conv_f16k.hip:
    conv3x3_resident_f16k); operands as conv2d_f16k_res.  The input buffer holds ceil16(Cin) channels per pixel.
    mask2: also y2 = y * act'(mask2) (the next layer's dy of a backward chain); sum_of = 1 / 2: also the per-channel sums of the value
    before the residual adds / of y2 (a bias gradient).  Returns y, or (y, y2, sums) -- None where not asked -- when either is given."""
    in_ctot = (Cin + 15) // 16 * 16 if in_ctot is None else in_ctot
    out_ctot = Cout if out_ctot is None else out_ctot
    if x16.dtype != torch.int16 or x16.numel() != B * in_ctot * H * W:
        raise RuntimeError("masic_amd.conv3x3_resident: input buffer does not match (B, in_ctot, H, W)")
    if y16 is None:
        y16 = torch.empty(B * out_ctot * H * W, dtype=torch.int16, device=x16.device)
    elif y16.dtype != torch.int16 or y16.numel() != B * out_ctot * H * W:
        raise RuntimeError("masic_amd.conv3x3_resident: output buffer does not match (B, out_ctot, H, W)")
    for r in (res1, res2, mask, y_pre, mask2):
        if r is not None and (r.dtype != torch.int16 or r.numel() != B * res_ctot * H * W):
            raise RuntimeError("masic_amd.conv3x3_resident: residual / mask / pre buffer does not match (B, res_ctot, H, W)")
    if mask2 is None and not sum_of:
        check(lib.masic_conv3x3_resident_fwd(_p(x16), _p(packed), _p(bias), _p(res1), _p(res2), int(res_ctot), _p(mask), float(mask_slope), _p(y_pre), _p(y16),
                                             B, Cin, Cout, H, W, in_ctot, in_coff, out_ctot, out_coff, int(act), _stream()), "conv3x3_resident_fwd")
        return y16
    y2 = torch.empty(B * res_ctot * H * W, dtype=torch.int16, device=x16.device) if mask2 is not None else None
    sums = torch.empty(Cout, dtype=torch.float32, device=x16.device) if sum_of else None
    ws = torch.empty(_int_query("masic_conv3x3_resident_sum_workspace_bytes") // 4, dtype=torch.float32, device=x16.device) if sum_of else None
    check(lib.masic_conv3x3_resident_ex_fwd(_p(x16), _p(packed), _p(bias), _p(res1), _p(res2), int(res_ctot), _p(mask), float(mask_slope), _p(y_pre), _p(y16),
                                            _p(mask2), float(mask2_slope), _p(y2), int(sum_of), _p(sums), _p(ws),
                                            B, Cin, Cout, H, W, in_ctot, in_coff, out_ctot, out_coff, int(act), _stream()), "conv3x3_resident_ex_fwd")
    return y16, y2, sums


def f16k_act_bwd(g16, y16, slope):
    """g * act'(y) on F16K buffers (slope 0.01: LeakyReLU, 0: ReLU); y is the activation's OUTPUT."""
    if g16.dtype != torch.int16 or y16.dtype != torch.int16 or g16.numel() != y16.numel() or g16.numel() % 8:
        raise RuntimeError("masic_amd.f16k_act_bwd: F16K buffers of equal size expected")
    out = torch.empty_like(g16)
    check(lib.masic_f16k_act_bwd(_p(g16), _p(y16), _p(out), g16.numel(), float(slope), _stream()), "f16k_act_bwd")
    return out


def f16k_act_bwd_sum(g16, y16, slope, B, C, HW):
    """(g * act'(y), its per-channel sums [C]) in one pass (masic_f16k_act_bwd_sum)."""
    if g16.dtype != torch.int16 or y16.dtype != torch.int16 or g16.numel() != y16.numel() or g16.numel() != B * C * HW or C % 16:
        raise RuntimeError("masic_amd.f16k_act_bwd_sum: F16K buffers of (B, C, HW), C % 16 == 0 expected")
    out = torch.empty_like(g16)
    sums = torch.empty(C, dtype=torch.float32, device=g16.device)
    ws = torch.empty(_int_query("masic_f16k_channel_sum_workspace_bytes", B, C) // 4, dtype=torch.float32, device=g16.device)
    check(lib.masic_f16k_act_bwd_sum(_p(g16), _p(y16), _p(out), _p(sums), _p(ws), B, C, HW, float(slope), _stream()), "f16k_act_bwd_sum")
    return out, sums


def f16k_channel_sum(x16, B, C, HW):
    """float32 [C] sums over (image, pixel) of an F16K tensor (a bias gradient)."""
    if x16.dtype != torch.int16 or C % 16 or x16.numel() != B * C * HW:
        raise RuntimeError("masic_amd.f16k_channel_sum: buffer does not match (B, C, HW), C % 16 == 0")
    out = torch.empty(C, dtype=torch.float32, device=x16.device)
    ws = torch.empty(_int_query("masic_f16k_channel_sum_workspace_bytes", B, C) // 4, dtype=torch.float32, device=x16.device)
    check(lib.masic_f16k_channel_sum(_p(x16), _p(out), _p(ws), B, C, HW, _stream()), "f16k_channel_sum")
    return out


def conv2d_f16k_few(x16, packed, bias32, desc, C, res32=None):
    """Convolution to C <= 32 channels (desc: the zero-padded 32-channel form) on an F16K input -> float32 NCHW [B, C, Ho, Wo] (+ res32)."""
    if x16.dtype != torch.int16 or x16.numel() != desc.B * desc.in_ctot * desc.Hi * desc.Wi:
        raise RuntimeError("masic_amd.conv2d_f16k_few: input buffer does not match the descriptor")
    y = torch.empty((desc.B, C, desc.Ho, desc.Wo), dtype=torch.float32, device=x16.device)
    if res32 is not None:
        _dev(res32, "residual")
        if tuple(res32.shape) != tuple(y.shape):
            raise RuntimeError("masic_amd.conv2d_f16k_few: residual shape mismatch")
    check(lib.masic_conv_f16k_few_fwd(_p(x16), _p(packed), _p(bias32), _p(res32), _p(y), int(C), ctypes.byref(desc), _stream()), "conv_f16k_few_fwd")
    return y


# --------------------------------------------------------------------------------------------- training-mode fused forward (dual outputs)
def conv2d_f16k_gdn_dual(x16, packed, bias, desc, gdn, products=None):
    """F16K in -> (conv + bias before the GDN, GDN result), both F16K of desc.out_ctot channels.  gdn = (pack_gdn_f16k(..), inverse)."""
    if x16.dtype != torch.int16 or x16.numel() != desc.B * desc.in_ctot * desc.Hi * desc.Wi:
        raise RuntimeError("masic_amd.conv2d_f16k_gdn_dual: input buffer does not match the descriptor")
    n = desc.B * desc.out_ctot * desc.Ho * desc.Wo
    pre = torch.empty(n, dtype=torch.int16, device=x16.device)
    y = torch.empty(n, dtype=torch.int16, device=x16.device)
    check(lib.masic_conv_f16k_gdn_dual_fwd(_p(x16), _p(packed), _p(bias), _p(gdn[0]), _gdn_flags(gdn[1], products), _p(pre), _p(y), ctypes.byref(desc), _stream()),
          "conv_f16k_gdn_dual_fwd")
    return pre, y


def conv_a_gdn_dual(x, packed, bias, gdn, in_coff=0, products=None):
    """First analysis layer + GDN -> (pre-GDN F16K, post-GDN F16K, Ho, Wo)."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pre = torch.empty(B * 128 * Ho * Wo, dtype=torch.int16, device=x.device)
    y = torch.empty(B * 128 * Ho * Wo, dtype=torch.int16, device=x.device)
    check(lib.masic_conv_a_gdn_dual_fwd(_p(x), _p(packed), _p(bias), _p(gdn[0]), _gdn_flags(gdn[1], products), _p(pre), _p(y), B, H, W, ctot, in_coff, _stream()),
          "conv_a_gdn_dual_fwd")
    return pre, y, Ho, Wo


def conv_a_f16k(x, packed, bias=None, in_coff=0):
    """Conv2d(3 -> 128, k5, s2, p2) on channels in_coff.. of a float32 NCHW tensor, no GDN -> (F16K, Ho, Wo) (masic_conv_a_fwd)."""
    _dev(x, "x")
    B, ctot, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B * 128 * Ho * Wo, dtype=torch.int16, device=x.device)
    check(lib.masic_conv_a_fwd(_p(x), _p(packed), _p(bias), _p(y), B, H, W, ctot, in_coff, _stream()), "conv_a_fwd")
    return y, Ho, Wo


def deconv_s2_as_conv_weight_dev(weight, bias=None):
    """deconv_s2_as_conv_weight in one launch (masic_deconv_s2_as_conv_weight): weight [Cin, C <= 8, 5, 5] -> ([32, Cin, 3, 3], [32])."""
    _dev(weight, "weight")
    Cin, C, KH, KW = weight.shape
    if (KH, KW) != (5, 5) or 4 * C > 32 or not weight.is_contiguous():
        raise RuntimeError("masic_amd.deconv_s2_as_conv_weight_dev: needs a contiguous 5x5 kernel and at most 8 output channels")
    w = torch.empty((32, Cin, 3, 3), dtype=torch.float32, device=weight.device)
    b = torch.empty(32, dtype=torch.float32, device=weight.device)
    check(lib.masic_deconv_s2_as_conv_weight(_p(weight), _p(bias), _p(w), _p(b), Cin, C, _stream()), "deconv_s2_as_conv_weight")
    return w, b


def gemm_wgrad_f16k(rows16, cols16, B, CA, CQ, HW, bias_of=0):
    """dw [CA, CQ] = sum_{b,p} rows[b, a, p] * cols[b, q, p] from F16K operands (masic_gemm_wgrad_f16k): the weight gradient of a 1x1
    layer -- Conv2d: (dy, x) -> [Cout, Cin]; ConvTranspose2d(k=1): (x, dy) -> [Cin, Cout].
    bias_of = 1 / 2: also the sums over batch and pixels of the rows / columns operand (the layer's bias gradient when that operand is
    dy) from the same launch; returns (dw, sums) -- two views of one buffer."""
    if rows16.dtype != torch.int16 or cols16.dtype != torch.int16 or rows16.numel() != B * CA * HW or cols16.numel() != B * CQ * HW:
        raise RuntimeError("masic_amd.gemm_wgrad_f16k: F16K buffer sizes do not match (B, C, HW)")
    nb = CA if bias_of == 1 else (CQ if bias_of == 2 else 0)
    out = torch.empty(CA * CQ + nb, dtype=torch.float32, device=rows16.device)
    check(lib.masic_gemm_wgrad_bias_f16k(_p(rows16), _p(cols16), _p(out), int(bias_of), B, CA, CQ, HW, _stream()), "gemm_wgrad_f16k")
    dw = out[:CA * CQ].view(CA, CQ)
    return (dw, out[CA * CQ:]) if nb else dw


_PIC_WGRAD_WS = {}


def pic_wgrad_f16k(p16, q, B, Hc, Wc, q_coff=0):
    """dW (float32, flat 128*3*25 = [a][q][5][5]) of the picture-end 5x5 stride-2 layers (masic_pic_wgrad_f16k): p16 = the 128-channel
    operand at Hc x Wc in F16K (dy of Conv2d(3 -> 128) / x of ConvTranspose2d(128 -> 3)), q = the 3-channel one, float32 NCHW at 2Hc x 2Wc."""
    _dev(q, "q")
    if p16.dtype != torch.int16 or p16.numel() != B * 128 * Hc * Wc or q.dim() != 4 or q.shape[0] != B or tuple(q.shape[2:]) != (2 * Hc, 2 * Wc) \
            or q_coff + 3 > q.shape[1]:
        raise RuntimeError("masic_amd.pic_wgrad_f16k: operand shapes do not match (B, 128, Hc, Wc) / (B, >= 3, 2 Hc, 2 Wc)")
    key = (q.device, int(_stream().value or 0))
    ws = _PIC_WGRAD_WS.get(key)
    if ws is None:
        ws = _PIC_WGRAD_WS[key] = torch.empty(lib.masic_pic_wgrad_f16k_workspace_bytes() // 4, dtype=torch.float32, device=q.device)
    dw = torch.empty(128 * 75, dtype=torch.float32, device=q.device)
    check(lib.masic_pic_wgrad_f16k(_p(p16), _p(q), _p(dw), _p(ws), B, Hc, Wc, q.shape[1], q_coff, _stream()), "pic_wgrad_f16k")
    return dw


def conv5x5_wgrad_f16k(x16, dy16, B, Cin, Cout, H, W):
    """dW [Cout, Cin, 5, 5] (float32) of Conv2d(k5, s1, p2) from x and dy in F16K (bf16 operands, float32 accumulate)."""
    if x16.dtype != torch.int16 or dy16.dtype != torch.int16 or x16.numel() != B * Cin * H * W or dy16.numel() != B * Cout * H * W:
        raise RuntimeError("masic_amd.conv5x5_wgrad_f16k: F16K buffer sizes do not match (B, C, H, W)")
    dw = torch.empty((Cout, Cin, 5, 5), dtype=torch.float32, device=x16.device)
    n = _int_query("masic_conv5x5_wgrad_f16k_workspace_bytes", Cin, Cout) // 4
    ws = _clean_workspace(x16.device, n)
    try:
        check(lib.masic_conv5x5_wgrad_f16k_ws(_p(x16), _p(dy16), _p(dw), _p(ws), B, Cin, Cout, H, W, 1, _stream()), "conv5x5_wgrad_f16k")
    except Exception:
        _drop_workspace(x16.device, n)
        raise
    return dw


def conv3x3_wgrad_f16k(x16, dy16, B, Cin, Cout, H, W):
    """dW [Cout, Cin, 3, 3] (float32) of Conv2d(k3, s1, p1) from x and dy in F16K (bf16 operands, float32 accumulate)."""
    if x16.dtype != torch.int16 or dy16.dtype != torch.int16 or x16.numel() != B * Cin * H * W or dy16.numel() != B * Cout * H * W:
        raise RuntimeError("masic_amd.conv3x3_wgrad_f16k: F16K buffer sizes do not match (B, C, H, W)")
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x16.device)
    n = _int_query("masic_conv3x3_wgrad_f16k_workspace_bytes", Cin, Cout) // 4
    ws = _clean_workspace(x16.device, n)
    try:
        check(lib.masic_conv3x3_wgrad_f16k_ws(_p(x16), _p(dy16), _p(dw), _p(ws), B, Cin, Cout, H, W, 1, _stream()), "conv3x3_wgrad_f16k")
    except Exception:
        _drop_workspace(x16.device, n)
        raise
    return dw
