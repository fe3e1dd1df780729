"""fp8 (OCP e4m3fn) operand mode of the inference forward -- BASELINE.json configs[4].

Which layers: the 128 -> 128 5x5 layers of the analysis / synthesis transforms (g_a_conv2/3, g_s_conv2/3: 52 % of the
forward's FLOPs) and the first two 1x1 layers of every entropy-parameter stack (15 %) run
`v_mfma_scale_f32_32x32x64_f8f6f4` (twice the bf16 MFMA rate, half the operand bytes); everything that produces a latent, a
likelihood parameter or a picture (g_a_conv4, the hyper transforms, the stacks' last layers, g_s_conv1 -- its input are
integers up to +-20, which e4m3 does not hold exactly -- and g_s_conv4) stays on bf16 operands, the entropy models and the
warp on float32: the int32 symbols that feed the range coder come out of a bf16 layer.

Scaling: one scale per activation tensor (x is stored as fp8(x / scale), saturating at +-448), fixed by `calibrate` from the
largest magnitude the tensor takes on calibration batches in the bf16 mode (scale = amax x margin / 448); one scale per output
channel of each weight (max|W[co]| / 448, computed when the weight is packed).  Producers quantise in their epilogue
(after the fused GDN / activation), consumers dequantise the float32 accumulator with weight scale x input scale.

There is no reference counterpart: the reference computes in float32.  The budget this mode is held to against the oracle
is declared and tested in tests/test_gpu_fp8.py and reported by `bench.py --precision fp8`.
"""
import torch

from . import nn as _mnn
from . import ops

MARGIN = 1.5            # head-room over the calibration maximum (values beyond saturate)
_RECORD = False


def recording():
    return _RECORD


def record(owner, tag, t):
    """During calibration: running max|t| of the tensor `tag` produced inside module `owner` (device side, no sync)."""
    if not _RECORD:
        return
    d = owner.__dict__.setdefault("_fp8_amax", {})
    d[tag] = ops.absmax(t, into=d.get(tag))


def scales(owner):
    """{tag: scale} of a calibrated module while the fp8 mode is on, else None."""
    if not _mnn._FP8:
        return None
    return owner.__dict__.get("_fp8_scale")


def calibrate(net, batches, margin=MARGIN):
    """Runs the eval forward of `net` (HSIC) in the bf16 mode on `batches` = iterable of (x1, x2, h_matrix), records the
    largest magnitude of every tensor the fp8 mode quantises, and stores the scales on the producing modules."""
    global _RECORD
    if net.training:
        raise RuntimeError("masic_amd.fp8.calibrate: eval mode only")
    prev = _mnn.get_precision()
    for m in net.modules():
        m.__dict__.pop("_fp8_amax", None)
        m.__dict__.pop("_fp8_scale", None)
    _mnn.set_precision("bf16")
    _RECORD = True
    try:
        with torch.no_grad():
            for x1, x2, hm in batches:
                net(x1, x2, hm)
        torch.cuda.synchronize()
    finally:
        _RECORD = False
        _mnn.set_precision(prev)
    table = {}
    for name, m in net.named_modules():
        amax = m.__dict__.pop("_fp8_amax", None)
        if amax:
            m.__dict__["_fp8_scale"] = {k: max(float(v.item()), 1e-12) * margin / 448.0 for k, v in amax.items()}
            table[name] = dict(m.__dict__["_fp8_scale"])
    return table


def export_scales(net):
    return {name: dict(m.__dict__["_fp8_scale"]) for name, m in net.named_modules() if "_fp8_scale" in m.__dict__}


def load_scales(net, table):
    mods = dict(net.named_modules())
    for name, sc in table.items():
        mods[name].__dict__["_fp8_scale"] = {k: float(v) for k, v in sc.items()}


# ---- the scales as part of a bitstream (HSIC.compress / decompress in the fp8 mode; container: masic_amd/codec.py)
_STREAM_OWNERS = ("encoder1", "encoder2", "decoder1", "decoder2", "_h_s1_same_resolution", "_h_s2_same_resolution")


def stream_table(net):
    """The calibration table of `net` as bytes for the .bin header.  Raises when a module the fp8 mode quantises in has no scales:
    such a module would silently fall back to bf16 operands under a stream stamped fp8."""
    import json
    table = export_scales(net)
    missing = [n for n in _STREAM_OWNERS if hasattr(net, n) and n not in table]
    if missing:
        raise RuntimeError("HSIC.compress in the fp8 operand mode needs masic_amd.fp8.calibrate(net, batches) (or load_scales) first: "
                           "no activation scales on " + ", ".join(missing))
    return json.dumps(table, sort_keys=True).encode()         # (repr of a Python float round-trips exactly)


class stream_scales:
    """Context of HSIC.decompress: while decoding, the modules carry the scales the ENCODER used (read from the stream); whatever the
    decoder's own calibration was comes back afterwards.  `raw` None: nothing to do (f32 / bf16 streams)."""

    def __init__(self, net, raw):
        self.net, self.raw = net, raw

    def __enter__(self):
        if self.raw is None:
            return self
        import json
        self.saved = {name: m.__dict__.get("_fp8_scale") for name, m in self.net.named_modules()}
        for m in self.net.modules():
            m.__dict__.pop("_fp8_scale", None)
        load_scales(self.net, json.loads(self.raw.decode()))
        return self

    def __exit__(self, *exc):
        if self.raw is None:
            return False
        for name, m in self.net.named_modules():
            m.__dict__.pop("_fp8_scale", None)
            if self.saved.get(name) is not None:
                m.__dict__["_fp8_scale"] = self.saved[name]
        return False
