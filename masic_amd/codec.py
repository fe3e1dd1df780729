"""Coding of the y latents for HSIC.compress / decompress (reference MASIC.py:855-1408).

The reference walks the latent in raster order and, per pixel, re-runs the masked context convolution on a 5x5 crop and
the nine 1x1 head layers, builds one CDF per non-zero channel on the host and calls a Python range coder per symbol.
Here:
  * the coding order is the anti-diagonal wavefront t = j + 3 i of the type-A 5x5 mask (pixel (i, j) needs rows i-2, i-1 up
    to column j+2 and the two pixels left of it: all have a smaller t), so a 32x32 latent takes 125 dependent steps
    instead of 1024, each coding all pixels of its wavefront x all non-zero channels at once;
  * the tables are built on the device (masic_gmm_cdf_rows) and only cross to the host as u16 rows for the decoder, or
    as (start, freq) pairs for the encoder;
  * the coder is the library's rANS (masic_rans_*), one stream per view.
Encoder and decoder evaluate the SAME kernels on tensors of the same shape, and the masked taps of the context
convolution are exact zeros, so a not-yet-decoded neighbour (zero in the decoder, the true value in the encoder) never
changes a bit of the parameters of the pixel being coded: the encoder needs ONE pass over the full latent, the decoder
one pass per wavefront.
Container (.bin): b"MSR1", u8 precision id (0 f32 / 1 bf16 / 2 fp8 operands: the tables depend on it), u8 flags (bit 0: an
activation-scale table follows), 2 pad bytes, [u32 length + the fp8 mode's calibration table as JSON, masic_amd/fp8.py], then per
view u32 length + rANS words.  The .npz header keeps the reference's layout (MASIC.py:916-948)."""
import ctypes

import numpy as np
import torch

from ._lib import check, lib

MAGIC = b"MSR1"


def wavefront_steps(h, w):
    """Pixel indices (i * w + j), rows ascending, of each wavefront t = j + 3 i."""
    ii = np.arange(h)
    steps = []
    for t in range(w + 3 * (h - 1)):
        j = t - 3 * ii
        keep = (j >= 0) & (j < w)
        steps.append((ii[keep] * w + j[keep]).astype(np.int32))
    return steps


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def gmm_tables(sigma, mu, logits, M, K, pix, chan, minmax, scale_bound, y_hat=None, want_starts=True):
    """Tables of rows (pix[i], chan[j]) -> (starts int16-viewed-u16 [rows][L] or None, start_freq int32 [rows][2] or None)."""
    for t in (sigma, mu, logits):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape[0] == 1 and t.shape[1] == K * M):
            raise RuntimeError("masic_amd.codec: head outputs must be contiguous float32 [1, K*M, h, w] device tensors")
    HW = sigma.shape[-2] * sigma.shape[-1]
    rows, L = pix.numel() * chan.numel(), 2 * minmax + 1
    dev = sigma.device
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    starts = torch.empty((rows, L), dtype=torch.int16, device=dev) if want_starts else None
    sf = torch.empty((rows, 2), dtype=torch.int32, device=dev) if y_hat is not None else None
    if rows:
        check(lib.masic_gmm_cdf_rows(_p(sigma), _p(mu), _p(logits), M, K, HW, _p(pix), pix.numel(), _p(chan), chan.numel(), int(minmax),
                                     float(scale_bound), _p(y_hat) if y_hat is not None else None,
                                     _p(starts) if starts is not None else None, _p(sf) if sf is not None else None, _p(err),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gmm_cdf_rows")
    return starts, sf, err


def check_err(err, what):
    e = int(err.item())
    if e & 1:
        raise RuntimeError(f"masic_amd.codec: {what}: a coding table could not be normalised to 2^16")
    if e & 2:
        raise RuntimeError(f"masic_amd.codec: {what}: a symbol lies outside the alphabet (minmax too small)")


def encode_freqs(start_freq):
    sf = np.ascontiguousarray(start_freq, dtype=np.int32).reshape(-1, 2)
    cap = (sf.shape[0] + 2) * 4
    out = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    check(lib.masic_rans_encode_freqs(sf.ctypes.data_as(ctypes.c_void_p), sf.shape[0], out.ctypes.data_as(ctypes.c_void_p), cap,
                                      ctypes.byref(n)), "rans_encode_freqs")
    return out[:n.value].tobytes()


class AdaptiveDecoder:
    """Incremental rANS decoder over per-symbol tables (masic_rans_decoder_*)."""

    def __init__(self, data):
        self._buf = np.frombuffer(bytes(data), dtype=np.uint8)
        self._h = ctypes.c_void_p()
        check(lib.masic_rans_decoder_open(self._buf.ctypes.data_as(ctypes.c_void_p), self._buf.size, ctypes.byref(self._h)), "rans_decoder_open")

    def decode_rows(self, starts_u16):
        s = np.ascontiguousarray(starts_u16)
        out = np.empty(s.shape[0], dtype=np.int32)
        check(lib.masic_rans_decoder_decode_rows(self._h, s.ctypes.data_as(ctypes.c_void_p), s.shape[0], s.shape[1],
                                                 out.ctypes.data_as(ctypes.c_void_p)), "rans_decoder_decode_rows")
        return out

    def close(self):
        if self._h:
            lib.masic_rans_decoder_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        self.close()


def encode_view(params_fn, y_hat, M, K, chan, minmax, scale_bound):
    """One pass: parameters of the full latent -> (start, freq) of every coded symbol in wavefront order -> rANS bytes."""
    h, w = y_hat.shape[-2:]
    if chan.numel() == 0:
        return encode_freqs(np.zeros((0, 2), dtype=np.int32))
    sigma, mu, logits = params_fn(y_hat)
    pix = torch.from_numpy(np.concatenate(wavefront_steps(h, w))).to(y_hat.device)
    _, sf, err = gmm_tables(sigma, mu, logits, M, K, pix, chan, minmax, scale_bound, y_hat=y_hat, want_starts=False)
    check_err(err, "encode")
    return encode_freqs(sf.cpu().numpy())


def decode_view(params_fn, data, shape, M, K, chan, minmax, scale_bound, device, use_graph=True):
    """Wavefront by wavefront: parameters from what is decoded so far -> tables of the wavefront -> symbols -> latent.
    The device side of a step (context convolution, nine head layers on three streams, table kernel: ~17 launches of a few
    microseconds) is captured once into a HIP graph over static buffers -- the latent, a wavefront list padded to h entries,
    the table rows -- and replayed per wavefront; issued eagerly the host side of those launches is 60 % of the decode time."""
    h, w = shape
    y_hat = torch.zeros((1, M, h, w), dtype=torch.float32, device=device)
    nch, L = chan.numel(), 2 * minmax + 1
    if nch == 0:
        return y_hat
    dec = AdaptiveDecoder(data)
    flat = y_hat.view(M, h * w)
    chan_l = chan.long()
    pix_buf = torch.full((h,), -1, dtype=torch.int32, device=device)          # a wavefront holds at most one pixel per row
    starts = torch.empty((h * nch, L), dtype=torch.int16, device=device)
    err = torch.zeros(1, dtype=torch.int32, device=device)
    pix_host = torch.full((h,), -1, dtype=torch.int32).pin_memory()
    starts_host = torch.empty((h * nch, L), dtype=torch.int16).pin_memory()
    val_host = torch.empty((h, nch), dtype=torch.float32).pin_memory()
    val_dev = torch.empty((h, nch), dtype=torch.float32, device=device)

    def device_step():
        sigma, mu, logits = params_fn(y_hat)
        for t in (sigma, mu, logits):
            if not (t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == (1, K * M, h, w)):
                raise RuntimeError("masic_amd.codec: head outputs must be contiguous float32 [1, K*M, h, w] device tensors")
        check(lib.masic_gmm_cdf_rows(_p(sigma), _p(mu), _p(logits), M, K, h * w, _p(pix_buf), h, _p(chan), nch, int(minmax), float(scale_bound),
                                     None, _p(starts), None, _p(err), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gmm_cdf_rows")

    graph = None
    cur = torch.cuda.current_stream()
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            device_step()                                     # weight packs and allocator warm before the capture
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            device_step()
    try:
        for pix_np in wavefront_steps(h, w):
            n = pix_np.size
            pix_host.fill_(-1)
            pix_host[:n] = torch.from_numpy(pix_np)
            pix_buf.copy_(pix_host, non_blocking=True)
            if graph is not None:
                graph.replay()
            else:
                device_step()
            starts_host[:n * nch].copy_(starts[:n * nch], non_blocking=True)
            cur.synchronize()
            sym = dec.decode_rows(starts_host[:n * nch].numpy().view(np.uint16))
            val_host[:n] = torch.from_numpy((sym - minmax).astype(np.float32).reshape(n, nch))
            val_dev[:n].copy_(val_host[:n], non_blocking=True)
            flat[chan_l[None, :], pix_buf[:n].long()[:, None]] = val_dev[:n]
    finally:
        dec.close()
    check_err(err, "decode")
    return y_hat
