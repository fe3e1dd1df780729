"""Coding of the y latents for HSIC.compress / decompress (reference MASIC.py:855-1408).

The reference walks the latent in raster order and, per pixel, re-runs the masked context convolution on a 5x5 crop and
the nine 1x1 head layers, builds one CDF per non-zero channel on the host and calls a Python range coder per symbol.
Here:
  * the coding order is the anti-diagonal wavefront t = j + 3 i of the type-A 5x5 mask (pixel (i, j) needs rows i-2, i-1 up
    to column j+2 and the two pixels left of it: all have a smaller t), so a 32x32 latent takes 125 dependent steps
    instead of 1024, each coding all pixels of its wavefront x all non-zero channels at once;
  * the tables are built on the device (masic_gmm_cdf_rows) and only cross to the host as u16 rows for the decoder, or
    as (start, freq) pairs for the encoder;
  * the coder is the library's rANS (masic_rans_*) with ONE STREAM PER LATENT CHANNEL: a stream is one serial dependency chain, the
    channels' chains are independent, so the decoder's symbol search runs on the device -- one wavefront per channel stream
    (masic_rans_decode_step) -- and writes the decoded values straight into the latent.  The whole loop over coding steps stays on the
    GPU: per step one HIP-graph replay (context convolution + nine head layers + table kernel + decode kernel), no table rows, symbols
    or synchronisation crossing PCIe until the last step (round 2 copied ~0.5 MB of u16 table rows to the host per step, searched
    them on one core and scattered the values back: two thirds of its 54 ms per 512 x 512 pair).
Encoder and decoder evaluate the SAME kernels on tensors of the same shape, and the masked taps of the context
convolution are exact zeros, so a not-yet-decoded neighbour (zero in the decoder, the true value in the encoder) never
changes a bit of the parameters of the pixel being coded: the encoder needs ONE pass over the full latent, the decoder
one pass per wavefront.
Container (.bin): b"MSR2", u8 precision id (0 f32 / 1 bf16 / 2 fp8 operands: the tables depend on it), u8 flags (bit 0: an
activation-scale table follows), 2 pad bytes, [u32 length + the fp8 mode's calibration table as JSON, masic_amd/fp8.py], then per
view u32 number of channel streams n, n x u32 stream lengths in bytes, the streams back to back (round 2's "MSR1" had one stream per view).  The .npz header keeps the reference's layout (MASIC.py:916-948)."""
import ctypes

import numpy as np
import torch

from ._lib import check, lib

MAGIC = b"MSR2"


import functools


@functools.lru_cache(maxsize=16)
def wavefront_steps(h, w):
    """Pixel indices (i * w + j), rows ascending, of each wavefront t = j + 3 i.  (A function of the latent's shape alone: cached -- the
    125 numpy selections of a 32 x 32 latent were 0.5 ms of every encode_view / decode_view call.  Callers must not write to the arrays.)"""
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
    if e & 4:
        raise RuntimeError(f"masic_amd.codec: {what}: a channel stream ended before its last symbol (truncated or foreign data)")


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


def encode_channels(start_freq, npix, nch):
    """(start, freq) rows [npix * nch][2] in coding order (pixel-major) -> u32 count, u32 lengths[nch], the nch rANS streams."""
    sf = np.ascontiguousarray(start_freq, dtype=np.int32).reshape(-1, 2)
    cap = (sf.shape[0] + 2 * max(nch, 1)) * 4 + 64
    out = np.empty(cap, dtype=np.uint8)
    lengths = np.zeros(max(nch, 1), dtype=np.uint32)
    n = ctypes.c_size_t(0)
    if nch:
        check(lib.masic_rans_encode_channels(sf.ctypes.data_as(ctypes.c_void_p), int(npix), int(nch), out.ctypes.data_as(ctypes.c_void_p), cap,
                                             lengths.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)), "rans_encode_channels")
    return np.array([nch], dtype=np.uint32).tobytes() + lengths[:nch].tobytes() + out[:n.value].tobytes()


def split_channels(blob):
    """Inverse of encode_channels' framing -> (list of per-channel byte strings, bytes consumed)."""
    nch = int(np.frombuffer(blob[:4], dtype=np.uint32)[0])
    lengths = np.frombuffer(blob[4:4 + 4 * nch], dtype=np.uint32)
    off, out = 4 + 4 * nch, []
    for n in lengths:
        out.append(blob[off:off + int(n)])
        off += int(n)
    return out, off


def encode_view(params_fn, y_hat, M, K, chan, minmax, scale_bound):
    """One pass: parameters of the full latent -> (start, freq) of every coded symbol in wavefront order -> one rANS stream per channel."""
    h, w = y_hat.shape[-2:]
    if chan.numel() == 0:
        return encode_channels(np.zeros((0, 2), dtype=np.int32), 0, 0)
    pix = torch.from_numpy(np.concatenate(wavefront_steps(h, w))).to(y_hat.device)
    if getattr(params_fn, "skinny", False):           # the decoder's own kernels, on all pixels at once (bit-identical parameters)
        params_fn.set_latent(y_hat)
        params_fn.run(pix, None, 0, pix.numel())
        sigma, mu, logits = params_fn.sigma, params_fn.mu, params_fn.logits
    else:
        sigma, mu, logits = params_fn(y_hat)
    _, sf, err = gmm_tables(sigma, mu, logits, M, K, pix, chan, minmax, scale_bound, y_hat=y_hat, want_starts=False)
    check_err(err, "encode")
    return encode_channels(sf.cpu().numpy(), pix.numel(), chan.numel())


_STEP_LISTS = {}


def _step_lists(h, w, device):
    """int32 [steps][h] on the device: the pixels of every coding wavefront (at most one per row), -1 padded; per shape and device, kept."""
    key = (h, w, str(device))
    t = _STEP_LISTS.get(key)
    if t is None:
        steps = wavefront_steps(h, w)
        pix_np = np.full((len(steps), h), -1, dtype=np.int32)
        for i, p in enumerate(steps):
            pix_np[i, :p.size] = p
        if len(_STEP_LISTS) >= 16:
            _STEP_LISTS.clear()
        t = _STEP_LISTS[key] = torch.from_numpy(pix_np.reshape(-1)).to(device)
    return t


def decode_view(params_fn, data, shape, M, K, chan, minmax, scale_bound, device, use_graph=True):
    """Coding step by coding step, entirely on the device: parameters from what is decoded so far -> tables of the step's pixels ->
    one wavefront per channel stream finds the symbols and writes them into the latent.  The device side of a step (context
    convolution, nine head layers on three streams, table kernel, decode kernel: ~18 launches of a few microseconds) is captured once
    into a HIP graph over static buffers and replayed w + 3 (h - 1) times back to back; the step index lives in device memory
    (incremented by the decode kernel), so the host neither feeds nor waits for anything until the last replay."""
    h, w = shape
    y_hat = torch.zeros((1, M, h, w), dtype=torch.float32, device=device)
    nch, L = chan.numel(), 2 * minmax + 1
    if nch == 0:
        return y_hat
    streams, _ = split_channels(data)
    if len(streams) != nch or any(len(s_) < 8 or len(s_) % 4 for s_ in streams):
        raise ValueError("masic_amd.codec: the y stream does not hold one rANS stream per coded channel")
    steps = wavefront_steps(h, w)
    pix_d = _step_lists(h, w, device)
    words = np.frombuffer(b"".join(streams), dtype=np.uint32)
    cnt = np.array([len(s_) // 4 for s_ in streams], dtype=np.uint32)
    off = np.concatenate(([0], np.cumsum(cnt)[:-1])).astype(np.uint32)
    state0 = words[off].astype(np.uint64) | (words[off + 1].astype(np.uint64) << np.uint64(32))
    dev_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(device)
    words_d, off_d, cnt_d = dev_i32(words), dev_i32(off), dev_i32(cnt)
    state_init = torch.from_numpy(state0.view(np.int64)).to(device)
    state_d = state_init.clone()
    pos_d = torch.full((nch,), 2, dtype=torch.int32, device=device)
    step_d = torch.zeros(1, dtype=torch.int32, device=device)
    done_d = torch.zeros(1, dtype=torch.int32, device=device)
    err = torch.zeros(1, dtype=torch.int32, device=device)
    starts = torch.empty((h * nch, L), dtype=torch.int16, device=device)

    skinny = getattr(params_fn, "skinny", False)
    y16 = params_fn.y16 if skinny else None

    def device_step():
        if skinny:                                            # the step's <= h pixels only (masic_amd/csrc/skinny.hip)
            params_fn.run(pix_d, step_d, h, h)
            sigma, mu, logits = params_fn.sigma, params_fn.mu, params_fn.logits
        else:
            sigma, mu, logits = params_fn(y_hat)
        for t in (sigma, mu, logits):
            if not (t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == (1, K * M, h, w)):
                raise RuntimeError("masic_amd.codec: head outputs must be contiguous float32 [1, K*M, h, w] device tensors")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.masic_gmm_cdf_rows_at(_p(sigma), _p(mu), _p(logits), M, K, h * w, _p(pix_d), _p(step_d), h, _p(chan), nch, int(minmax),
                                        float(scale_bound), _p(starts), _p(err), st), "gmm_cdf_rows_at")
        check(lib.masic_rans_decode_step(_p(words_d), _p(off_d), _p(cnt_d), _p(state_d), _p(pos_d), _p(starts), _p(pix_d), _p(step_d), h, _p(chan), nch,
                                         L, int(minmax), _p(y_hat), _p(y16) if y16 is not None else None, h * w, _p(err), _p(done_d), st), "rans_decode_step")

    def reset():
        y_hat.zero_()
        if y16 is not None:
            y16.zero_()
        state_d.copy_(state_init)
        pos_d.fill_(2)
        step_d.zero_()
        done_d.zero_()
        err.zero_()

    cur = torch.cuda.current_stream()
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            device_step()                                     # weight packs and allocator warm before the capture (advances the coder: reset below)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            device_step()
        reset()
        for _ in steps:
            graph.replay()
    else:
        for _ in steps:
            device_step()
    check_err(err, "decode")
    return y_hat
