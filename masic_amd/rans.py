"""numpy-level wrappers of the host-side entropy coder in libmasic_hip.so (masic_amd/csrc/rans.hip, include/masic_hip.h):
rANS encode / decode with per-symbol CDF indexes and the PMF -> quantised CDF routine -- the replacements of the
reference's pybind11 modules `compressai.ans` and `compressai._CXX`."""
import ctypes

import numpy as np

from ._lib import check, lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def pmf_to_quantized_cdf(pmf, precision=16):
    """float32 probabilities [n] -> uint32 CDF [n+1] with cdf[0] = 0, cdf[n] = 2^precision and every step >= 1."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32).reshape(-1))
    cdf = np.empty(p.size + 1, dtype=np.uint32)
    check(lib.masic_pmf_to_quantized_cdf(_ptr(p), p.size, int(precision), _ptr(cdf)), "pmf_to_quantized_cdf")
    return cdf


def _tables(cdfs, cdf_sizes, offsets):
    if isinstance(cdfs, np.ndarray) and cdfs.ndim == 2:
        table = _i32(cdfs)
    else:                                   # list of rows of different lengths (the reference's vector<vector<int>>)
        rows = [np.asarray(r, dtype=np.int32) for r in cdfs]
        table = np.zeros((len(rows), max(r.size for r in rows)), dtype=np.int32)
        for i, r in enumerate(rows):
            table[i, :r.size] = r
    sizes, offs = _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    if not (table.shape[0] == sizes.size == offs.size):
        raise ValueError("masic_amd.rans: cdfs, cdf_sizes and offsets disagree on the number of tables")
    return table, sizes, offs


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    """-> bytes (reference RansEncoder.encode_with_indexes, rans_interface.cpp:203-212)"""
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    if sym.size != idx.size:
        raise ValueError("masic_amd.rans: symbols and indexes differ in length")
    table, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    cap = lib.masic_rans_encode_bound(sym.size)
    out = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    check(lib.masic_rans_encode_with_indexes(_ptr(sym), _ptr(idx), sym.size, _ptr(table), table.shape[1], _ptr(sizes), _ptr(offs),
                                             table.shape[0], _ptr(out), cap, ctypes.byref(n)), "rans_encode_with_indexes")
    return out[:n.value].tobytes()


def decode_with_indexes(encoded, indexes, cdfs, cdf_sizes, offsets):
    """-> int32 array (reference RansDecoder.decode_with_indexes, rans_interface.cpp:214-283)"""
    idx = _i32(indexes).reshape(-1)
    table, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
    buf = np.frombuffer(bytes(encoded), dtype=np.uint8)
    out = np.empty(idx.size, dtype=np.int32)
    check(lib.masic_rans_decode_with_indexes(_ptr(buf), buf.size, _ptr(idx), idx.size, _ptr(table), table.shape[1], _ptr(sizes), _ptr(offs),
                                             table.shape[0], _ptr(out)), "rans_decode_with_indexes")
    return out


class StreamDecoder:
    """Incremental decode_with_indexes over one stream (reference RansDecoder.set_stream / decode_stream, rans_interface.cpp:286-353)."""

    def __init__(self, encoded):
        import ctypes
        self._buf = np.frombuffer(bytes(encoded), dtype=np.uint8)
        self._h = ctypes.c_void_p()
        check(lib.masic_rans_decoder_open(_ptr(self._buf), self._buf.size, ctypes.byref(self._h)), "rans_decoder_open")

    def decode(self, indexes, cdfs, cdf_sizes, offsets):
        idx = _i32(indexes).reshape(-1)
        table, sizes, offs = _tables(cdfs, cdf_sizes, offsets)
        out = np.empty(idx.size, dtype=np.int32)
        check(lib.masic_rans_decoder_decode_indexes(self._h, _ptr(idx), idx.size, _ptr(table), table.shape[1], _ptr(sizes), _ptr(offs), table.shape[0],
                                                    _ptr(out)), "rans_decoder_decode_indexes")
        return out

    def close(self):
        if self._h:
            lib.masic_rans_decoder_close(self._h)
            self._h = None

    def __del__(self):
        self.close()
