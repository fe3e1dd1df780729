"""`compressai.ans` (reference: the pybind11 module built from compressai/cpp_exts/rans/rans_interface.cpp:355-381):
the same three classes over the host-side coder of libmasic_hip.so (masic_amd/csrc/rans.hip), including the streaming
decode (`set_stream` / `decode_stream`, rans_interface.cpp:286-353)."""
from masic_amd import rans as _rans


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        return _rans.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)


class BufferedRansEncoder:
    """Collects (symbols, indexes) batches; `flush` codes them as one stream (the tables of the first call are used
    for all of them, as every caller in the reference passes the same ones)."""

    def __init__(self):
        self._sym, self._idx, self._tables = [], [], None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        self._sym.extend(int(s) for s in symbols)
        self._idx.extend(int(i) for i in indexes)
        if self._tables is None:
            self._tables = (cdfs, cdfs_sizes, offsets)

    def flush(self):
        if self._tables is None:
            raise RuntimeError("BufferedRansEncoder.flush: nothing was encoded")
        out = _rans.encode_with_indexes(self._sym, self._idx, *self._tables)
        self._sym, self._idx, self._tables = [], [], None
        return out


class RansDecoder:
    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        return _rans.decode_with_indexes(encoded, indexes, cdfs, cdfs_sizes, offsets).tolist()

    def set_stream(self, encoded):
        self._stream = _rans.StreamDecoder(encoded)

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        if getattr(self, "_stream", None) is None:
            raise RuntimeError("RansDecoder.decode_stream: call set_stream first")
        return self._stream.decode(indexes, cdfs, cdfs_sizes, offsets).tolist()
