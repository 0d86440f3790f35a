"""``compressai.ans``-compatible entropy coder backed by the C++ rANS of libclc_hip.so.

Same class / method names as the pybind11 module the reference imports
(/root/reference/models/CLC_run.py:2,658,712-713,762-763,793): ``BufferedRansEncoder``
(``encode_with_indexes``, ``flush``), ``RansEncoder`` (``encode_with_indexes``) and
``RansDecoder`` (``set_stream``, ``decode_stream``, ``decode_with_indexes``).  Arguments may be
Python lists (reference style) or int32 numpy arrays (zero-copy fast path used by clc_amd.models).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _tables(cdfs, cdfs_sizes, offsets):
    if isinstance(cdfs, np.ndarray) and cdfs.dtype == np.int32 and cdfs.ndim == 2 and cdfs.flags.c_contiguous:
        cdf = cdfs
    else:  # ragged python lists are allowed by the reference API: pad to a rectangle
        rows = [np.asarray(r, dtype=np.int32) for r in cdfs]
        width = max(len(r) for r in rows)
        cdf = np.zeros((len(rows), width), dtype=np.int32)
        for i, r in enumerate(rows):
            cdf[i, : len(r)] = r
    return cdf, _i32(cdfs_sizes).reshape(-1), _i32(offsets).reshape(-1)


def encode(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
    L = _lib.load()
    sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    if sym.size != idx.size:
        raise ValueError("symbols and indexes must have the same length")
    cdf, ln, off = _tables(cdfs, cdfs_sizes, offsets)
    if idx.size and (idx.min() < 0 or idx.max() >= cdf.shape[0]):
        raise ValueError("index out of range of the CDF table")
    cap = L.clc_rans_encode_bound(sym.size)
    buf = np.empty(cap, dtype=np.uint8)
    n = L.clc_rans_encode(sym.ctypes.data, idx.ctypes.data, sym.size, cdf.ctypes.data, cdf.shape[1], ln.ctypes.data, off.ctypes.data,
                          buf.ctypes.data, cap)
    _lib.check(n, "clc_rans_encode")
    return buf[:n].tobytes()


class _Decoder:
    def __init__(self, stream: bytes):
        L = _lib.load()
        b = np.frombuffer(stream, dtype=np.uint8)
        self._h = L.clc_rans_decoder_create(b.ctypes.data, b.size)
        if not self._h:
            raise _lib.ClcError(f"clc_rans_decoder_create failed: {L.clc_last_error().decode()}")
        self.words = 2

    def decode(self, indexes, cdfs, cdfs_sizes, offsets):
        L = _lib.load()
        idx = _i32(indexes).reshape(-1)
        cdf, ln, off = _tables(cdfs, cdfs_sizes, offsets)
        if idx.size and (idx.min() < 0 or idx.max() >= cdf.shape[0]):
            raise ValueError("index out of range of the CDF table")
        out = np.empty(idx.size, dtype=np.int32)
        n = L.clc_rans_decoder_decode(self._h, idx.ctypes.data, idx.size, cdf.ctypes.data, cdf.shape[1], ln.ctypes.data, off.ctypes.data,
                                      out.ctypes.data)
        _lib.check(n, "clc_rans_decoder_decode")
        self.words = n
        return out

    def close(self):
        if self._h:
            _lib.load().clc_rans_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode(stream: bytes, indexes, cdfs, cdfs_sizes, offsets) -> np.ndarray:
    d = _Decoder(stream)
    try:
        return d.decode(indexes, cdfs, cdfs_sizes, offsets)
    finally:
        d.close()


class BufferedRansEncoder:
    """Buffers (symbols, indexes) chunks; flush() encodes them as ONE stream (same bytes as the reference coder)."""

    def __init__(self):
        self._sym, self._idx, self._tables = [], [], None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        self._sym.append(_i32(symbols).reshape(-1))
        self._idx.append(_i32(indexes).reshape(-1))
        self._tables = (cdfs, cdfs_sizes, offsets)

    def flush(self) -> bytes:
        if self._tables is None:
            out = encode(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((1, 2), np.int32), [2], [0])
        else:
            out = encode(np.concatenate(self._sym), np.concatenate(self._idx), *self._tables)
        self._sym, self._idx, self._tables = [], [], None
        return out


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
        return encode(symbols, indexes, cdfs, cdfs_sizes, offsets)


class RansDecoder:
    def __init__(self):
        self._d = None

    def set_stream(self, stream: bytes):
        if self._d is not None:
            self._d.close()
        self._d = _Decoder(stream)

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        if self._d is None:
            raise ValueError("set_stream() must be called first")
        out = self._d.decode(indexes, cdfs, cdfs_sizes, offsets)
        return out if isinstance(indexes, np.ndarray) else out.tolist()

    def decode_with_indexes(self, stream, indexes, cdfs, cdfs_sizes, offsets):
        self.set_stream(stream)
        return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets)
