"""ctypes binding of the plain-C oracle coder (oracle/rans_oracle.c). TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_rans.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rans_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        i32p, u8p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8)
        L.oracle_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.oracle_rans_encode.argtypes = [i32p, i32p, ctypes.c_long, i32p, ctypes.c_int, i32p, i32p, u8p, ctypes.c_long]
        L.oracle_rans_encode.restype = ctypes.c_long
        L.oracle_rans_decode.argtypes = [u8p, ctypes.c_long, i32p, ctypes.c_long, i32p, ctypes.c_int, i32p, i32p, i32p]
        L.oracle_rans_decode.restype = ctypes.c_long
        _lib = L
    return _lib


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(p.size + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p.size, precision,
                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    if rc != 0:
        raise ValueError(f"oracle_pmf_to_quantized_cdf rc={rc}")
    return out.astype(np.int64).tolist()


def encode(symbols, indexes, cdf, cdf_len, offset) -> bytes:
    s, sp = _i32(symbols)
    i, ip = _i32(indexes)
    c, cp = _i32(cdf)
    l, lp = _i32(cdf_len)
    o, op = _i32(offset)
    cap = lib().oracle_rans_encode(sp, ip, s.size, cp, c.shape[1], lp, op, None, 0)
    buf = np.zeros(cap, dtype=np.uint8)
    n = lib().oracle_rans_encode(sp, ip, s.size, cp, c.shape[1], lp, op, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap)
    if n < 0:
        raise RuntimeError(f"oracle_rans_encode rc={n}")
    return buf[:n].tobytes()


def decode(stream: bytes, indexes, cdf, cdf_len, offset):
    b = np.frombuffer(stream, dtype=np.uint8).copy()
    i, ip = _i32(indexes)
    c, cp = _i32(cdf)
    l, lp = _i32(cdf_len)
    o, op = _i32(offset)
    out = np.zeros(i.size, dtype=np.int32)
    n = lib().oracle_rans_decode(b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), b.size, ip, i.size, cp, c.shape[1], lp, op,
                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if n < 0:
        raise RuntimeError(f"oracle_rans_decode rc={n}")
    return out, n
