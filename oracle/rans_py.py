"""Pure-Python rANS64 coder + CDF quantiser — oracle (TEST INFRASTRUCTURE).

Restates the published algorithm of CompressAI's ``compressai.ans`` C++ extension
(over ryg_rans ``rans64.h``) that the reference calls at
/root/reference/models/CLC_run.py:658,712-713 (BufferedRansEncoder),
:762-763,793 (RansDecoder) and, through EntropyBottleneck.compress/decompress,
:643-644,749.  The dependency is NOT vendored and un-pinned → PARITY UNPINNED
(SURVEY.md §8c, Appendix A.4/A.5).  This file is deliberately an independent,
slow, literal implementation: the C oracle (rans_oracle.c) and the product's C++
coder (clc_amd/csrc/rans_host.cpp) are checked for bit-identity against it.
"""
from __future__ import annotations

import struct
from typing import List, Sequence

import numpy as np

PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS_VAL = (1 << BYPASS_PRECISION) - 1
RANS64_L = 1 << 31
MASK64 = (1 << 64) - 1


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    """A.4: float32 ``round(p * 2^prec)`` (half away from zero), renormalise with u64
    integer divide, prefix-sum, then steal frequency for zero-width bins."""
    p32 = np.asarray(pmf, dtype=np.float32)
    if np.any(p32 < 0) or not np.all(np.isfinite(p32)):
        raise ValueError("invalid pmf")
    scaled = p32 * np.float32(1 << precision)  # float32 product, as in C++ `p * (1 << precision)` on a float
    # std::round on float: half away from zero
    rounded = np.floor(np.abs(scaled).astype(np.float64) + 0.5) * np.sign(scaled)
    cdf = [0] + [int(v) & 0xFFFFFFFF for v in rounded]
    total = sum(cdf) & 0xFFFFFFFF
    if total == 0:
        raise ValueError("pmf sums to zero")
    cdf = [(((1 << precision) * c) // total) & 0xFFFFFFFF for c in cdf]
    acc = 0
    for i in range(len(cdf)):
        acc = (acc + cdf[i]) & 0xFFFFFFFF
        cdf[i] = acc
    cdf[-1] = 1 << precision
    n = len(cdf)
    for i in range(n - 1):
        if cdf[i] == cdf[i + 1]:
            best_freq, best_steal = 0xFFFFFFFF, -1
            for j in range(n - 1):
                freq = cdf[j + 1] - cdf[j]
                if freq > 1 and freq < best_freq:
                    best_freq, best_steal = freq, j
            assert best_steal != -1
            if best_steal < i:
                for j in range(best_steal + 1, i + 1):
                    cdf[j] -= 1
            else:
                assert best_steal > i
                for j in range(i + 1, best_steal + 1):
                    cdf[j] += 1
    assert cdf[0] == 0 and cdf[-1] == (1 << precision)
    for i in range(n - 1):
        assert cdf[i + 1] > cdf[i], "non-monotone cdf"
    return cdf


class BufferedRansEncoder:
    def __init__(self):
        self._syms = []  # (start, range, bypass)

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        assert len(symbols) == len(indexes)
        for sym, cdf_idx in zip(symbols, indexes):
            cdf = cdfs[cdf_idx]
            max_value = cdfs_sizes[cdf_idx] - 2
            value = int(sym) - offsets[cdf_idx]
            raw_val = 0
            if value < 0:
                raw_val = -2 * value - 1
                value = max_value
            elif value >= max_value:
                raw_val = 2 * (value - max_value)
                value = max_value
            self._syms.append((cdf[value] & 0xFFFF, (cdf[value + 1] - cdf[value]) & 0xFFFF, False))
            if value == max_value:
                n_bypass = 0
                while (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0:
                    n_bypass += 1
                val = n_bypass
                while val >= MAX_BYPASS_VAL:
                    self._syms.append((MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, True))
                    val -= MAX_BYPASS_VAL
                self._syms.append((val, val + 1, True))
                for j in range(n_bypass):
                    v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL
                    self._syms.append((v, v + 1, True))

    def flush(self) -> bytes:
        x = RANS64_L
        words = []  # emitted in reverse (each new word goes in FRONT of the stream)
        while self._syms:
            start, rng, bypass = self._syms.pop()
            if not bypass:
                x_max = ((RANS64_L >> PRECISION) << 32) * rng
                if x >= x_max:
                    words.append(x & 0xFFFFFFFF)
                    x >>= 32
                x = ((x // rng) << PRECISION) + (x % rng) + start
            else:
                freq = 1 << (16 - BYPASS_PRECISION)
                x_max = ((RANS64_L >> 16) << 32) * freq
                if x >= x_max:
                    words.append(x & 0xFFFFFFFF)
                    x >>= 32
                x = ((x << BYPASS_PRECISION) | start) & MASK64
        words.append((x >> 32) & 0xFFFFFFFF)
        words.append(x & 0xFFFFFFFF)
        words.reverse()
        return struct.pack(f"<{len(words)}I", *words)


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
        enc = BufferedRansEncoder()
        enc.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return enc.flush()


class RansDecoder:
    def set_stream(self, stream: bytes):
        assert len(stream) % 4 == 0 and len(stream) >= 8
        self._w = struct.unpack(f"<{len(stream) // 4}I", stream)
        self._x = self._w[0] | (self._w[1] << 32)
        self._p = 2

    def _renorm(self):
        if self._x < RANS64_L:
            self._x = (self._x << 32) | self._w[self._p]
            self._p += 1

    def _get_bits(self, n):
        val = self._x & ((1 << n) - 1)
        self._x >>= n
        self._renorm()
        return val

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        out = []
        for cdf_idx in indexes:
            cdf = cdfs[cdf_idx]
            max_value = cdfs_sizes[cdf_idx] - 2
            offset = offsets[cdf_idx]
            cum_freq = self._x & ((1 << PRECISION) - 1)
            s = 0
            n = cdfs_sizes[cdf_idx]
            while s < n and not (cdf[s] > cum_freq):
                s += 1
            s -= 1
            start, rng = cdf[s], cdf[s + 1] - cdf[s]
            self._x = rng * (self._x >> PRECISION) + (self._x & ((1 << PRECISION) - 1)) - start
            self._renorm()
            value = s
            if value == max_value:
                val = self._get_bits(BYPASS_PRECISION)
                n_bypass = val
                while val == MAX_BYPASS_VAL:
                    val = self._get_bits(BYPASS_PRECISION)
                    n_bypass += val
                raw_val = 0
                for j in range(n_bypass):
                    val = self._get_bits(BYPASS_PRECISION)
                    raw_val |= val << (j * BYPASS_PRECISION)
                value = raw_val >> 1
                if raw_val & 1:
                    value = -value - 1
                else:
                    value += max_value
            out.append(value + offset)
        return out

    def decode_with_indexes(self, stream, indexes, cdfs, cdfs_sizes, offsets):
        self.set_stream(stream)
        return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets)

    @property
    def words_consumed(self):
        return self._p
