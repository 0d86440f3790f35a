/* rans_oracle.c — plain-C oracle for the integer (bit-exact) part of the CLC codec path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; never by clc_amd/.
 *
 * Restates the published algorithm of CompressAI's C++ entropy coder
 * (compressai/cpp_exts/rans/rans_interface.cpp over ryg_rans rans64.h) and of
 * compressai/cpp_exts/ops/ops.cpp:pmf_to_quantized_cdf — a third-party dependency
 * of the reference that is NOT under /root/reference and is un-pinned (SURVEY.md §8c,
 * Appendix A.4/A.5).  Reference call sites: /root/reference/models/CLC_run.py:658,
 * 712-713 (BufferedRansEncoder.encode_with_indexes / flush), :762-763,793
 * (RansDecoder.set_stream / decode_stream), :643-644,749 (EntropyBottleneck
 * compress/decompress).  PARITY UNPINNED against the real library; pinned against
 * oracle/rans_py.py (independent pure-Python implementation) and tests/golden/rans_kat.json.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS_VAL ((1 << BYPASS_PRECISION) - 1)
#define RANS64_L (1ull << 31)

typedef struct {
  uint16_t start, range;
  uint8_t bypass;
} sym_t;

/* returns 0 on success; cdf_out must hold n+1 entries */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
  for (int i = 0; i < n; ++i)
    if (pmf[i] < 0 || !isfinite(pmf[i])) return -1;
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
  uint32_t total = 0;
  for (int i = 0; i <= n; ++i) total += cdf[i];
  if (total == 0) return -2;
  for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
  for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;
  for (int i = 0; i < n; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best = -1;
      for (int j = 0; j < n; ++j) {
        uint32_t f = cdf[j + 1] - cdf[j];
        if (f > 1 && f < best_freq) { best_freq = f; best = j; }
      }
      if (best < 0) return -3;
      if (best < i) { for (int j = best + 1; j <= i; ++j) cdf[j]--; }
      else { for (int j = i + 1; j <= best; ++j) cdf[j]++; }
    }
  }
  return 0;
}

/* Encode: symbols/indexes of length n; cdfs is [n_cdf][cdf_stride] int32.
 * out must hold at least 4*(max_syms+2) bytes where max_syms bounds the expanded symbol
 * count (call with out==NULL to get the required byte capacity). Returns stream bytes (<0 error). */
long oracle_rans_encode(const int32_t *symbols, const int32_t *indexes, long n, const int32_t *cdfs, int cdf_stride,
                        const int32_t *cdf_sizes, const int32_t *offsets, uint8_t *out, long out_cap) {
  long cap = n * 12 + 16, ns = 0; /* worst case: 1 + unary(<=3 for 32-bit) + 8 nibbles */
  if (!out) return 4 * (cap + 2);
  sym_t *syms = (sym_t *)malloc(sizeof(sym_t) * (size_t)cap);
  if (!syms) return -1;
  for (long i = 0; i < n; ++i) {
    int32_t ci = indexes[i];
    const int32_t *cdf = cdfs + (long)ci * cdf_stride;
    int32_t max_value = cdf_sizes[ci] - 2;
    int32_t value = symbols[i] - offsets[ci];
    uint32_t raw = 0;
    if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
    else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
    syms[ns++] = (sym_t){(uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0};
    if (value == max_value) {
      int32_t nb = 0;
      while (nb < 8 && (raw >> (nb * BYPASS_PRECISION)) != 0) ++nb;
      int32_t val = nb;
      while (val >= MAX_BYPASS_VAL) { syms[ns++] = (sym_t){MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1}; val -= MAX_BYPASS_VAL; }
      syms[ns++] = (sym_t){(uint16_t)val, (uint16_t)(val + 1), 1};
      for (int32_t j = 0; j < nb; ++j) {
        int32_t v = (raw >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
        syms[ns++] = (sym_t){(uint16_t)v, (uint16_t)(v + 1), 1};
      }
    }
  }
  long nwords = ns + 2;
  uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nwords);
  if (!buf) { free(syms); return -1; }
  uint32_t *ptr = buf + nwords;
  uint64_t x = RANS64_L;
  while (ns > 0) {
    sym_t s = syms[--ns];
    if (!s.bypass) {
      uint64_t x_max = ((RANS64_L >> PRECISION) << 32) * s.range;
      if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
      x = ((x / s.range) << PRECISION) + (x % s.range) + s.start;
    } else {
      uint32_t freq = 1u << (16 - BYPASS_PRECISION);
      uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
      if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
      x = (x << BYPASS_PRECISION) | s.start;
    }
  }
  ptr -= 2;
  ptr[0] = (uint32_t)x;
  ptr[1] = (uint32_t)(x >> 32);
  long nbytes = (long)((buf + nwords) - ptr) * 4;
  long rc = nbytes;
  if (nbytes > out_cap) rc = -2; else memcpy(out, ptr, (size_t)nbytes);
  free(buf);
  free(syms);
  return rc;
}

typedef struct {
  const uint32_t *ptr, *end;
  uint64_t x;
} dec_t;

static inline void renorm(dec_t *d) {
  if (d->x < RANS64_L) { d->x = (d->x << 32) | (d->ptr < d->end ? *d->ptr : 0u); d->ptr++; }
}
static inline uint32_t get_bits(dec_t *d, int n) {
  uint32_t v = (uint32_t)(d->x & ((1u << n) - 1));
  d->x >>= n;
  renorm(d);
  return v;
}

/* Decode n symbols (indexes given) from stream; returns number of 32-bit words consumed (<0 error). */
long oracle_rans_decode(const uint8_t *stream, long nbytes, const int32_t *indexes, long n, const int32_t *cdfs,
                        int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
  if (nbytes < 8 || (nbytes & 3)) return -1;
  dec_t d;
  d.ptr = (const uint32_t *)stream;
  d.end = d.ptr + nbytes / 4;
  d.x = (uint64_t)d.ptr[0] | ((uint64_t)d.ptr[1] << 32);
  d.ptr += 2;
  for (long i = 0; i < n; ++i) {
    int32_t ci = indexes[i];
    const int32_t *cdf = cdfs + (long)ci * cdf_stride;
    int32_t size = cdf_sizes[ci], max_value = size - 2;
    uint32_t cf = (uint32_t)(d.x & ((1u << PRECISION) - 1));
    int32_t s = 0;
    while (s < size && !((uint32_t)cdf[s] > cf)) ++s;
    s -= 1;
    uint32_t start = (uint32_t)cdf[s], range = (uint32_t)(cdf[s + 1] - cdf[s]);
    d.x = (uint64_t)range * (d.x >> PRECISION) + (d.x & ((1u << PRECISION) - 1)) - start;
    renorm(&d);
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)get_bits(&d, BYPASS_PRECISION), nb = val;
      while (val == MAX_BYPASS_VAL) { val = (int32_t)get_bits(&d, BYPASS_PRECISION); nb += val; }
      uint32_t raw = 0;
      for (int32_t j = 0; j < nb; ++j) raw |= get_bits(&d, BYPASS_PRECISION) << (j * BYPASS_PRECISION);
      value = (int32_t)(raw >> 1);
      if (raw & 1) value = -value - 1; else value += max_value;
    }
    out[i] = value + offsets[ci];
  }
  return (long)(d.ptr - (const uint32_t *)stream);
}
