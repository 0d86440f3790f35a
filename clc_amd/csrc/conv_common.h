// conv_common.h — kernel parameters, gather helpers and the shared epilogues of the convolution kernels (conv_igemm.hip, conv_halo.hip).
// Included INSIDE an anonymous namespace of each translation unit (after common.h).
#pragma once


constexpr int BK = 32;   // K-tile (floats)
constexpr int LD = 36;   // LDS row stride (floats)
constexpr unsigned kOOB = 0x80000000u;  // >= num_records of every SRD -> load returns 0

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// 8 f32 -> 8 bf16 (round to nearest even: v_cvt_pk_bf16_f32), the operand format of v_mfma_f32_32x32x16_bf16
__device__ __forceinline__ bf16x8 pack_bf16(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo[0]; r[1] = (__bf16)lo[1]; r[2] = (__bf16)lo[2]; r[3] = (__bf16)lo[3];
  r[4] = (__bf16)hi[0]; r[5] = (__bf16)hi[1]; r[6] = (__bf16)hi[2]; r[7] = (__bf16)hi[3];
  return r;
}

struct ConvParams {
  const float* x; const float* w; const float* bias; float* y;
  const float* mul; const float* res; float* y_pre;
  int N, H, W, Cin, ldx;       // source tensor geometry
  int OH, OW, Cout, ldy;       // destination geometry (pre-shuffle)
  int ks, stride, pad, transposed, in_op, act, norm, shuffle, res_first;
  int ldm, ldr, ldp, ldw;
  float res_scale;
  int M;                       // rows per class (transposed&stride2: per parity class)
  int kc_tiles;                // ceil(Cin/32)
  unsigned x_bytes, w_bytes;
  const float* xs; int ldxs, xs_act, xs_pre; unsigned xs_bytes;   // fused activation backward on the gathered operand
  int vec_epi;                 // every epilogue operand is 16-B addressable per 4 channels -> float4 epilogue
  const float* w2; const float* bias2; int group_rows;   // rows [k * group_rows, (k + 1) * group_rows) use filter set k: w, w2, w3, w4
  const float* w3; const float* bias3; const float* w4; const float* bias4;
  int pre_deriv;               // y_pre <- act'(v) instead of v
  const float* res_gate; int ldg, rg_act, rg_pre;   // residual term *= act'(res_gate)
  const float* out_gate; int ldog, og_act, og_pre;   // whole result *= act'(out_gate)
  int xcd_map;                 // conv_igemm_dma2_kernel: workgroups that share a pixel tile run back to back on ONE XCD (see the kernel)
  int dma_place;               // conv_igemm_dma2_kernel: 1 = next tile's DMA pieces at the top of the iteration, 0 = between the MFMA groups
  int ksplit; float* partial;   // conv_igemm_dma2_kernel: K range split over `ksplit` workgroups per tile (blockIdx.z = class * ksplit + split), raw partial tiles to `partial`
  int reg_epi;                 // conv_igemm_dma2_kernel: per-wave register epilogue (epilogue_regs) instead of the C tile through LDS
  int batch_variant_ok;        // clc_conv_desc.batch_variant_ok
  int bf16;                    // reduced-precision mode for THIS launch: set by clc_conv2d for the LDS-tiled family on maps larger than 16x16 only
  int ablate;                  // CLC_TUNE_ABLATE (diagnostic builds of the timing only, results are WRONG): 1 = no MFMAs, 2 = no result stores, 4 = no operand DMA
};

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}

// LDS-DMA: 64 lanes x 16 B from per-lane byte offsets of the buffer into LDS at dst (wave-uniform) + 16 * lane.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc cannot type-check the LDS address-space cast)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, byte_off, 0, 0, 0);
#endif
}

// Block-uniform description of the live filter taps: a (nkh x nkw) grid kh = kh0 + step*j.
struct TapGrid { int kh0, kw0, step, nkh, nkw; };
__device__ __forceinline__ TapGrid make_taps(const ConvParams& p, int ph, int pw) {
  TapGrid t;
  if (p.transposed && p.stride == 2) {   // data-gradient of a stride-2 conv: only taps of the class's parity
    t.kh0 = (ph + p.pad) & 1; t.kw0 = (pw + p.pad) & 1; t.step = 2;
    t.nkh = (p.ks - t.kh0 + 1) >> 1; t.nkw = (p.ks - t.kw0 + 1) >> 1;
  } else {
    t.kh0 = t.kw0 = 0; t.step = 1; t.nkh = t.nkw = p.ks;
  }
  return t;
}

// Per-thread gather state for one A row (fixed over the K loop).
struct RowState { int base, y0, x0; bool ok; };
template <bool TR>
__device__ __forceinline__ RowState make_row(const ConvParams& p, int m, int DH, int DW, int ph, int pw) {
  RowState r;
  r.ok = m < p.M;
  const int mm = r.ok ? m : 0;
  const int n = mm / (DH * DW), q = mm - n * (DH * DW);
  int oy = q / DW, ox = q - oy * DW;
  if (TR) {
    if (p.stride == 2) { oy = 2 * oy + ph; ox = 2 * ox + pw; }
    r.y0 = oy + p.pad; r.x0 = ox + p.pad;
  } else {
    r.y0 = oy * p.stride - p.pad; r.x0 = ox * p.stride - p.pad;
  }
  r.base = n * p.H * p.W;
  return r;
}
// source pixel index of (row, tap kh/kw), or 0xFFFFFFFF when it falls outside the image / the row is past M
template <bool TR>
__device__ __forceinline__ unsigned a_pixel(const ConvParams& p, const RowState& r, int kh, int kw) {
  int iy, ix;
  bool ok = r.ok;
  if (TR) {
    const int sh = p.stride - 1;                 // parity already guaranteed by the tap grid
    const int ty = r.y0 - kh, tx = r.x0 - kw;
    iy = ty >> sh; ix = tx >> sh;
    ok = ok && ty >= 0 && tx >= 0;
  } else {
    iy = r.y0 + kh; ix = r.x0 + kw;
  }
  ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
  return ok ? (unsigned)(r.base + iy * p.W + ix) : 0xFFFFFFFFu;
}
__device__ __forceinline__ unsigned pix_off(unsigned pix, int ld, int c, bool c_ok) {
  return (pix != 0xFFFFFFFFu && c_ok) ? (pix * (unsigned)ld + (unsigned)c) * 4u : kOOB;
}
// byte offset of (row, tap kh/kw, channel c) in x, or kOOB
template <bool TR>
__device__ __forceinline__ unsigned a_offset(const ConvParams& p, const RowState& r, int kh, int kw, int c, bool c_ok) {
  int iy, ix;
  bool ok = r.ok && c_ok;
  if (TR) {
    const int sh = p.stride - 1;                 // parity already guaranteed by the tap grid
    const int ty = r.y0 - kh, tx = r.x0 - kw;
    iy = ty >> sh; ix = tx >> sh;
    ok = ok && ty >= 0 && tx >= 0;
  } else {
    iy = r.y0 + kh; ix = r.x0 + kw;
  }
  ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
  const unsigned off = ((unsigned)(r.base + iy * p.W + ix) * (unsigned)p.ldx + (unsigned)c) * 4u;
  return ok ? off : kOOB;
}

// The LEAN form of the same epilogue for the launches the Winograd kernels take most: bias, LeakyReLU / ReLU / none, a residual (before or after the
// activation), the saved pre-activation, the consumer-side LeakyReLU / ReLU gate — no GDN, no gated residual, no stored derivative.  Same operations
// in the same order on the same values as epilogue_math4 (same bits); the activation is two selects instead of a switch over every activation the
// ABI knows, which in the general form costs ~1.3 k cycles per pixel PAIR once several pixels per thread are unrolled behind each other.
__device__ __forceinline__ bool epilogue_is_lean(const ConvParams& p) {
  return p.norm == CLC_NORM_NONE && !p.res_gate && !p.pre_deriv && (p.act == CLC_ACT_NONE || p.act == CLC_ACT_LRELU || p.act == CLC_ACT_RELU) &&
         (!p.out_gate || p.og_act == CLC_ACT_LRELU || p.og_act == CLC_ACT_RELU);
}
__device__ __forceinline__ float lean_act(float v, int act) {
  return act == CLC_ACT_NONE ? v : (v > 0.f ? v : (act == CLC_ACT_LRELU ? 0.01f * v : 0.f));
}
struct Lean4In { f32x4 rr, og; };
__device__ __forceinline__ Lean4In lean_load4(const ConvParams& p, size_t pix, int co) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  Lean4In in;
  in.rr = p.res ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + co) : z;
  in.og = p.out_gate ? *reinterpret_cast<const f32x4*>(p.out_gate + pix * p.ldog + co) : z;
  return in;
}
__device__ __forceinline__ void lean_finish4(const ConvParams& p, f32x4 bv, f32x4 acc, const Lean4In& in, size_t pix, int co) {
  f32x4 v = acc;
  v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
  f32x4 rv = {0.f, 0.f, 0.f, 0.f};
  if (p.res) {
    rv = p.res_scale * in.rr;
    if (p.res_first) v = v + rv;
  }
  if (p.y_pre) *reinterpret_cast<f32x4*>(p.y_pre + pix * p.ldp + co) = v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = lean_act(v[q], p.act);
  if (p.res && !p.res_first) v = v + rv;
  if (p.out_gate) {
    const float gs = p.og_act == CLC_ACT_LRELU ? 0.01f : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = v[q] * (in.og[q] > 0.f ? 1.f : gs);
  }
  *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + co) = v;
}
// ... and its PixelShuffle(2) form: channels co .. co + 3 of pixel (oy, ox) are the 2 x 2 sub-pixels of output channel co / 4
__device__ __forceinline__ void lean_store_shuffle4(const ConvParams& p, f32x4 bv, f32x4 acc, int n, int oy, int ox, int co) {
  const int ch = co >> 2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const size_t pix = (size_t)(n * 2 * p.OH + 2 * oy + (e >> 1)) * (2 * p.OW) + 2 * ox + (e & 1);
    float v = acc[e] + bv[e];
    float rterm = 0.f;
    if (p.res) rterm = p.res_scale * p.res[pix * p.ldr + ch];
    if (p.res && p.res_first) v += rterm;
    if (p.y_pre) p.y_pre[pix * p.ldp + ch] = v;
    v = lean_act(v, p.act);
    if (p.res && !p.res_first) v += rterm;
    if (p.out_gate) v *= (p.out_gate[pix * p.ldog + ch] > 0.f ? 1.f : (p.og_act == CLC_ACT_LRELU ? 0.01f : 0.f));
    p.y[pix * p.ldy + ch] = v;
  }
}

// Shared epilogue for one accumulator element.
__device__ __forceinline__ void epilogue_store(const ConvParams& p, float acc, float bv, int m, int co, int DH, int DW, int ph, int pw) {
  size_t pix; int ch = co;
  if (p.transposed && p.stride == 2) {
    const int n = m / (DH * DW), rr = m - n * (DH * DW);
    const int oy = rr / DW, ox = rr - oy * DW;
    pix = (size_t)(n * p.OH + 2 * oy + ph) * p.OW + 2 * ox + pw;
  } else if (p.shuffle) {
    const int n = m / (p.OH * p.OW), rr = m - n * (p.OH * p.OW);
    const int oy = rr / p.OW, ox = rr - oy * p.OW;
    ch = co >> 2;
    pix = (size_t)(n * 2 * p.OH + 2 * oy + ((co >> 1) & 1)) * (2 * p.OW) + 2 * ox + (co & 1);
  } else {
    pix = (size_t)m;
  }
  float v = acc + bv;
  float rterm = 0.f;
  if (epilogue_is_lean(p)) {   // (same operations, two selects instead of the activation switches: see epilogue_is_lean)
    if (p.res) rterm = p.res_scale * p.res[pix * p.ldr + ch];
    if (p.res && p.res_first) v += rterm;
    if (p.y_pre) p.y_pre[pix * p.ldp + ch] = v;
    v = lean_act(v, p.act);
    if (p.res && !p.res_first) v += rterm;
    if (p.out_gate) v *= (p.out_gate[pix * p.ldog + ch] > 0.f ? 1.f : (p.og_act == CLC_ACT_LRELU ? 0.01f : 0.f));
    p.y[pix * p.ldy + ch] = v;
    return;
  }
  if (p.res) {
    rterm = p.res_scale * p.res[pix * p.ldr + ch];
    if (p.res_gate) rterm *= act_deriv(p.res_gate[pix * p.ldg + ch], p.rg_act, p.rg_pre);
  }
  if (p.res && p.res_first) v += rterm;
  if (p.y_pre) p.y_pre[pix * p.ldp + ch] = p.pre_deriv ? act_deriv(v, p.act, 1) : v;
  if (p.norm != CLC_NORM_NONE) {
    const float mv = p.mul[pix * p.ldm + ch];
    v = (p.norm == CLC_NORM_GDN) ? mv * rsqrtf(v) : ((p.norm == CLC_NORM_IGDN) ? mv * sqrtf(v) : 2.f * (mv * v));
  }
  v = apply_act(v, p.act);
  if (p.res && !p.res_first) v += rterm;
  if (p.out_gate) v *= act_deriv(p.out_gate[pix * p.ldog + ch], p.og_act, p.og_pre);
  p.y[pix * p.ldy + ch] = v;
}

// Same arithmetic, element for element, on 4 consecutive channels of one pixel (p.vec_epi: no shuffle, Cout % 4 == 0,
// every row stride a multiple of 4 floats and every base 16-B aligned): b128 loads / stores instead of dword ones.
// epilogue_math4: the arithmetic alone, on values already in registers (unused operands: anything).
struct Epi4 { f32x4 y, pre; };
__device__ __forceinline__ Epi4 epilogue_math4(const ConvParams& p, f32x4 bv, f32x4 acc, f32x4 res_raw, f32x4 rg_raw, f32x4 mul_raw, f32x4 og_raw) {
  f32x4 v = acc;
  v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
  f32x4 rv = {0.f, 0.f, 0.f, 0.f};
  if (p.res) {
    rv = p.res_scale * res_raw;
    if (p.res_gate) rv = rv * act_deriv4(rg_raw, p.rg_act, p.rg_pre);
  }
  if (p.res && p.res_first) v = v + rv;
  Epi4 o;
  o.pre = p.pre_deriv ? act_deriv4(v, p.act, 1) : v;
  if (p.norm != CLC_NORM_NONE) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (p.norm == CLC_NORM_GDN) ? mul_raw[q] * rsqrtf(v[q]) : ((p.norm == CLC_NORM_IGDN) ? mul_raw[q] * sqrtf(v[q]) : 2.f * (mul_raw[q] * v[q]));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = apply_act(v[q], p.act);
  if (p.res && !p.res_first) v = v + rv;
  if (p.out_gate) v = v * act_deriv4(og_raw, p.og_act, p.og_pre);
  o.y = v;
  return o;
}
__device__ __forceinline__ size_t epilogue_pixel(const ConvParams& p, int m, int DH, int DW, int ph, int pw) {
  if (p.transposed && p.stride == 2) {
    const int n = m / (DH * DW), rr = m - n * (DH * DW);
    const int oy = rr / DW, ox = rr - oy * DW;
    return (size_t)(n * p.OH + 2 * oy + ph) * p.OW + 2 * ox + pw;
  }
  return (size_t)m;
}
// (in three steps, so that a caller with several pixels per thread can issue every load before the first store: on gfx9 stores count in vmcnt
//  too, and a load behind a store waits for the store's write acknowledgement)
struct Epi4In { f32x4 rr, rg, mv, og; };
__device__ __forceinline__ f32x4 epilogue_bias4(const float* bias, int co) {
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) { bv[0] = bias[co]; bv[1] = bias[co + 1]; bv[2] = bias[co + 2]; bv[3] = bias[co + 3]; }
  return bv;
}
__device__ __forceinline__ Epi4In epilogue_load4(const ConvParams& p, size_t pix, int co) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  Epi4In in;
  in.rr = p.res ? *reinterpret_cast<const f32x4*>(p.res + pix * p.ldr + co) : z;
  in.rg = (p.res && p.res_gate) ? *reinterpret_cast<const f32x4*>(p.res_gate + pix * p.ldg + co) : z;
  in.mv = p.norm != CLC_NORM_NONE ? *reinterpret_cast<const f32x4*>(p.mul + pix * p.ldm + co) : z;
  in.og = p.out_gate ? *reinterpret_cast<const f32x4*>(p.out_gate + pix * p.ldog + co) : z;
  return in;
}
__device__ __forceinline__ void epilogue_finish4(const ConvParams& p, f32x4 bv, f32x4 acc, const Epi4In& in, size_t pix, int co) {
  const Epi4 o = epilogue_math4(p, bv, acc, in.rr, in.rg, in.mv, in.og);
  if (p.y_pre) *reinterpret_cast<f32x4*>(p.y_pre + pix * p.ldp + co) = o.pre;
  *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + co) = o.y;
}
__device__ __forceinline__ void epilogue_store4(const ConvParams& p, const float* bias, f32x4 acc, int m, int co, int DH, int DW, int ph, int pw) {
  const size_t pix = epilogue_pixel(p, m, DH, DW, ph, pw);
  const f32x4 bv = epilogue_bias4(bias, co);
  if (epilogue_is_lean(p)) {
    const Lean4In in = lean_load4(p, pix, co);
    lean_finish4(p, bv, acc, in, pix, co);
    return;
  }
  const Epi4In in = epilogue_load4(p, pix, co);
  epilogue_finish4(p, bv, acc, in, pix, co);
}

// Per-WAVE epilogue of one 32 x 32 accumulator block straight from the registers: lane (col = lane & 31, h = lane >> 5) holds rows
// (r&3) + 8 (r>>2) + 4 h, r = 0..15.  Same arithmetic per element as epilogue_store; organised in whole-block phases (one uniform
// branch per phase instead of one per element), addresses = one 32-bit lane offset per tensor + a scalar row offset (SRD buffer
// accesses, 16 dword stores of two 128-B row pieces each), the activation chosen once per block.  Rows are output pixels row0 + ...
// (no PixelShuffle / stride-2 data-gradient pixel mapping: the callers check), every tensor addressable with 32-bit byte offsets.
struct RegEpi {
  __amdgpu_buffer_rsrc_t y_r, pre_r, res_r, rg_r, mul_r, og_r;
  bool has_rg, has_mul;
};
__device__ __forceinline__ RegEpi make_reg_epi(const ConvParams& p) {
  auto srd = [](const float* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, 0x7FFFFFFF, 0x00020000); };
  RegEpi e;
  e.has_rg = p.res && p.res_gate;
  e.has_mul = p.norm != CLC_NORM_NONE;
  e.y_r = srd(p.y); e.pre_r = srd(p.y_pre ? p.y_pre : p.y); e.res_r = srd(p.res ? p.res : p.y);
  e.rg_r = srd(e.has_rg ? p.res_gate : p.y); e.mul_r = srd(e.has_mul ? p.mul : p.y); e.og_r = srd(p.out_gate ? p.out_gate : p.y);
  return e;
}
#define CLC_ROWIDX(r) (((r) & 3) + 8 * ((r) >> 2))
#define CLC_ROWOFF(r, ld) ((unsigned)((CLC_ROWIDX(r) + (CLC_ROWIDX(r) >= 16 ? jump : 0u)) * (unsigned)(ld)) * 4u)
// jump: extra pixels between rows 15 and 16 of the block (conv_halo.hip: a wave's 32 rows are two 16-pixel pieces of consecutive image rows,
// jump = W - 16); 0 for 32 consecutive pixels.
__device__ __forceinline__ void epilogue_regs(const ConvParams& p, const RegEpi& e, const f32x16& acc, float bv, unsigned row0 /* incl. 4 h */, int co, unsigned jump = 0u) {
  auto ld32 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
  auto st32 = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0); };
  const bool has_rg = e.has_rg, has_mul = e.has_mul;
  float v[16], rt[16], tq[16];
  if (p.res) {   // the residual operand first: its latency runs under the bias / address arithmetic
    const unsigned o = (row0 * (unsigned)p.ldr + (unsigned)co) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = ld32(e.res_r, o, CLC_ROWOFF(r, p.ldr));
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
  if (p.res) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rt[r] = p.res_scale * rt[r];
    if (has_rg) {
      const unsigned o = (row0 * (unsigned)p.ldg + (unsigned)co) * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) tq[r] = ld32(e.rg_r, o, CLC_ROWOFF(r, p.ldg));
#pragma unroll
      for (int r = 0; r < 16; ++r) rt[r] *= act_deriv(tq[r], p.rg_act, p.rg_pre);
    }
    if (p.res_first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += rt[r];
    }
  }
  bool act_done = false;
  if (p.y_pre) {
    const unsigned o = (row0 * (unsigned)p.ldp + (unsigned)co) * 4u;
    if (p.pre_deriv && p.act == CLC_ACT_GELU && !has_mul) {
      // fc1 of the Swin MLPs: gelu'(v) is stored for the backward pass and gelu(v) is the result — one evaluation of the
      // erf / exp parts for both (the same expressions gelu_f / gelu_grad_f evaluate)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {   // (two values per packed-f32 instruction: gelu_parts2, same bits as gelu_parts)
        f32x2 cdf, pdf;
        const f32x2 vv = {v[r], v[r + 1]};
        gelu_parts2(vv, cdf, pdf);
        const f32x2 dd = cdf + vv * pdf, gg = vv * cdf;
        st32(dd[0], e.pre_r, o, CLC_ROWOFF(r, p.ldp));
        st32(dd[1], e.pre_r, o, CLC_ROWOFF(r + 1, p.ldp));
        v[r] = gg[0]; v[r + 1] = gg[1];
      }
      act_done = true;
    } else if (p.pre_deriv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st32(act_deriv(v[r], p.act, 1), e.pre_r, o, CLC_ROWOFF(r, p.ldp));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) st32(v[r], e.pre_r, o, CLC_ROWOFF(r, p.ldp));
    }
  }
  if (has_mul) {
    const unsigned o = (row0 * (unsigned)p.ldm + (unsigned)co) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) tq[r] = ld32(e.mul_r, o, CLC_ROWOFF(r, p.ldm));
#pragma unroll
    for (int r = 0; r < 16; ++r)
      v[r] = (p.norm == CLC_NORM_GDN) ? tq[r] * rsqrtf(v[r]) : ((p.norm == CLC_NORM_IGDN) ? tq[r] * sqrtf(v[r]) : 2.f * (tq[r] * v[r]));
  }
  switch (act_done ? CLC_ACT_NONE : p.act) {   // (the same functions apply_act dispatches to)
    case CLC_ACT_LRELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
      break;
    case CLC_ACT_RELU:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      break;
    case CLC_ACT_GELU:
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 cdf, pdf;
        const f32x2 vv = {v[r], v[r + 1]};
        gelu_parts2(vv, cdf, pdf);
        const f32x2 gg = vv * cdf;
        v[r] = gg[0]; v[r + 1] = gg[1];
      }
      break;
    case CLC_ACT_NONE: break;
    default:
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], p.act);
  }
  if (p.res && !p.res_first) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += rt[r];
  }
  if (p.out_gate) {
    const unsigned o = (row0 * (unsigned)p.ldog + (unsigned)co) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) tq[r] = ld32(e.og_r, o, CLC_ROWOFF(r, p.ldog));
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] *= act_deriv(tq[r], p.og_act, p.og_pre);
  }
  {
    const unsigned o = (row0 * (unsigned)p.ldy + (unsigned)co) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) st32(v[r], e.y_r, o, CLC_ROWOFF(r, p.ldy));
  }
}
#undef CLC_ROWOFF
#undef CLC_ROWIDX
// host side: may this launch use epilogue_regs?  (32-bit byte offsets into every epilogue tensor, whole tiles, plain pixel rows)
static bool reg_epi_ok(const ConvParams& p, int BM, int BN) {
  if (!p.vec_epi || p.shuffle || (p.transposed && p.stride == 2) || p.M % BM || p.Cout % BN) return false;
  int ldmax = p.ldy;
  if (p.y_pre && p.ldp > ldmax) ldmax = p.ldp;
  if (p.res && p.ldr > ldmax) ldmax = p.ldr;
  if (p.res_gate && p.ldg > ldmax) ldmax = p.ldg;
  if (p.mul && p.ldm > ldmax) ldmax = p.ldm;
  if (p.out_gate && p.ldog > ldmax) ldmax = p.ldog;
  return (size_t)p.M * (size_t)ldmax * 4 < (1ull << 31);
}

