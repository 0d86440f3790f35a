// Internal helpers shared by the HIP translation units of libclc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/clc_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void clc_set_error(const char* fmt, ...);

// Process-wide tuning switches (clc_set_tuning): A/B knobs that select between kernel variants computing the SAME bits.
enum { CLC_TUNE_DMA_LOOP = 0 /* 1: conv_igemm_dma_kernel, 2: conv_igemm_dma2_kernel (re-timed K loop) */, CLC_TUNE_WGRAD_STREAMK = 1,
       CLC_TUNE_DMA_PLACE = 2 /* dma2 kernel: 1 = DMA pieces at the top of the K iteration, 0 = between the MFMA groups */, CLC_TUNE_SK_HALF = 3 /* stream-K grids of one workgroup per CU */,
       CLC_TUNE_SPLITK_PIX = 4 /* per-image map size up to which the split-K conv family is used (>= 256) */, CLC_TUNE_SPLITK_MAXC = 5 /* ... for at most this many output channels */,
       CLC_TUNE_SPLITK_PF = 6 /* K-tiles in flight per wave of the 8-wave split-K conv kernel: 3 (one workgroup per CU) or 1 (two) */,
       CLC_TUNE_1X1_TILE = 7 /* large-map 1x1 convolutions with at most this many K-tiles use the 128x64 tile (0: off) */,
       CLC_TUNE_XCD_MAP = 8 /* dma2 conv kernel: channel tiles of a pixel tile back to back on one XCD: 0 off, 1 = 1x1 layers, 2 = all */,
       CLC_TUNE_WGRAD_DMA = 9 /* filter-gradient tile kernels: LDS-DMA staging for problems without operand arithmetic */,
       CLC_TUNE_REG_EPI = 10 /* conv_igemm_dma2_kernel: per-wave register epilogue where the launch qualifies */,
       CLC_TUNE_DGRAD_SPLITK = 11 /* data gradients on the 64x64 tile whose grid under-fills the chip: K split over 2-4 workgroups + finish launch */,
       CLC_TUNE_ABLATE = 12 /* diagnostic: conv_igemm_dma2_kernel without its MFMAs (1) / result stores (2) / operand DMA (4) — wrong results, timing only */,
       CLC_TUNE_P1X1 = 13 /* large-map 1x1 layers on the persistent pipelined kernel (conv_igemm_p1x1_kernel) */,
       CLC_TUNE_BF16 = 14 /* opt-in reduced-precision mode: bf16-in / f32-accumulate MFMA in the 3x3 convolutions, data- and filter-gradient kernels of maps larger than 16x16 */,
       CLC_TUNE_TILE256 = 15 /* 64-channel 3x3 layers on >= 131072 rows: 256 x 64 tiles (64 x 32 per wave) */,
       CLC_TUNE_ATTN_4B = 16 /* window attention: the N = head_dim products on 4-block 16x16x1 MFMAs; bit mask: 1 = head_dim 16, 2 = head_dim-8 backward, 4 = head_dim-8 forward (off: see winattn.hip) */,
       CLC_TUNE_HEAVY128 = 17 /* long-K 3x3 layers on <= 16x16 maps (the slice-parameter nets): 128x128 LDS tiles with the K range split to fill the chip */,
       CLC_TUNE_ATTN_SPLIT = 18 /* attention backward on small grids (<= 1024 workgroups): two workgroups per window group, one tile each */,
       CLC_TUNE_MLP_PK = 19 /* fused Swin MLP forward: GELU on the packed-f32 VALU instructions (same bits either way) */,
       CLC_TUNE_N16 = 20 /* <= 16 output channels on large maps (the 12-channel tail of g_s): 16-column MFMAs (v_mfma_f32_16x16x4_f32); ANOTHER summation order */,
       CLC_TUNE_LIN = 21 /* 128 -> 128 / 64 -> 64 1x1 layers on >= 32 768 rows: the wave-private persistent kernel with whole-line stores (fused_mlp.hip: lin_kernel; same bits) */,
       CLC_TUNE_HALO = 22 /* 3x3 / stride-1 layers with 128 input channels whose caller supplies the packed filter (clc_conv_desc.w_packed): the halo-resident barrier-free kernel (conv_halo.hip; same bits); bit mask: 1 = 128-input-channel layers, 2 = 64-input-channel layers */,
       CLC_TUNE_WINO = 23 /* 3x3 / stride-1 layers with 128 k input channels whose caller supplies the Winograd-transformed filter (clc_conv_desc.w_wino): conv_wino_kernel / conv_wino64_kernel (F(2x2, 3x3); ANOTHER summation order); bit mask: 1 = forward launches, 2 = data gradients, 4 = the 64-wide kernel for layers of 64 k channels, 8 = the 64-wide kernel on every layer (experiment) */,
       CLC_TUNE_SPLIT = 24 /* f32 products from three-way bf16 splits (six v_mfma_f32_32x32x16_bf16 per 16 k, f32 accumulate; error of an f32 multiply-add) instead of v_mfma_f32_32x32x2_f32 in the filter-gradient kernels: bit 0 = all-taps 3x3 kernels, bit 1 = tiled kernels */,
       CLC_TUNE_COUNT = 25 };
extern int clc_tuning[CLC_TUNE_COUNT];
// fused_mlp.hip: y = W x + b (+ res_scale * res) on the wave-private persistent kernel; 0 = shape not built (the caller falls through)
// conv_halo.hip: 0 = the launch does not qualify (conv_params: the ConvParams clc_conv2d filled)
int clc_conv_halo_launch(const void* conv_params, const float* wpk, hipStream_t st);
// conv_wino.hip: 0 = the launch does not qualify
int clc_conv_wino_launch(const void* conv_params, const float* u, hipStream_t st);
int clc_lin_launch(const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr, float res_scale, float* y, int ldy, long M, int Cin,
                   int Cout, hipStream_t st);

#define CLC_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      clc_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

#define CLC_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e_ = hipGetLastError();                                  \
    if (e_ != hipSuccess) {                                             \
      clc_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -2;                                                        \
    }                                                                   \
  } while (0)

// hipFuncSetAttribute is a per-DEVICE setting: a once-flag per (call site, device), so that a process driving several GPUs
// (nn.DataParallel replicas, /root/reference/train_CLC.py:472-473) opts every one of them in
struct PerDeviceOnce {
  bool done[32] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// GELU (exact-erf form, CLC_run.py's nn.GELU) on the hardware exp2 / rcp: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7
// absolute, i.e. ~1.5e-7 relative on gelu itself) — OCML's erff + expf cost ~60 VALU instructions per element and made the
// fc1+GELU epilogues (256 channels at 128x128) epilogue-bound.
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  // cdf = Phi(x) = 0.5 (1 + erf(x / sqrt 2)), pdf = phi(x) = exp(-x^2 / 2) / sqrt(2 pi)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float eh = exp2_fast(-0.72134752044448170368f * x * x);            // exp(-x^2/2) = exp(-z^2)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = 1.0f - poly * eh;                                     // erf(|x|/sqrt2), eh = exp(-z^2)
  cdf = 0.5f + copysignf(0.5f * erf_abs, x);
  pdf = 0.39894228040143267794f * eh;
}
// Two values at once on the packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per lane and
// issue slot): the same operations in the same order as gelu_parts, element for element -> the same bits, 21 instructions per pair
// instead of 2 x 19.  f32 MFMAs and VALU work do not overlap on gfx950 (the f32 matrix rate is the vector rate), so an epilogue's GELU
// is paid in full: fc1 of a Swin MLP spends a quarter of its MFMA time in it.
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 z = ax * 0.70710678118654752440f;
  const f32x2 d = __builtin_elementwise_fma((f32x2){0.3275911f, 0.3275911f}, z, (f32x2){1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 e = (-0.72134752044448170368f * x) * x;
  const f32x2 eh = {exp2_fast(e[0]), exp2_fast(e[1])};
  f32x2 q = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
  q = __builtin_elementwise_fma(t, q, (f32x2){1.421413741f, 1.421413741f});
  q = __builtin_elementwise_fma(t, q, (f32x2){-0.284496736f, -0.284496736f});
  q = __builtin_elementwise_fma(t, q, (f32x2){0.254829592f, 0.254829592f});
  const f32x2 poly = t * q;
  const f32x2 erf_abs = 1.0f - poly * eh;
  const f32x2 hh = 0.5f * erf_abs;
  cdf = (f32x2){0.5f + copysignf(hh[0], x[0]), 0.5f + copysignf(hh[1], x[1])};
  pdf = 0.39894228040143267794f * eh;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case CLC_ACT_LRELU: return v > 0.f ? v : 0.01f * v;
    case CLC_ACT_RELU: return v > 0.f ? v : 0.f;
    case CLC_ACT_GELU: return gelu_f(v);
    case CLC_ACT_HALFTANH: return 0.5f * tanhf(v);
    case CLC_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// d act(v)/dv given the saved tensor s (= v when use_pre, else the activation output)
__device__ __forceinline__ float act_deriv(float s, int act, int use_pre) {
  switch (act) {
    case CLC_ACT_LRELU: return s > 0.f ? 1.f : 0.01f;
    case CLC_ACT_RELU: return s > 0.f ? 1.f : 0.f;
    case CLC_ACT_GELU: return gelu_grad_f(s);
    case CLC_ACT_HALFTANH: {
      const float t = use_pre ? tanhf(s) : 2.f * s;
      return 0.5f * (1.f - t * t);
    }
    case CLC_ACT_SAVED_DERIV: return s;
    default: return 1.f;
  }
}
__device__ __forceinline__ f32x4 act_deriv4(f32x4 s, int act, int use_pre) {
  return (f32x4){act_deriv(s[0], act, use_pre), act_deriv(s[1], act, use_pre), act_deriv(s[2], act, use_pre), act_deriv(s[3], act, use_pre)};
}

// wave64 sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Building blocks of the WAVE-PRIVATE kernels (fused_mlp.hip: Swin MLP, LayerNorm + Linear, plain 1x1 layers; conv_igemm.hip: the 64-channel 3x3
// layers): a wave owns 32 pixels and all channels, swapped MFMA roles (A = filter rows from a resident LDS image, B = pixels from registers),
// whole-line stores through a 2 KB wave-private LDS scratch.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const float* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void st4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}

// Deposit a [rows][kcols] K-contiguous filter as a slot-swizzled [kcols / 32][rows][32] LDS image: the 16-B chunk c of row r of a
// K-tile sits in slot c ^ ((r >> 1) & 7).  One LDS-DMA wave-instruction = 8 rows x 128 B; instructions are dealt round-robin to the waves.
template <int NW>
__device__ __forceinline__ void fill_image(float* img, const float* src, int rows, int kcols, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t sr = srd(src, (unsigned)rows * (unsigned)kcols * 4u);
  const int per_kt = rows >> 3, n = (kcols >> 5) * per_kt;
  for (int ii = wave; ii < n; ii += NW) {
    const int kt = ii / per_kt, r0 = (ii - kt * per_kt) << 3;
    const int row = r0 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
    const unsigned off = ((unsigned)row * (unsigned)kcols + (unsigned)(kt * 32 + chunk * 4)) * 4u;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sr, (__attribute__((address_space(3))) void*)(img + ((kt * rows + r0) << 5)), 16, off, 0, 0, 0);
#endif
  }
}


// Store one 32-pixel x 32-channel block that a wave holds in the accumulator layout (lane (pixel li, half h): the 16-B quads q = 0..3 at
// channels 8 q + 4 h) as FULL 128-B lines.  Stored straight from that layout every instruction would write 32-B pieces of 32 different
// lines — measured on the plain 1x1 layers (conv_w1x1_kernel, built and removed in round 4): 1.7 TB/s where whole-line stores reach
// 3.5-4 TB/s, and the 268 MB of dh / g per 8 x 128 x 128 block made the backward kernel store-bound.  So the block goes through a 2 KB
// wave-private LDS scratch, 16 pixels at a time (chunk c of pixel p in slot c ^ (p & 7): conflict-free both ways), and comes back with
// 8 lanes per pixel.  No barrier: one wave's LDS operations execute in program order.
__device__ __forceinline__ void store_block_lines(float* scratch, const f32x4 (&v)[4], __amdgpu_buffer_rsrc_t dst, unsigned pix0, unsigned ld, unsigned ch0,
                                                  int lane, int li, int h) {
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if ((li >> 4) == rd) {
      float* row = scratch + ((li & 15) << 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(row + ((((2 * q + h) ^ (li & 7))) << 2)) = v[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the other lanes' quads are in the scratch (and the compiler keeps the order)
    f32x4 t[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pl = 8 * k + (lane >> 3), j = lane & 7;
      t[k] = *reinterpret_cast<const f32x4*>(scratch + (pl << 5) + ((j ^ (pl & 7)) << 2));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read out before the next round overwrites it
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pl = 8 * k + (lane >> 3), j = lane & 7;
      st4(t[k], dst, ((pix0 + (unsigned)(16 * rd + pl)) * ld + ch0 + (unsigned)(j * 4)) * 4u);
    }
  }
}



