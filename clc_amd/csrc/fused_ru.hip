// fused_ru.hip — ONE launch for a CompressAI ResidualUnit on the 16x16 latent maps of the slice loop (gfx950).
//
//   y = relu(x + W3 . relu(conv3x3(W2, relu(W1 . x + b1)) + b2) + b3)           N -> N/2 -> N/2 -> N channels, N = 128
// (AttentionBlock.ResidualUnit behind SWAtten's conv_a / conv_b, /root/reference/models/CLC_run.py:222-244; six of them per attention
// block, two attention blocks per slice, five slices).  As three launches of the split-K convolution family one unit of the stacked
// four-filter-set case (8 192 rows) measured 10.7 + 16.5 + 11.4 us forward for 0.87 GFLOP: each launch is a dispatch, one memory
// round trip and a 16-72-MFMA chain.  Here a workgroup (8 waves) owns an 8 x 4 pixel tile of one image:
//   stage 0  the 10 x 6 halo of x (60 rows, padded to 64) -> LDS by LDS-DMA (32 KB, source-swizzled 16-B slots)
//   stage 1  t1 = relu(W1 x + b1) on the 64 halo rows (the first 1x1 is recomputed on the halo: 1.9x of a cheap layer), zeroed outside
//            the image (the 3x3's zero padding), kept in LDS; K = 128 split over two wave groups, combined through LDS
//   stage 2  t2 = relu(conv3x3(t1) + b2) on the 32 pixels; K = 9 x 64 split over four wave groups (fixed-order combine)
//   stage 3  y  = relu(x + W3 t2 + b3); K = 64 split over two wave groups; x comes from the LDS tile
// The filters (212 KB per set) are read straight from L2 into MFMA fragments (one b128 per 4 MFMAs, issued ahead).  t1 and t2 are also
// written to HBM: the unchanged backward kernels (data / filter gradients of the three layers) take them as saved activations.
// Up to four filter sets on equal parts of the batch, like clc_conv2d.  Results depend on nothing but the image itself.
//
// The DATA GRADIENT of the unit is the same chain with the filters transposed (128 -> 64 -> 64 -> 128 again):
//   g3 = dy . [y > 0];  g2 = (W3^T g3) . [t2 > 0];  g1 = conv3x3^T(W2, g2) . [t1 > 0];  dx = W1^T g1 + g3
// so the BWD instantiation of the kernel differs only in its loaders / epilogues: the y tile rides next to the dy tile in LDS and gates
// it at the fragment read, the stage epilogues multiply by the saved activations' signs instead of adding a bias, the 3x3 gathers
// mirrored, and the identity branch's gradient (g3) is the stage-3 "residual".  g2 and g1 go to HBM for the filter-gradient kernels.
#include "common.h"

#include <type_traits>

namespace {

constexpr unsigned kOOB = 0x80000000u;

struct RUParams {
  const float* x; int ldx;
  float* t1; float* t2; float* y;          // [N,16,16,64] / [N,16,16,64] / [N,16,16,128], dense
  const float* w1[4]; const float* b1[4];  // [64][128], [64]
  const float* w2[4]; const float* b2[4];  // [64][9][64], [64]
  const float* w3[4]; const float* b3[4];  // [128][64], [128]
  // BWD: x = dy, w1 / w2 / w3 = the transposed filters of layers 3 / 2 / 1 ([64][128], [64][9][64], [128][64]), t1 / t2 / y = the outputs
  // g2 / g1 / dx, and the saved activations of the forward pass (dense):
  const float* sy; const float* st2; const float* st1;
  int N, per_set;                          // images, images per filter set
  int ablate;                              // CLC_TUNE_ABLATE (timing diagnostics, results WRONG): 2 = no result stores
  unsigned x_bytes;
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// LDS hand-over between wave groups: wait for this wave's LDS traffic only (a __syncthreads() would also drain the filter prefetch)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int rowidx(int r) { return (r & 3) + 8 * (r >> 2); }

template <bool BWD>
__global__ __launch_bounds__(512, 1) void ru_fused_kernel(const RUParams p) {
  constexpr int C = 128, M = 64, H = 16, W = 16, TW = 8, TH = 4, HW = TW + 2, HR = (TW + 2) * (TH + 2);   // HR = 60 halo rows
  constexpr int LDT = M + 4;                       // row stride of the t1 / t2 LDS images (floats)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                                // [64][128], 16-B slot c of row r at slot c ^ (r & 31)
  float* T1s = Xs + 64 * C;                        // [64][LDT]
  float* T2s = T1s + 64 * LDT;                     // [32][LDT]
  float* red = T2s + 32 * LDT;                     // accumulator hand-over between the K groups: up to 8192 floats
  float* Ys = red + 8192;                   // BWD: the saved output y on the same halo, same layout as Xs (the ReLU gate of dy)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int img = blockIdx.x >> 3, tile = blockIdx.x & 7;
  const int y0 = (tile >> 1) * TH, x0 = (tile & 1) * TW;
  const int fs = min(img / p.per_set, 3);
  const float* w1 = p.w1[fs]; const float* w2 = p.w2[fs]; const float* w3 = p.w3[fs];

  // ---- stage 0: halo tile of x -> LDS (4 DMA pieces of 1 KB = 2 rows per wave)
  {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 2 + (lane >> 5), slot = lane & 31, chunk = slot ^ (row & 31);
      const int hy = (row * 205) >> 11, hx = row - hy * HW;          // row / 10, row % 10 (row < 64)
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = row < HR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = ok ? ((unsigned)((img * H + iy) * W + ix) * (unsigned)p.ldx + (unsigned)chunk * 4u) * 4u : kOOB;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(Xs + (wave * 4 + i) * 256), 16, off, 0, 0, 0);
      if constexpr (BWD) {
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sy), 0, (unsigned)p.N * (H * W * C * 4u), 0x00020000);
        const unsigned offy = ok ? ((unsigned)((img * H + iy) * W + ix) * (unsigned)C + (unsigned)chunk * 4u) * 4u : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)(Ys + (wave * 4 + i) * 256), 16, offy, 0, 0, 0);
      }
#endif
    }
  }
  asm volatile("" ::: "memory");   // (the counted wait below relies on the tile requests being the OLDEST: nothing may be hoisted above them)
  // filter fragments of stages 1 and 3 (this wave's share) and the three bias values, requested while the tile is in flight.  (Requesting
  // stage 2's 18 fragments here as well measured SLOWER, 20.0 vs 18.1 us: each b128 request touches 32 filter rows = 32 cache lines, and
  // 30 of them per wave up front hold the texture path while the tile is still waiting behind them.)
  f32x4 bw1[8], bw3[4];
  float bv1 = 0.f, bv2 = 0.f, bv3 = 0.f;
  {
    const int ct = wave & 1, kh = wave >> 2, ct3 = wave & 3;
    if constexpr (!BWD) {
      bv1 = p.b1[fs][ct * 32 + li];
      bv2 = p.b2[fs][ct * 32 + li];
      bv3 = p.b3[fs][ct3 * 32 + li];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) bw1[t] = ldg4(w1 + (size_t)(ct * 32 + li) * C + 64 * kh + 8 * t + 4 * h);
#pragma unroll
    for (int t = 0; t < 4; ++t) bw3[t] = ldg4(w3 + (size_t)(ct3 * 32 + li) * M + 32 * kh + 8 * t + 4 * h);
  }
  // the tile (the oldest requests) has landed; the 12 filter requests (+ 3 bias values) may still be in flight
  if constexpr (BWD) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(15)\n\ts_barrier" ::: "memory");

  // ---- stage 1: t1 = relu(W1 x + b1) on 64 halo rows x 64 channels: wave -> (row tile rt, column tile ct, K half kh)
  {
    const int tw = wave & 3, rt = tw >> 1, ct = tw & 1, kh = wave >> 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row = rt * 32 + li;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int chunk = 16 * kh + 2 * t + h;
      f32x4 a = *reinterpret_cast<const f32x4*>(Xs + row * C + ((chunk ^ (row & 31)) << 2));
      if constexpr (BWD) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(Ys + row * C + ((chunk ^ (row & 31)) << 2));
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = yv[s] > 0.f ? a[s] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw1[t][s], acc, 0, 0, 0);
    }
    // the two K halves meet through LDS, and each of the two waves finishes HALF of the rows (registers 8 kh .. 8 kh + 7): every wave has
    // epilogue work, none waits idle
    auto give = [&](auto OTHER) {
#pragma unroll
      for (int r = 0; r < 8; ++r) red[((tw * 2 + kh) * 8 + r) * 64 + lane] = acc[decltype(OTHER)::value + r];
    };
    if (kh == 0) give(std::integral_constant<int, 8>{}); else give(std::integral_constant<int, 0>{});
    lds_barrier();
    auto finish = [&](auto MINE) {
      constexpr int R0 = decltype(MINE)::value;
      const int co = ct * 32 + li;
      int hrs[8]; bool in_img[8]; size_t pix[8];
      float sv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int hr = rt * 32 + rowidx(R0 + r) + 4 * h;
        const int hy = (hr * 205) >> 11, hx = hr - hy * HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        hrs[r] = hr;
        in_img[r] = hr < HR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        pix[r] = ((size_t)(img * H + iy) * W + ix) * M + co;
        if constexpr (BWD) sv[r] = in_img[r] ? p.st2[pix[r]] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float other = red[((tw * 2 + (kh ^ 1)) * 8 + r) * 64 + lane];
        float v = (kh == 0 ? acc[R0 + r] + other : other + acc[R0 + r]) + bv1;      // (half 0 + half 1, whoever adds)
        if constexpr (BWD) v = sv[r] > 0.f ? v : 0.f;
        else v = v > 0.f ? v : 0.f;
        v = in_img[r] ? v : 0.f;                                  // the 3x3's zero padding / rows past the halo
        T1s[hrs[r] * LDT + co] = v;
        const int hy = (hrs[r] * 205) >> 11, hx = hrs[r] - hy * HW;
        if (in_img[r] && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW && !(p.ablate & 2)) p.t1[pix[r]] = v;
      }
    };
    if (kh == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 8>{});
    lds_barrier();
  }
  // ---- stage 2: t2 = relu(conv3x3(t1) + b2) on 32 pixels x 64 channels: wave -> (column tile ct, K quarter kq of 72 groups of 8)
  {
    const int ct = wave & 1, kq = wave >> 1;
    const int py = li >> 3, px = li & 7;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* wrow = w2 + (size_t)(ct * 32 + li) * (9 * M) + 4 * h;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const int g = 18 * kq + j, tap = g >> 3, kh = (tap * 11) >> 5, kw = tap - 3 * kh;     // group g: tap g >> 3, channels 8 (g & 7) ..
      const int hr = BWD ? (py + 2 - kh) * HW + px + 2 - kw : (py + kh) * HW + px + kw;     // (the data gradient gathers mirrored)
      const f32x4 a = *reinterpret_cast<const f32x4*>(T1s + hr * LDT + 8 * (g & 7) + 4 * h);
      const f32x4 b = ldg4(wrow + tap * M + 8 * (g & 7));       // (the compiler keeps two of these in flight ahead of the MFMAs)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    // four K quarters: wave kq finishes registers 4 kq .. 4 kq + 3 and hands the other twelve to their owners
    auto give = [&](auto Q) {
      constexpr int q = decltype(Q)::value;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (o == q) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((ct * 4 + o) * 4 + q) * 4 + r) * 64 + lane] = acc[4 * o + r];   // [ct][owner][source][r]
      }
    };
    auto finish = [&](auto Q) {
      constexpr int q = decltype(Q)::value;
      const int co = ct * 32 + li;
      float sv[4];
      size_t pix[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = rowidx(4 * q + r) + 4 * h, qy = i >> 3, qx = i & 7;
        pix[r] = ((size_t)(img * H + y0 + qy) * W + x0 + qx) * M + co;
        if constexpr (BWD) sv[r] = p.st1[pix[r]];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float part[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) part[src] = src == q ? acc[4 * q + r] : red[(((ct * 4 + q) * 4 + src) * 4 + r) * 64 + lane];
        float v = ((part[0] + part[1]) + (part[2] + part[3])) + bv2;           // fixed order, whoever adds
        if constexpr (BWD) v = sv[r] > 0.f ? v : 0.f;
        else v = v > 0.f ? v : 0.f;
        const int i = rowidx(4 * q + r) + 4 * h;
        T2s[i * LDT + co] = v;
        if (!(p.ablate & 2)) p.t2[pix[r]] = v;
      }
    };
    switch (kq) {
      case 0: give(std::integral_constant<int, 0>{}); break;
      case 1: give(std::integral_constant<int, 1>{}); break;
      case 2: give(std::integral_constant<int, 2>{}); break;
      default: give(std::integral_constant<int, 3>{}); break;
    }
    lds_barrier();
    switch (kq) {
      case 0: finish(std::integral_constant<int, 0>{}); break;
      case 1: finish(std::integral_constant<int, 1>{}); break;
      case 2: finish(std::integral_constant<int, 2>{}); break;
      default: finish(std::integral_constant<int, 3>{}); break;
    }
    lds_barrier();
  }
  // ---- stage 3: y = relu(x + W3 t2 + b3) on 32 pixels x 128 channels: wave -> (column tile ct, K half kh)
  {
    const int ct = wave & 3, kh = wave >> 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(T2s + li * LDT + 32 * kh + 8 * t + 4 * h);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw3[t][s], acc, 0, 0, 0);
    }
    auto give = [&](auto OTHER) {
#pragma unroll
      for (int r = 0; r < 8; ++r) red[((ct * 2 + kh) * 8 + r) * 64 + lane] = acc[decltype(OTHER)::value + r];
    };
    if (kh == 0) give(std::integral_constant<int, 8>{}); else give(std::integral_constant<int, 0>{});
    lds_barrier();
    auto finish = [&](auto MINE) {
      constexpr int R0 = decltype(MINE)::value;
      const int co = ct * 32 + li;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int i = rowidx(R0 + r) + 4 * h, qy = i >> 3, qx = i & 7;
        const int hr = (qy + 1) * HW + qx + 1;
        const int xi = hr * C + ((((co >> 2) ^ (hr & 31))) << 2) + (co & 3);
        float xv = Xs[xi];
        if constexpr (BWD) xv = Ys[xi] > 0.f ? xv : 0.f;                        // the identity branch's gradient: dy . [y > 0]
        const float other = red[((ct * 2 + (kh ^ 1)) * 8 + r) * 64 + lane];
        float v = ((kh == 0 ? acc[R0 + r] + other : other + acc[R0 + r]) + bv3) + xv;   // residual BEFORE the activation (relu(out + identity))
        if constexpr (!BWD) v = v > 0.f ? v : 0.f;
        if (!(p.ablate & 2) || v == 123.456f) p.y[((size_t)(img * H + y0 + qy) * W + x0 + qx) * C + co] = v;
      }
    };
    if (kh == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 8>{});
  }
}

}  // namespace

static int ru_launch(const clc_ru_desc* d, bool bwd, clc_stream_t stream, const char* who) {
  CLC_CHECK(d && d->x && d->t1 && d->t2 && d->y, "%s: null pointer", who);
  CLC_CHECK(d->H == 16 && d->W == 16 && d->C == 128, "%s: built for 16x16 maps with 128 channels (got %dx%d, %d)", who, d->H, d->W, d->C);
  CLC_CHECK(d->sets == 1 || d->sets == 2 || d->sets == 4, "%s: 1, 2 or 4 filter sets", who);
  CLC_CHECK(d->N > 0 && d->N % d->sets == 0, "%s: the batch must be a multiple of the number of filter sets", who);
  CLC_CHECK(d->ldx >= 128 && d->ldx % 4 == 0 && aligned16(d->x) && aligned16(d->t1) && aligned16(d->t2) && aligned16(d->y), "%s: alignment", who);
  CLC_CHECK(!bwd || (d->saved_y && d->saved_t2 && d->saved_t1 && aligned16(d->saved_y)), "%s: the saved activations are missing", who);
  RUParams p;
  p.x = d->x; p.ldx = d->ldx; p.t1 = d->t1; p.t2 = d->t2; p.y = d->y; p.N = d->N; p.per_set = d->N / d->sets;
  p.sy = d->saved_y; p.st2 = d->saved_t2; p.st1 = d->saved_t1;
  for (int s = 0; s < 4; ++s) {
    const int k = s < d->sets ? s : 0;
    CLC_CHECK(d->w1[k] && d->w2[k] && d->w3[k] && aligned16(d->w1[k]) && aligned16(d->w2[k]) && aligned16(d->w3[k]), "%s: filter set %d missing / unaligned", who, k);
    CLC_CHECK(bwd || (d->b1[k] && d->b2[k] && d->b3[k]), "%s: bias of filter set %d missing", who, k);
    p.w1[s] = d->w1[k]; p.b1[s] = d->b1[k]; p.w2[s] = d->w2[k]; p.b2[s] = d->b2[k]; p.w3[s] = d->w3[k]; p.b3[s] = d->b3[k];
  }
  const size_t xb = ((size_t)d->N * 256 - 1) * d->ldx * 4 + 128 * 4;
  CLC_CHECK(xb < (1ull << 31), "%s: tensor larger than 2 GiB", who);
  p.x_bytes = (unsigned)xb;
  p.ablate = clc_tuning[CLC_TUNE_ABLATE];
  constexpr size_t lds_f = (size_t)(64 * 128 + 64 * 68 + 32 * 68 + 8192) * sizeof(float), lds_b = lds_f + 64 * 128 * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ru_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ru_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
  }
  if (bwd) hipLaunchKernelGGL(ru_fused_kernel<true>, dim3(d->N * 8), dim3(512), lds_b, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ru_fused_kernel<false>, dim3(d->N * 8), dim3(512), lds_f, (hipStream_t)stream, p);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_residual_unit_fwd(const clc_ru_desc* d, clc_stream_t stream) { return ru_launch(d, false, stream, "clc_residual_unit_fwd"); }
extern "C" int clc_residual_unit_dgrad(const clc_ru_desc* d, clc_stream_t stream) { return ru_launch(d, true, stream, "clc_residual_unit_dgrad"); }
