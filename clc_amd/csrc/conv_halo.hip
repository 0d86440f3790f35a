// conv_halo.hip — 3x3 / stride-1 convolutions with 128 input channels on maps of 32x32 and larger, barrier-free K loop (gfx950).
//
// The layers: the second convolutions of ResidualBlockWithStride / ResidualBlockUpsample (128 -> 128) and the sub-pixel convolutions
// (128 -> 512 + PixelShuffle) of the synthesis transform, forward, and the data gradients of the former
// (/root/reference/models/CLC_run.py:335-354 via compressai.layers).  On conv_igemm_dma2_kernel<128,128,4,2>
// they run at 72-75 % MFMA-busy: every 32-deep K step re-stages a 16 KB operand tile AND a 16 KB filter tile through LDS behind a
// workgroup barrier + vmcnt(0) (2 048 MFMA cycles per wave between barriers), and every operand pixel is gathered nine times.
//
// Here:
//   * the workgroup's 8 x 16-pixel output tile keeps its 10 x 18-pixel input HALO resident in LDS (180 pixels x 128 channels x 4 B = 90 KB),
//     deposited ONCE per tile by LDS-DMA.  Tap (kh, kw) of the K loop is a constant offset into that image: no re-gather, no bounds logic
//     in the loop.  A pixel's 16-B chunk c sits in slot c ^ (halo column & 15): the ds_read_b128 lane groups of a fragment read ({0-3, 12-15,
//     20-27}, {4-11, 16-19, 28-31} per half) then hit 16 distinct 16-B bank groups for every tap (MI355X_MICROARCH.md, LDS table);
//   * the filter never touches LDS: it is pre-packed in FRAGMENT ORDER (clc_filter_pack_halo: [n-tile][wave column][K step][j][t8][lane] x 16 B,
//     so that the 8 KB a wave needs for one K step are contiguous and every wave-instruction reads 1 KB of whole lines) and streamed from L2
//     straight into the B operand registers, one K step ahead;
//   * so the K loop has NO barrier and no vmcnt(0): the 8 waves (4 pixel-row pairs x 2 column halves, 2 per SIMD) drift apart, one wave's
//     loads land under the other's MFMAs.  Workgroups are persistent over (pixel tile, n-tile) items in pixel-tile-major order: the halo is
//     loaded once for all the n-tiles of a pixel tile (128 -> 512: 4).
// K order per output element = conv_igemm_dma2_kernel's (tap-major: kh, kw, then 32-channel chunk, pairs {8t+s, 8t+4+s}) and the epilogue is
// epilogue_regs / the same arithmetic -> THE SAME BITS as the tiled kernel (tests/test_kernels_gpu.py::test_halo_conv_same_bits_as_tiled):
// the codec may meet either.
#include "common.h"

namespace {
#include "conv_common.h"

constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2, HPIX = HH * HW;   // output tile, halo
// CI = input channels: 128 (8 waves = 4 pixel-row pairs x 2 column halves, n-tile 128, halo 90 KB: one PERSISTENT workgroup per CU) or
// 64 (the ResidualBlocks of the ConvTransBlocks: 4 waves = 4 row pairs, n-tile 64, halo 45 KB: TWO workgroups per CU — one's halo deposit
// and epilogue run under the other's MFMAs, which the single persistent workgroup of the 128-channel form cannot do; three fit the LDS but spill)

struct HaloParams {
  ConvParams c;
  const float* wpk;      // packed filter (clc_filter_pack_halo)
  int items, ntn;        // (pixel tile, n-tile) pairs; n-tiles of 128 output channels
  unsigned wpk_bytes;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// TR: data gradient of a stride-1 'same' convolution: "x" is dY, the packed filter is the transposed one, tap (tj, ti) reads source pixel
// (oy + 1 - tj, ox + 1 - ti).  SHUF: PixelShuffle(2) store with bias + {none, LeakyReLU, ReLU} (the sub-pixel convolutions).
template <int CI, bool TR, bool SHUF>
__global__ __launch_bounds__(CI == 128 ? 512 : 256, CI == 128 ? 1 : 2) void conv_halo3x3_kernel(const HaloParams hp) {
  constexpr int KCN = CI / 32, KSTEPS = 9 * KCN;                 // 32-channel groups per tap, K steps
  constexpr int PXB = CI * 4;                                    // bytes per halo pixel
  constexpr int NWAVE = CI == 128 ? 8 : 4, NWC = NWAVE / 4, NTW = 64 * NWC;
  constexpr int PPP = 1024 / PXB, SPP = CI / 4, PIECES = HPIX / PPP;   // pixels per 1-KB DMA piece, 16-B slots per pixel
  static_assert(HPIX % PPP == 0, "halo pieces");
  const ConvParams& p = hp.c;
  extern __shared__ __attribute__((aligned(16))) float halo[];   // [HPIX][CI], chunk-swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wr = wave & 3, wc = wave >> 2;
  const int tiles_w = p.W / TW, tiles_h = p.H / TH;
  const int i0 = (int)((long)hp.items * blockIdx.x / gridDim.x), i1 = (int)((long)hp.items * (blockIdx.x + 1) / gridDim.x);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hp.wpk), 0, hp.wpk_bytes, 0x00020000);

  // this lane's pixel inside the tile and its fragment addresses: byte address of chunk c of halo pixel (hy, hx) = (hy * HW + hx) * 512 +
  // ((c ^ (hx & 15)) << 4); c = kc * 8 + 2 * t8 + h.  Per tap column q (hx = px + q forward, px + 2 - q for a data gradient) the lane keeps
  // hg[q] = (h ^ (hx & 15)) << 4, so a fragment address is base + tap offset + (((kc * 8 + 2 * t8) << 4) ^ hg[q]).
  const int py = 2 * wr + (li >> 4), px = li & 15;
  const int a_base = (py * HW + px) * PXB;
  int hg[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) hg[q] = (h ^ ((px + (TR ? 2 - q : q)) & 15)) << 4;
  const unsigned b_lane = (unsigned)lane * 16u;

  const RegEpi re = make_reg_epi(p);
  int cur_tile = -1;
  for (int it = i0; it < i1; ++it) {
    const int pt = it / hp.ntn, nt = it - pt * hp.ntn;
    const int tx = pt % tiles_w, t2 = pt / tiles_w, ty = t2 % tiles_h, n = t2 / tiles_h;
    const int oy0 = ty * TH, ox0 = tx * TW;
    if (pt != cur_tile) {   // block-uniform: a new pixel tile -> its halo
      if (cur_tile >= 0) __syncthreads();   // every wave has finished reading the old one
      cur_tile = pt;
      // pieces of 1 KB (two halo pixels each at 128 channels, four at 64): wave w deposits pieces w, w + NWAVE, ...
      const int org = ((n * p.H + oy0 - 1) * p.W + ox0 - 1) * p.ldx;   // halo pixel (0, 0); may lie outside the image (never dereferenced then)
#pragma unroll 1
      for (int q = wave; q < PIECES; q += NWAVE) {
        const int P = q * PPP + lane / SPP, hy = P / HW, hx = P - hy * HW;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int c = (lane % SPP) ^ (hx & 15);
        dma16(xr, halo + q * 256, ok ? (unsigned)(org + (hy * p.W + hx) * p.ldx + c * 4) * 4u : kOOB);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // ---- K loop: 9 x CI / 32 steps, barrier-free.  Step (kh, kw, kc): A fragments from the halo, B fragments prefetched one step ahead from L2.
    const unsigned b_item = (unsigned)((nt * NWC + wc) * KSTEPS) * 8192u;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 bq[2][2][4], aq[2][4];
    // one quarter (t8) of a K step's operands: 2 x 1 KB of filter fragments from L2, one fragment read from the halo
    auto fetch = [&](int slot, int t8, unsigned kt, int toff, int q, int kc) {
      const unsigned so = b_item + kt * 8192u;
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[slot][j][t8] = buf_load4(wr_, b_lane + so + (unsigned)((j * 4 + t8) * 1024));
      const char* base = reinterpret_cast<const char*>(halo) + a_base + toff;
      aq[slot][t8] = *reinterpret_cast<const f32x4*>(base + ((((kc * 8 + 2 * t8) << 4)) ^ hg[q]));
    };
    auto tap_off = [&](int kh, int kw) { return TR ? ((2 - kh) * HW + (2 - kw)) * PXB : (kh * HW + kw) * PXB; };
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) fetch(0, t8, 0u, tap_off(0, 0), 0, 0);
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int s = 0; s < 3 * KCN; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        // The next step's operands are requested quarter by quarter IN FRONT of this step's MFMA groups and pinned there (sched_barrier):
        // left to itself the scheduler sinks every load to just before its first use, i.e. to one step later, and the loop waits for L2
        // on every group.  (Past the last step: a harmless re-read of step 0's operands.)
        const int kh1 = s < 3 * KCN - 1 ? kh : (kh < 2 ? kh + 1 : 0), s1 = s < 3 * KCN - 1 ? s + 1 : 0;
        const unsigned kt1 = (unsigned)(kh1 * 3 * KCN + s1);
        const int toff1 = tap_off(kh1, s1 / KCN);
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) {
          fetch(nxt, t8, kt1, toff1, s1 / KCN, s1 % KCN);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ss = 0; ss < 4; ++ss)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = MFMA(aq[cur][t8][ss], bq[cur][j][t8][ss], acc[j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- epilogue, per wave, straight from the accumulators
    const int n0 = nt * NTW + wc * 64;
    if (!SHUF) {
      const unsigned row0 = (unsigned)((n * p.OH + oy0 + 2 * wr) * p.OW + ox0 + 4 * h);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = n0 + j * 32 + li;
        epilogue_regs(p, re, acc[j], p.bias ? p.bias[co] : 0.f, row0, co, (unsigned)(p.OW - TW));
      }
    } else {
      // PixelShuffle(2): channel co of pixel (oy, ox) -> channel co >> 2 of pixel (2 oy + ((co >> 1) & 1), 2 ox + (co & 1)); bias, then the
      // activation (epilogue_store's arithmetic for these launches: no residual / norm / gates / saved pre-activation: host-checked)
      auto st32 = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0); };
      const int OW2 = 2 * p.OW;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = n0 + j * 32 + li;
        const float bv = p.bias ? p.bias[co] : 0.f;
        const unsigned lane_off = (unsigned)(((((co >> 1) & 1) * OW2 + (co & 1)) * p.ldy + (co >> 2)) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = (r & 3) + 8 * (r >> 2);                       // + 4 h: row of the wave's 32 (two 16-pixel pieces)
          const int ry = idx >> 4, rx = (idx & 15) + 4 * h;             // (idx & 15) + 4 h <= 15: the same piece
          const unsigned soff = (unsigned)((((n * 2 * p.OH + 2 * (oy0 + 2 * wr + ry)) * OW2) + 2 * ox0) * p.ldy * 4);
          float v = acc[j][r] + bv;
          v = apply_act(v, p.act);
          st32(v, re.y_r, lane_off + (unsigned)(2 * rx * p.ldy * 4), soff);
        }
      }
    }
  }
}

// ---- filter packing: [N][9][K] (K-contiguous rows: the forward filter [Cout][kh][kw][Cin], or the transposed one [Cin][kh][kw][Cout]; K = 128 or
// 64) -> fragment order [n-tile of 64 NWC][NWC wave columns][9 K / 32 K steps][2 blocks][4 t8][64 lanes] x 16 B, NWC = K / 64.
// One thread per 16-B element of the output.
__device__ __forceinline__ void pack_one(const float* __restrict__ w, float* __restrict__ out, long e, int K) {
  const int KCN = K / 32, KST = 9 * KCN, NWC = K / 64;
  const int lane = (int)(e & 63);
  long r = e >> 6;
  const int t8 = (int)(r & 3); r >>= 2;
  const int j = (int)(r & 1); r >>= 1;
  const int kt = (int)(r % KST); r /= KST;
  const int wc = (int)(r % NWC); const int nt = (int)(r / NWC);
  const int row = nt * 64 * NWC + wc * 64 + j * 32 + (lane & 31);
  const int k = (kt % KCN) * 32 + 8 * t8 + 4 * (lane >> 5);
  *reinterpret_cast<f32x4*>(out + e * 4) = *reinterpret_cast<const f32x4*>(w + ((size_t)row * 9 + kt / KCN) * K + k);
}
__global__ void filter_pack_halo_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // index of a float4 of the output
  if (e >= (long)N * 9 * K / 4) return;
  pack_one(w, out, e, K);
}
// every eligible filter of a model in ONE launch (clc_amd.train.HaloPacker): block b finds its entry by binary search over block_begin
__global__ void filter_pack_halo_batched_kernel(const clc_halo_pack_entry* __restrict__ table, int n_entries) {
  int lo = 0, hi = n_entries - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  const clc_halo_pack_entry en = table[lo];
  const long e = (long)(b - en.block_begin) * blockDim.x + threadIdx.x;
  if (e >= (long)en.N * 9 * en.K / 4) return;
  pack_one(en.w, en.out, e, en.K);
}

}  // namespace

extern "C" int clc_filter_pack_halo_batched(const clc_halo_pack_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream) {
  CLC_CHECK(table_dev && n_entries > 0 && total_blocks > 0, "clc_filter_pack_halo_batched: bad args");
  hipLaunchKernelGGL(filter_pack_halo_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n_entries);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_filter_pack_halo(const float* w, float* out, int N, int K, clc_stream_t stream) {
  CLC_CHECK(w && out && (K == 128 || K == 64) && N > 0 && N % K == 0 && aligned16(w) && aligned16(out),
            "clc_filter_pack_halo: K must be 128 or 64, N a positive multiple of it, pointers 16-B aligned (got N=%d K=%d)", N, K);
  const long total = (long)N * 9 * K / 4;
  hipLaunchKernelGGL(filter_pack_halo_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, N, K);
  CLC_LAUNCH_CHECK();
  return 0;
}

// Called by clc_conv2d (conv_igemm.hip) with the filled kernel parameters; returns 0 when the launch does not qualify (the caller falls
// through to the tiled kernels), else the variant id (family 12).
int clc_conv_halo_launch(const void* conv_params, const float* wpk, hipStream_t st) {
  const ConvParams& p = *reinterpret_cast<const ConvParams*>(conv_params);
  const int CI = p.Cin;
  // bit 0: the 128-channel layers; bit 1: the 64-channel ones — OFF by default: graph-replayed alone the 64 -> 64 layers gain (8 x 128 x 128:
  // 91.9 -> 83.4 us, 8 x 64 x 64: 30.7 -> 27.0 at two workgroups per CU; three per CU spill 18 registers in the epilogue: 89.2), inside the
  // training step they gain NOTHING (26.625 vs 26.619 ms over three interleaved rounds; the three-per-CU build lost 0.15 ms): their epilogues —
  // bias + LeakyReLU + residual + saved pre-activation, gates — are heavier than the micro-benchmark's and two waves per SIMD hide less of them
  // than the tiled kernel's four
  if (!(clc_tuning[CLC_TUNE_HALO] & (CI == 128 ? 1 : 2))) return 0;
  if (!wpk || p.ks != 3 || p.stride != 1 || p.pad != 1 || (CI != 128 && CI != 64) || p.Cout % CI || p.H % TH || p.W % TW || p.OH != p.H || p.OW != p.W) return 0;
  if (p.xs || p.in_op != CLC_IN_NONE || p.group_rows || p.bf16 || p.ksplit > 1 || p.ldx % 4 || !aligned16(p.x) || !aligned16(wpk)) return 0;
  const bool shuf = p.shuffle != 0;
  if (shuf) {
    if (CI != 128 || p.transposed || p.res || p.norm != CLC_NORM_NONE || p.y_pre || p.out_gate || !(p.act == CLC_ACT_NONE || p.act == CLC_ACT_LRELU || p.act == CLC_ACT_RELU)) return 0;
    if ((size_t)p.N * p.OH * p.OW * 4 * (size_t)p.ldy * 4 >= (1ull << 31)) return 0;
  } else if (!reg_epi_ok(p, 128, CI)) {
    return 0;
  }
  HaloParams hp;
  hp.c = p;
  hp.wpk = wpk;
  hp.ntn = p.Cout / CI;
  hp.items = p.N * (p.H / TH) * (p.W / TW) * hp.ntn;
  hp.wpk_bytes = (unsigned)((size_t)p.Cout * 9 * CI * 4);
  if (hp.items < 128) return 0;   // a quarter-filled chip: the 64 x 64 tiles do better
  // Data gradients whose workgroups each walk several pixel tiles (128 -> 128 on 8 x 128 x 128: 1024 tiles): every tile boundary is a halo
  // deposit all 256 workgroups make at once (6-8 us of a 61-us tile) plus an epilogue with gate / residual operands that only two waves per
  // SIMD overlap — measured inside the step 316 / 345 us against 301 / 311 on the tiled kernel (forward launches of the same shape: 293
  // against 307).  A double-buffered variant with a dedicated loader wave (4 x 16 tiles, nine waves) was built to hide the deposit and
  // measured SLOWER everywhere (profiles/r5_halo_v2_loader_wave_microbench.txt): shorter items, more restarts of the operand pipeline.
  if (CI == 128 && p.transposed && hp.items >= 1024) return 0;
  // 128 channels: one persistent workgroup per CU over contiguous item ranges; 64 channels: one item per workgroup, three resident per CU
  // (64 channels: a grid that fills whole rounds of the 768 resident workgroups — 1024 tiles as 768 + 256 left a round with one workgroup, one wave
  //  per SIMD, per CU: 101.7 us against 92.6 tiled; as 512 workgroups of two tiles each every round has two)
  int grid = hp.items < 256 ? hp.items : 256;
  if (CI == 64) {
    const int per = (hp.items + 767) / 768;
    grid = (hp.items + per - 1) / per;
  }
  const int lds = HPIX * CI * 4;
  static PerDeviceOnce once[8];
  auto launch = [&](auto kern, int slot, int threads) {
    if (once[slot].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, hp);
  };
  if (CI == 128) {
    if (p.transposed) launch(conv_halo3x3_kernel<128, true, false>, 0, 512);
    else if (shuf) launch(conv_halo3x3_kernel<128, false, true>, 1, 512);
    else launch(conv_halo3x3_kernel<128, false, false>, 2, 512);
  } else {
    if (p.transposed) launch(conv_halo3x3_kernel<64, true, false>, 3, 256);
    else launch(conv_halo3x3_kernel<64, false, false>, 4, 256);
  }
  CLC_LAUNCH_CHECK();
  return (12 << 20) | ((CI / 64) << 4) | ((p.transposed ? 1 : 0) << 1) | (shuf ? 1 : 0);
}
