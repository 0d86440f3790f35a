// conv_wino.hip — Winograd F(2x2, 3x3) for the 3x3 / stride-1 convolutions with 128 k input channels on maps of 32x32 and larger (gfx950).
//
// The f32 matrix rate of this chip is its vector rate (157 TF nominal, ~143 TF at the 2.2 GHz it holds under load), the 3x3 / stride-1 layers of the
// analysis / synthesis transforms (second convolutions of ResidualBlockWithStride / Upsample, the sub-pixel convolutions, the ConvTransBlocks'
// ResidualBlocks; /root/reference/models/CLC_run.py:335-354 via compressai.layers) carry most of the step's FLOPs, and the tiled / halo kernels run
// them at 80-88 % MFMA-busy: what is left is doing fewer multiplications.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16
// multiplications per (input channel, output channel) instead of 36: 2.25x fewer MFMAs for the same convolution
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          (Lavin & Gray 2016; B, G, A below; summed over input channels in the transformed domain)
// at an fp32 error of a few ulp of the operands (the transforms are sums of <= 4 terms and halvings), inside the 2e-5 / 1e-4 bars of the kernel tests.
//
// One workgroup (8 waves, one per CU, persistent over (pixel tile, n-tile) items) per 8 x 16-pixel output tile = 4 x 8 Winograd tiles:
//   * the 10 x 18-pixel halo of 128 input channels is resident in LDS (conv_halo.hip's image: 90 KB, LDS-DMA, chunk-swizzled);
//   * per 32-channel chunk: every thread transforms the patches of one (tile, 4 channels, half) -> V[16][32 tiles][32 ch] in LDS (64 KB);
//     wave w then owns transformed positions xi = 2 w, 2 w + 1: D_xi[32 tiles x 128 out] += V_xi [32 x 32] . U_xi [32 x 128], 128 MFMAs, the
//     transformed filter U streamed from L2 in fragment order (clc_filter_wino: 16 / 9 of the filter's size, packed per step like the halo
//     kernel's image), one (xi, 32-channel block) group ahead, pinned by sched_barrier;
//   * after the last chunk the 16 x (32 x 128) accumulators meet through LDS, 32 output channels at a time, and every thread applies A^T . A to
//     one (tile, 4 channels, output row) and hands its two pixels to the SHARED epilogue (epilogue_store4 / epilogue_store: bias, activation,
//     residual, gates, saved pre-activation, PixelShuffle — everything the tiled kernels do).
// Input channels beyond 128 (data gradients of the sub-pixel convolutions: 512) pass through the halo 128 at a time.
// ANOTHER summation order than the direct kernels (transformed domain): the host side hands the transformed filter (clc_conv_desc.w_wino) to
// TRAINING launches only — recorded forwards and data gradients; eval forwards, the parity measurement and the codec keep the direct kernels' bits.
#include "common.h"

namespace {
#include "conv_common.h"

constexpr int TH = 8, TW = 16, HW = TW + 2, HPIX = (TH + 2) * HW;   // output tile, halo
constexpr int HALO_FLOATS = HPIX * 128;                              // 92 160 B
constexpr int V_FLOATS = 16 * 32 * 32;                               // 65 536 B
constexpr int PIECES = HPIX / 2;                                      // 1-KB LDS-DMA pieces of the halo (two pixels each)

struct WinoParams {
  ConvParams c;
  const float* u;        // transformed filter, fragment order (clc_filter_wino)
  int items, ntn, ncg;   // (pixel tile, n-tile) pairs; n-tiles of 128 output channels; groups of 128 input channels
  unsigned u_bytes;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <bool SHUF>
__global__ __launch_bounds__(512, 1) void conv_wino_kernel(const WinoParams wp) {
  const ConvParams& p = wp.c;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                 // [HPIX][128], 16-B chunk c of halo pixel (hy, hx) in slot c ^ (hx & 15)
  float* V = smem + HALO_FLOATS;      // [16][32][32]: chunk q of row t in slot q ^ ((t >> 1) & 7); later the accumulators' meeting place [16][32][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_w = p.W / TW, tiles_h = p.H / TH;
  const int KCT = wp.ncg * 4;         // 32-channel chunks in all
  const int i0 = (int)((long)wp.items * blockIdx.x / gridDim.x), i1 = (int)((long)wp.items * (blockIdx.x + 1) / gridDim.x);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp.u), 0, wp.u_bytes, 0x00020000);
  // transform / output-transform role of this thread: (tile t, 16-B chunk q, half)
  const int q = tid & 7, t = (tid >> 3) & 31, half = tid >> 8;
  const int ty = t >> 3, tx = t & 7;
  // MFMA role: A fragments of row li of V_xi
  int afo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) afo[t8] = li * 32 + (((2 * t8 + h) ^ ((li >> 1) & 7)) << 2);
  const unsigned b_lane = (unsigned)lane * 16u;

  int cur_pt = -1, cur_cg = -1;
  for (int it = i0; it < i1; ++it) {
    const int pt = it / wp.ntn, nt = it - pt * wp.ntn;
    const int txx = pt % tiles_w, t2 = pt / tiles_w, tyy = t2 % tiles_h, n = t2 / tiles_h;
    const int oy0 = tyy * TH, ox0 = txx * TW;
    f32x16 acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][nb][r] = 0.f;
    f32x4 bq[2][4];
    auto load_b = [&](int slot, int xi, int kcg, int nb) {
      const unsigned base = (unsigned)((((nt * 16 + xi) * KCT + kcg) * 4 + nb) * 4) * 1024u;
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) bq[slot][t8] = buf_load4(ur, b_lane + base + (unsigned)(t8 * 1024));
    };
    load_b(0, 2 * wave, 0, 0);      // the first group's filter fragments: in flight under the halo deposit and the first transform

    for (int cg = 0; cg < wp.ncg; ++cg) {
      if (pt != cur_pt || cg != cur_cg) {   // block-uniform: this (pixel tile, channel group)'s halo.  (Every wave is past the last transform that
        cur_pt = pt; cur_cg = cg;           //  read the old one: the barrier behind that transform.)
        const int org = ((n * p.H + oy0 - 1) * p.W + ox0 - 1) * p.ldx + cg * 128;
#pragma unroll 1
        for (int pc = wave; pc < PIECES; pc += 8) {
          const int P = 2 * pc + h, hy = P / HW, hx = P - hy * HW;
          const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const int c = li ^ (hx & 15);
          dma16(xr, halo + pc * 256, ok ? (unsigned)(org + (hy * p.W + hx) * p.ldx + c * 4) * 4u : kOOB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also this wave's pending filter fragments: harmless)
      }
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        __syncthreads();     // the halo has landed; every wave is done reading V (previous chunk's MFMAs / previous item's output rounds)
        // ---- input transform B^T d B of this thread's (tile, 4 channels): half 0 -> rows 0, 1 of the 4 x 4 result, half 1 -> rows 2, 3
        {
          const int cidx = kc * 8 + q;
          f32x4 d[3][4];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int hy = 2 * ty + r + half, hx = 2 * tx + c;
              d[r][c] = *reinterpret_cast<const f32x4*>(halo + (hy * HW + hx) * 128 + ((cidx ^ (hx & 15)) << 2));
            }
          // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows (0, 1) from d rows 0..2, rows (2, 3) from d rows 1..3 (= this thread's d[0..2])
          f32x4 T[2][4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (half == 0) { T[0][c] = d[0][c] - d[2][c]; T[1][c] = d[1][c] + d[2][c]; }
            else           { T[0][c] = d[1][c] - d[0][c]; T[1][c] = d[0][c] - d[2][c]; }
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x4 v0 = T[i][0] - T[i][2], v1 = T[i][1] + T[i][2], v2 = T[i][2] - T[i][1], v3 = T[i][1] - T[i][3];
            float* dst = V + (((2 * half + i) * 4) * 32 + t) * 32 + ((q ^ ((t >> 1) & 7)) << 2);
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 1 * 32 * 32) = v1;
            *reinterpret_cast<f32x4*>(dst + 2 * 32 * 32) = v2;
            *reinterpret_cast<f32x4*>(dst + 3 * 32 * 32) = v3;
          }
        }
        __syncthreads();
        // ---- 16 batched GEMMs, two per wave: D_xi += V_xi . U_xi over this chunk's 32 channels
        const int kcg = cg * 4 + kc;
        f32x4 aq[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int t8 = 0; t8 < 4; ++t8) aq[s][t8] = *reinterpret_cast<const f32x4*>(V + (2 * wave + s) * 1024 + afo[t8]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int s = g >> 2, nb = g & 3, cur = g & 1;
          // the next group's filter fragments first (pinned: the scheduler would sink them to their first use) — past the chunk's last
          // group, the first group of the next chunk / channel group (or a harmless re-read at the very end)
          if (g < 7) load_b(cur ^ 1, 2 * wave + ((g + 1) >> 2), kcg, (g + 1) & 3);
          else load_b(cur ^ 1, 2 * wave, kcg + 1 < KCT ? kcg + 1 : 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t8 = 0; t8 < 4; ++t8)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) acc[s][nb] = MFMA(aq[s][t8][ss], bq[cur][t8][ss], acc[s][nb]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- output transform A^T M A and the epilogue, 32 output channels at a time through LDS
#pragma unroll 1
    for (int nb = 0; nb < 4; ++nb) {
      __syncthreads();       // V / the previous round's image is free
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float* dst = V + (2 * wave + s) * 1024 + li;
        // (a switch over nb keeps the accumulator index static)
        const f32x16& a = nb == 0 ? acc[s][0] : (nb == 1 ? acc[s][1] : (nb == 2 ? acc[s][2] : acc[s][3]));
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * h) * 32] = a[r];
      }
      __syncthreads();
      {
        // this thread: tile t, channels 4 q .. 4 q + 3 of the block, output row `half` of the 2 x 2
        f32x4 m[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(V + (((i + half) * 4 + j) * 32 + t) * 32 + (q << 2));
        // A^T = [1 1 1 0; 0 1 -1 -1]
        f32x4 S[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) S[j] = half == 0 ? (m[0][j] + m[1][j]) + m[2][j] : (m[0][j] - m[1][j]) - m[2][j];
        const f32x4 y0 = (S[0] + S[1]) + S[2], y1 = (S[1] - S[2]) - S[3];
        const int py = oy0 + 2 * ty + half, px = ox0 + 2 * tx;
        const int mrow = (n * p.OH + py) * p.OW + px;
        const int co = nt * 128 + nb * 32 + q * 4;
        if (!SHUF) {
          epilogue_store4(p, p.bias, y0, mrow, co, p.OH, p.OW, 0, 0);
          epilogue_store4(p, p.bias, y1, mrow + 1, co, p.OH, p.OW, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bv = p.bias ? p.bias[co + e] : 0.f;
            epilogue_store(p, y0[e], bv, mrow, co + e, p.OH, p.OW, 0, 0);
            epilogue_store(p, y1[e], bv, mrow + 1, co + e, p.OH, p.OW, 0, 0);
          }
        }
      }
    }
    __syncthreads();         // the next item's transforms write V
  }
}

// ---- the 64-wide instantiation: 64 input channels resident (45 KB halo), 64 output channels per item, 16-channel chunks (V: 32 KB) — 78 KB of LDS and
// <= 256 VGPRs, so TWO workgroups of 4 waves share a CU and one's transforms / output rounds run under the other's MFMAs.  Wave w owns row w of
// the 4 x 4 transformed positions (xi = 4 w .. 4 w + 3): 4 x (32 tiles x 64 out) accumulators, 64 MFMAs per chunk.  Takes the 64 -> 64 layers of
// the transforms' first / last stages (128 x 128 maps at 256 x 256 input), which the 128-wide kernel cannot.
constexpr int HALO2_FLOATS = HPIX * 64;   // 46 080 B
constexpr int V2_FLOATS = 16 * 32 * 16;   // 32 768 B
constexpr int PIECES2 = HPIX / 4;         // 1-KB LDS-DMA pieces (four pixels each)

template <bool SHUF>
__global__ __launch_bounds__(256, 2) void conv_wino64_kernel(const WinoParams wp) {
  const ConvParams& p = wp.c;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                  // [HPIX][64], 16-B chunk c of halo pixel (hy, hx) in slot c ^ (hx & 15)
  float* V = smem + HALO2_FLOATS;      // [16][32][16]: chunk q of row t in slot q ^ ((t >> 2) & 3)
  float* S = smem;                     // output rounds: [16][32][32] over the halo and the head of V (both free by then)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_w = p.W / TW, tiles_h = p.H / TH;
  const int KCT = wp.ncg * 2;          // 32-channel chunks of the packed filter (wp.ncg: groups of 64 input channels)
  const int NC16 = wp.ncg * 4;         // 16-channel chunks in all
  const int i0 = (int)((long)wp.items * blockIdx.x / gridDim.x), i1 = (int)((long)wp.items * (blockIdx.x + 1) / gridDim.x);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp.u), 0, wp.u_bytes, 0x00020000);
  // input-transform role: (tile t, 16-B chunk q of the 16-channel chunk, half); output-transform role: (tile to, 16-B chunk qo of a 32-channel block)
  const int q = tid & 3, t = (tid >> 2) & 31, half = tid >> 7;
  const int ty = t >> 3, tx = t & 7;
  const int qo = tid & 7, to = tid >> 3;
  int afo[2];
#pragma unroll
  for (int t8 = 0; t8 < 2; ++t8) afo[t8] = li * 16 + (((2 * t8 + h) ^ ((li >> 2) & 3)) << 2);
  const unsigned b_lane = (unsigned)lane * 16u;

  int cur_pt = -1, cur_cg = -1;
  for (int it = i0; it < i1; ++it) {
    const int pt = it / wp.ntn, nt = it - pt * wp.ntn;      // nt: n-tile of 64 output channels = blocks 2 (nt & 1), +1 of 128-row tile nt >> 1
    const int txx = pt % tiles_w, t2 = pt / tiles_w, tyy = t2 % tiles_h, n = t2 / tiles_h;
    const int oy0 = tyy * TH, ox0 = txx * TW;
    f32x16 acc[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][nb][r] = 0.f;
    f32x4 bq[2][2][2];
    auto load_b = [&](int slot, int s, int c16) {     // xi = 4 wave + s, 16-channel chunk c16 (global): both 32-row blocks
      const int kcg = c16 >> 1, t8g = 2 * (c16 & 1);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const unsigned base = (unsigned)((((((nt >> 1) * 16 + 4 * wave + s) * KCT + kcg) * 4 + 2 * (nt & 1) + nb) * 4 + t8g)) * 1024u;
#pragma unroll
        for (int t8 = 0; t8 < 2; ++t8) bq[slot][nb][t8] = buf_load4(ur, b_lane + base + (unsigned)(t8 * 1024));
      }
    };
    load_b(0, 0, 0);

    for (int cg = 0; cg < wp.ncg; ++cg) {
      if (pt != cur_pt || cg != cur_cg) {   // block-uniform.  (Every wave is past the last read of the old halo / the output rounds: barriers below.)
        cur_pt = pt; cur_cg = cg;
        const int org = ((n * p.H + oy0 - 1) * p.W + ox0 - 1) * p.ldx + cg * 64;
#pragma unroll 1
        for (int pc = wave; pc < PIECES2; pc += 4) {
          const int P = 4 * pc + (lane >> 4), hy = P / HW, hx = P - hy * HW;
          const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const int c = (lane & 15) ^ (hx & 15);
          dma16(xr, halo + pc * 256, ok ? (unsigned)(org + (hy * p.W + hx) * p.ldx + c * 4) * 4u : kOOB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#pragma unroll 1
      for (int kc = 0; kc < 4; ++kc) {
        __syncthreads();     // the halo has landed; every wave is done reading V
        {
          const int cidx = kc * 4 + q;
          f32x4 d[3][4];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int hy = 2 * ty + r + half, hx = 2 * tx + c;
              d[r][c] = *reinterpret_cast<const f32x4*>(halo + (hy * HW + hx) * 64 + ((cidx ^ (hx & 15)) << 2));
            }
          f32x4 T[2][4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (half == 0) { T[0][c] = d[0][c] - d[2][c]; T[1][c] = d[1][c] + d[2][c]; }
            else           { T[0][c] = d[1][c] - d[0][c]; T[1][c] = d[0][c] - d[2][c]; }
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x4 v0 = T[i][0] - T[i][2], v1 = T[i][1] + T[i][2], v2 = T[i][2] - T[i][1], v3 = T[i][1] - T[i][3];
            float* dst = V + (((2 * half + i) * 4) * 32 + t) * 16 + ((q ^ ((t >> 2) & 3)) << 2);
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 1 * 32 * 16) = v1;
            *reinterpret_cast<f32x4*>(dst + 2 * 32 * 16) = v2;
            *reinterpret_cast<f32x4*>(dst + 3 * 32 * 16) = v3;
          }
        }
        __syncthreads();
        const int c16 = cg * 4 + kc;
        f32x4 aq[2][2];   // (one xi ahead, like the filter fragments: all four at once would not fit beside the accumulators)
#pragma unroll
        for (int t8 = 0; t8 < 2; ++t8) aq[0][t8] = *reinterpret_cast<const f32x4*>(V + (4 * wave) * 512 + afo[t8]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int cur = s & 1;
          if (s < 3) {
            load_b(cur ^ 1, s + 1, c16);
#pragma unroll
            for (int t8 = 0; t8 < 2; ++t8) aq[cur ^ 1][t8] = *reinterpret_cast<const f32x4*>(V + (4 * wave + s + 1) * 512 + afo[t8]);
          } else {
            load_b(cur ^ 1, 0, c16 + 1 < NC16 ? c16 + 1 : 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t8 = 0; t8 < 2; ++t8)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) acc[s][nb] = MFMA(aq[cur][t8][ss], bq[cur][nb][t8][ss], acc[s][nb]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- output transform A^T M A and the epilogue, 32 output channels at a time through LDS (over the halo: the next item re-deposits it)
    cur_pt = -1;
#pragma unroll 1
    for (int nb = 0; nb < 2; ++nb) {
      __syncthreads();       // the last chunk's MFMAs have read V / the previous round's image is free
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float* dst = S + (4 * wave + s) * 1024 + li;
        const f32x16& a = nb == 0 ? acc[s][0] : acc[s][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * h) * 32] = a[r];
      }
      __syncthreads();
      {
        const int tyo = to >> 3, txo = to & 7;
        const int co = nt * 64 + nb * 32 + qo * 4;
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {       // output row r of the 2 x 2: A^T = [1 1 1 0; 0 1 -1 -1] -> M rows r .. r + 2, signs + + / - -
          const float sg = r ? -1.f : 1.f;
          f32x4 Sr[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 m[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) m[i] = *reinterpret_cast<const f32x4*>(S + (((i + r) * 4 + j) * 32 + to) * 32 + (qo << 2));
            Sr[j] = (m[0] + sg * m[1]) + sg * m[2];
          }
          const f32x4 y0 = (Sr[0] + Sr[1]) + Sr[2], y1 = (Sr[1] - Sr[2]) - Sr[3];
          const int py = oy0 + 2 * tyo + r, px = ox0 + 2 * txo;
          const int mrow = (n * p.OH + py) * p.OW + px;
          if (!SHUF) {
#pragma unroll 1
            for (int e = 0; e < 2; ++e) epilogue_store4(p, p.bias, e ? y1 : y0, mrow + e, co, p.OH, p.OW, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float bv = p.bias ? p.bias[co + e] : 0.f;
              epilogue_store(p, y0[e], bv, mrow, co + e, p.OH, p.OW, 0, 0);
              epilogue_store(p, y1[e], bv, mrow + 1, co + e, p.OH, p.OW, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();         // the next item's halo deposit / transforms write over S
  }
}

// ---- filter transform U = G g G^T, [N][9][K] rows (K-contiguous) -> [n-tile of 128][16 xi][K / 32 chunks][4 blocks of 32 rows][4 t8][64 lanes] x 16 B.
// G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  flip: the taps reversed (the data gradient of a stride-1 'same' convolution is the same convolution
// with the transposed, 180-degree-rotated filter).  One thread per (row, 4 channels).
__device__ __forceinline__ void wino_u_one(const float* __restrict__ w, float* __restrict__ out, long e, int K, int flip) {
  const int kq = (int)(e % (K / 4)), n = (int)(e / (K / 4)), k = kq * 4;
  f32x4 g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int tap = flip ? 8 - (a * 3 + b) : a * 3 + b;
      g[a][b] = *reinterpret_cast<const f32x4*>(w + ((size_t)n * 9 + tap) * K + k);
    }
  f32x4 tt[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    tt[0][b] = g[0][b];
    tt[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
    tt[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
    tt[3][b] = g[2][b];
  }
  const int KCT = K / 32, nt = n >> 7, nb = (n & 127) >> 5, kcg = k >> 5, kk = k & 31, t8 = kk >> 3, hh = (kk & 7) >> 2;
  const int lane = (n & 31) + 32 * hh;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 u[4];
    u[0] = tt[i][0];
    u[1] = 0.5f * ((tt[i][0] + tt[i][1]) + tt[i][2]);
    u[2] = 0.5f * ((tt[i][0] - tt[i][1]) + tt[i][2]);
    u[3] = tt[i][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t idx = ((((((size_t)nt * 16 + (i * 4 + j)) * KCT + kcg) * 4 + nb) * 4 + t8) * 64 + lane);
      *reinterpret_cast<f32x4*>(out + idx * 4) = u[j];
    }
  }
}
__global__ void filter_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K, int flip) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)N * K / 4) return;
  wino_u_one(w, out, e, K, flip);
}
__global__ void filter_wino_batched_kernel(const clc_wino_entry* __restrict__ table, int n_entries) {
  int lo = 0, hi = n_entries - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  const clc_wino_entry en = table[lo];
  const long e = (long)(b - en.block_begin) * blockDim.x + threadIdx.x;
  if (e >= (long)en.N * en.K / 4) return;
  wino_u_one(en.w, en.out, e, en.K, en.flip);
}

}  // namespace

extern "C" int clc_filter_wino(const float* w, float* out, int N, int K, int flip, clc_stream_t stream) {
  CLC_CHECK(w && out && N > 0 && N % 64 == 0 && K > 0 && K % 64 == 0 && aligned16(w) && aligned16(out),
            "clc_filter_wino: rows and channels must be positive multiples of 64, pointers 16-B aligned (got N=%d K=%d)", N, K);
  const long total = (long)N * K / 4;
  hipLaunchKernelGGL(filter_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, N, K, flip);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_filter_wino_batched(const clc_wino_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream) {
  CLC_CHECK(table_dev && n_entries > 0 && total_blocks > 0, "clc_filter_wino_batched: bad args");
  hipLaunchKernelGGL(filter_wino_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n_entries);
  CLC_LAUNCH_CHECK();
  return 0;
}

// Called by clc_conv2d (conv_igemm.hip) with the filled kernel parameters; 0 = the launch does not qualify (the caller falls through), else the variant
// id (family 13: bit 0 shuffle, bit 1 transposed, bits 4..11 input-channel groups, bit 12 the 64-wide kernel).
int clc_conv_wino_launch(const void* conv_params, const float* u, hipStream_t st) {
  const ConvParams& p = *reinterpret_cast<const ConvParams*>(conv_params);
  if (!u || p.ks != 3 || p.stride != 1 || p.pad != 1 || p.Cin % 64 || p.Cin > 1024 || p.Cout % 64 || p.H % TH || p.W % TW || p.OH != p.H || p.OW != p.W) return 0;
  if (p.xs || p.in_op != CLC_IN_NONE || p.group_rows || p.bf16 || p.ksplit > 1 || p.ldx % 4 || !aligned16(p.x) || !aligned16(u)) return 0;
  const bool shuf = p.shuffle != 0;
  if (!shuf && !p.vec_epi) return 0;
  const size_t ub = (size_t)((p.Cout + 127) / 128 * 128) * 16 * p.Cin * 4;
  if (ub >= (1ull << 31)) return 0;
  const int mode = clc_tuning[CLC_TUNE_WINO];
  const bool wide = p.Cin % 128 == 0 && p.Cout % 128 == 0 && !(mode & 8);   // (bit 3: the 64-wide kernel on every layer — an experiment switch)
  if (!wide && !(mode & 4)) return 0;                                          // bit 2: the 64-wide kernel for layers of 64 k channels
  const int nw = wide ? 128 : 64;
  WinoParams wp;
  wp.c = p;
  wp.u = u;
  wp.ntn = p.Cout / nw;
  wp.ncg = p.Cin / nw;
  wp.items = p.N * (p.H / TH) * (p.W / TW) * wp.ntn;
  wp.u_bytes = (unsigned)ub;
  // A per-IMAGE rule (an image's result must not depend on the batch it is in): at least 32 items of 128 output channels per image (128 -> 128
  // from 64 x 64 maps up, 128 -> 512 from 32 x 32), 64 items of 64 (64 -> 64 from 64 x 128 maps up).
  if ((p.H / TH) * (p.W / TW) * wp.ntn < (wide ? 32 : 64)) return 0;
  static PerDeviceOnce once[4];
  const int k = (wide ? 0 : 2) + (shuf ? 1 : 0);
  if (wide) {
    const int grid = wp.items < 256 ? wp.items : 256;
    const int lds = (HALO_FLOATS + V_FLOATS) * 4;
    if (shuf) {
      if (once[k].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(conv_wino_kernel<true>, dim3(grid), dim3(512), lds, st, wp);
    } else {
      if (once[k].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(conv_wino_kernel<false>, dim3(grid), dim3(512), lds, st, wp);
    }
  } else {
    const int grid = wp.items < 512 ? wp.items : 512;
    const int lds = (HALO2_FLOATS + V2_FLOATS) * 4;
    if (shuf) {
      if (once[k].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(conv_wino64_kernel<true>, dim3(grid), dim3(256), lds, st, wp);
    } else {
      if (once[k].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(conv_wino64_kernel<false>, dim3(grid), dim3(256), lds, st, wp);
    }
  }
  CLC_LAUNCH_CHECK();
  return (13 << 20) | ((wide ? 0 : 1) << 12) | (wp.ncg << 4) | ((p.transposed ? 1 : 0) << 1) | (shuf ? 1 : 0);
}
