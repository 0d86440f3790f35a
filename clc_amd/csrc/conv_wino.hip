// conv_wino.hip — Winograd F(2x2, 3x3) for the 3x3 / stride-1 convolutions with 64 k input and output channels on maps of 32x32 and larger (gfx950).
//
// The f32 matrix rate of this chip is its vector rate (157 TF nominal, ~143 TF at the 2.2 GHz it holds under load), the 3x3 / stride-1 layers of the
// analysis / synthesis transforms (second convolutions of ResidualBlockWithStride / Upsample, the sub-pixel convolutions, the ConvTransBlocks'
// ResidualBlocks; /root/reference/models/CLC_run.py:335-354 via compressai.layers) carry most of the step's FLOPs, and the tiled / halo kernels run
// them at 80-88 % MFMA-busy: what is left is doing fewer multiplications.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16
// multiplications per (input channel, output channel) instead of 36: 2.25x fewer MFMAs for the same convolution
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          (Lavin & Gray 2016; B, G, A below; summed over input channels in the transformed domain)
// at an fp32 error of a few ulp of the operands (the transforms are sums of <= 4 terms and halvings), inside the 2e-5 / 1e-4 bars of the kernel tests.
//
// conv_wino_kernel — one workgroup (8 waves, one per CU, persistent over (pixel tile, n-tile) items) per 8 x 16-pixel output tile = 4 x 8 Winograd tiles:
//   * the 10 x 18-pixel halo of 128 input channels is resident in LDS (90 KB, LDS-DMA, chunk-swizzled); the halo of the next channel group / the
//     next item is deposited under the last chunk's MFMAs and the output rounds;
//   * 16-channel chunks in TWO V buffers (2 x 32 KB): while the MFMAs of chunk k run, every thread transforms its share of chunk k + 1 (one row
//     of B^T d B for one (tile, 4 channels): 8 LDS reads, 8 vector adds, 4 LDS writes, spread over the MFMA groups) — one barrier per chunk, the
//     transform in the matrix pipe's shadow;
//   * wave (wi, nh) owns row wi of the 4 x 4 transformed positions for 64 of the 128 output channels: D_xi[32 tiles x 64] += V_xi [32 x 16] .
//     U_xi [16 x 64] for xi = 4 wi .. 4 wi + 3, the transformed filter U streamed from L2 in fragment order (clc_filter_wino: 16 / 9 of the
//     filter's size, packed per step like the halo kernel's image), one group of 16 MFMAs ahead, pinned by sched_barrier;
//   * output transform: the column half (M A) in registers — a wave holds a whole row of M —, the row half through LDS, 64 channels per round,
//     two rounds; every thread finishes the 2 x 2 pixels of one (tile, 4 channels) through the epilogue: the LEAN form (conv_common.h: bias,
//     LeakyReLU / ReLU, residual, saved pre-activation, consumer-side gate, PixelShuffle — what these layers use) or the shared general one.
// Measured (tools/bench_wino.py with a -DWINO_AB=32 build: s_memtime stamps of one item): 9.3 k cycles per chunk against 8.2 k of MFMA issue,
// ~17 k for the last chunk + the wait for the slowest wave, 2 x 4 k for the output rounds — 93 k per item, 65.5 k of them MFMA.
// conv_wino64_kernel (below) is the 64-wide instantiation for the 64 -> 64 layers: two workgroups of 4 waves per CU.
// Input channels beyond 128 (data gradients of the sub-pixel convolutions: 512) pass through the halo 128 at a time.
// ANOTHER summation order than the direct kernels (transformed domain): the host side hands the transformed filter (clc_conv_desc.w_wino) to
// TRAINING launches only — recorded forwards and data gradients; eval forwards, the parity measurement and the codec keep the direct kernels' bits.
#include "common.h"

namespace {
#include "conv_common.h"

constexpr int TH = 8, TW = 16, HW = TW + 2, HPIX = (TH + 2) * HW;   // output tile, halo
constexpr int HALO_FLOATS = HPIX * 128;                              // 92 160 B
constexpr int PIECES = HPIX / 2;                                      // 1-KB LDS-DMA pieces of the halo (two pixels each)

struct WinoParams {
  ConvParams c;
  const float* u;        // transformed filter, fragment order (clc_filter_wino)
  int items, ntn, ncg;   // (pixel tile, n-tile) pairs; n-tiles of 128 output channels; groups of 128 input channels
  unsigned u_bytes;
};

#ifndef WINO_AB
#define WINO_AB 0   // timing builds of conv_wino_kernel (results WRONG for 1..8): 1 no MFMAs, 2 one output round of two, 4 no filter loads, 8 no transform-ahead, 32 phase time stamps
#endif
#if WINO_AB & 32   // phase time stamps of workgroup 0 / wave 0's second item (tools/bench_wino.py prints them through clc_wino_debug)
__device__ unsigned long long wino_dbg[32];
#define WINO_STAMP(k) do { if (blockIdx.x == 0 && tid == 0 && it == i0 + 1) wino_dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WINO_STAMP(k) do { } while (0)
#endif
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int VB_FLOATS = 16 * 32 * 16;   // one 16-channel V buffer: 32 768 B

template <bool SHUF>
__global__ __launch_bounds__(512, 1) void conv_wino_kernel(const WinoParams wp) {
  const ConvParams& p = wp.c;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;                  // [HPIX][128], 16-B chunk c of halo pixel (hy, hx) in slot c ^ (((hx >> 1) & 3) << 2)
  float* bias_s = smem + HALO_FLOATS + 2 * VB_FLOATS;   // [128]: the n-tile's bias
  float* VV = smem + HALO_FLOATS;      // 2 x [16][32][16]: chunk q of row t in slot q ^ ((t >> 2) & 3); the output rounds' [16][32][32] image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int tiles_w = p.W / TW, tiles_h = p.H / TH;
  const int KCT = wp.ncg * 4;          // 32-channel chunks of the packed filter
  const int NC16 = wp.ncg * 8;         // 16-channel chunks in all
  const int i0 = (int)((long)wp.items * blockIdx.x / gridDim.x), i1 = (int)((long)wp.items * (blockIdx.x + 1) / gridDim.x);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp.u), 0, wp.u_bytes, 0x00020000);
  // input-transform role: (16-B chunk q, tile t, row `part` of B^T d B — wave-uniform); output-transform role: (chunk qo of a 32-channel block, tile, row)
  const int q = tid & 3, t = (tid >> 2) & 31, part = tid >> 7;
  const int ty = t >> 3, tx = t & 7;
  // MFMA role: wave (wi, nh) owns row wi of the 4 x 4 transformed positions (xi = 4 wi + j) for output blocks 2 nh, 2 nh + 1 of the n-tile — a
  // whole row, so that the column half of the output transform (M A) happens in registers before anything meets in LDS
  const int wi = wave >> 1, nh = wave & 1;
  // output role: (16-B chunk qo of the round's 64 channels, tile to): all four pixels of the tile
  const int qo = tid & 15, to = tid >> 4;
  // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: row `part` = d[ra] + sg d[rb]
  const int ra = part == 0 ? 0 : (part == 2 ? 2 : 1), rb = part == 0 ? 2 : (part == 1 ? 2 : (part == 2 ? 1 : 3));
  const float sg = part == 1 ? 1.f : -1.f;
  // halo float offsets of this thread's 2 x 4 patch: pixel (2 ty + ra | rb, 2 tx + c) = hbase + c * 128 (+ hdelta for the second row); the chunk
  // swizzle ((hx >> 1) & 3) << 2 takes two values over c = 0..3
  const int hbase = ((2 * ty + ra) * HW + 2 * tx) * 128, hdelta = (rb - ra) * HW * 128;
  const int sw0 = (tx & 3) << 2, sw1 = ((tx + 1) & 3) << 2;
  const int vdst = (part * 4 * 32 + t) * 16 + ((q ^ ((t >> 2) & 3)) << 2);
  int afo[2];
#pragma unroll
  for (int t8 = 0; t8 < 2; ++t8) afo[t8] = li * 16 + (((2 * t8 + h) ^ ((li >> 2) & 3)) << 2);
  const unsigned b_lane = (unsigned)lane * 16u;
  const bool lean = epilogue_is_lean(p);     // (uniform: which form of the epilogue the output rounds take)

  // LDS-DMA of (pixel tile pt, channel group cg)'s halo: 90 one-KB pieces (two pixels each) over the 8 waves.  The halo pixel of a piece advances by
  // 16 per step of a wave (HW = 18 > 16: at most one row wrap) — no division in the loop: its VALU work delays the MFMAs it is issued between.
  auto deposit = [&](int pt, int cg) {
    const int txx = pt % tiles_w, t2 = pt / tiles_w, tyy = t2 % tiles_h, n = t2 / tiles_h;
    const int oy0 = tyy * TH, ox0 = txx * TW;
    const int org = ((n * p.H + oy0 - 1) * p.W + ox0 - 1) * p.ldx + cg * 128;
    int hy = 0, hx = 2 * wave + h;
#pragma unroll 1
    for (int pc = wave; pc < PIECES; pc += 8) {
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int c = li ^ (((hx >> 1) & 3) << 2);
      dma16(xr, halo + pc * 256, ok ? (unsigned)(org + (hy * p.W + hx) * p.ldx + c * 4) * 4u : kOOB);
      hx += 16;
      if (hx >= HW) { hx -= HW; ++hy; }
    }
  };

  int res_pt = -1, res_cg = -1;            // the halo resident (or in flight) in LDS
  for (int it = i0; it < i1; ++it) {
    const int pt = it / wp.ntn, nt = it - pt * wp.ntn;
    const int txx = pt % tiles_w, t2 = pt / tiles_w, tyy = t2 % tiles_h, n = t2 / tiles_h;
    const int oy0 = tyy * TH, ox0 = txx * TW;
    f32x16 acc[4][2];                                 // [column j][block nbi]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int nbi = 0; nbi < 2; ++nbi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][nbi][r] = 0.f;
    f32x4 bq[2][2][2];
    auto load_b = [&](int slot, int jj, int c16) {    // xi = 4 wi + jj, blocks 2 nh + nbi, 16-channel chunk c16 (global)
      const int kcg = c16 >> 1, t8g = 2 * (c16 & 1);
#pragma unroll
      for (int nbi = 0; nbi < 2; ++nbi) {
        const unsigned base = (unsigned)(((((nt * 16 + 4 * wi + jj) * KCT + kcg) * 4 + 2 * nh + nbi) * 4 + t8g)) * 1024u;
#pragma unroll
        for (int t8 = 0; t8 < 2; ++t8) bq[slot][nbi][t8] = buf_load4(ur, b_lane + base + (unsigned)(t8 * 1024));
      }
    };
    const float bias_v = (p.bias && tid < 128) ? p.bias[nt * 128 + tid] : 0.f;   // this n-tile's bias: to LDS behind the first barrier, read by the output rounds
    load_b(0, 0, 0);
    WINO_STAMP(0);

    for (int cg = 0; cg < wp.ncg; ++cg) {
      bool landed = cg == 0;                // (a prefetch for the next ITEM is waited for in front of the output rounds, see below)
      if (pt != res_pt || cg != res_cg) {   // block-uniform: not prefetched (the first item of this workgroup)
        res_pt = pt; res_cg = cg;
        deposit(pt, cg);
        landed = false;
      }
      if (!landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      WINO_STAMP(1);       // the halo has landed (everyone's pieces); V is free (previous item's output rounds / previous group's last MFMAs)
      if (cg == 0 && tid < 128) bias_s[tid] = bias_v;
      {                      // chunk 0 of this group, not overlapped
        f32x4 T[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* src = halo + hbase + c * 128 + ((q ^ (c < 2 ? sw0 : sw1)) << 2);
          T[c] = *reinterpret_cast<const f32x4*>(src) + sg * *reinterpret_cast<const f32x4*>(src + hdelta);
        }
        float* dst = VV + vdst;
        *reinterpret_cast<f32x4*>(dst) = T[0] - T[2];
        *reinterpret_cast<f32x4*>(dst + 1 * 32 * 16) = T[1] + T[2];
        *reinterpret_cast<f32x4*>(dst + 2 * 32 * 16) = T[2] - T[1];
        *reinterpret_cast<f32x4*>(dst + 3 * 32 * 16) = T[1] - T[3];
      }
      WINO_STAMP(2);
#pragma unroll 1
      for (int kc = 0; kc < 8; ++kc) {
        __syncthreads();
        WINO_STAMP(3 + kc);     // chunk kc's V buffer is complete; everyone is done with the other one (MFMAs of chunk kc - 1) and, at kc = 7, with the halo
        const float* Vc = VV + (kc & 1) * VB_FLOATS;
        float* Vn = VV + ((kc + 1) & 1) * VB_FLOATS;
        const int c16 = cg * 8 + kc;
        f32x4 aq[2][2];
        f32x4 da[4], T[4];
#pragma unroll
        for (int t8 = 0; t8 < 2; ++t8) aq[0][t8] = *reinterpret_cast<const f32x4*>(Vc + (4 * wi) * 512 + afo[t8]);
        int npt = -1, ncg2 = 0;
        if (kc == 7) {       // the last chunk of the group: the halo is free -> the next group's / the next item's, under these MFMAs and the output rounds
          npt = pt; ncg2 = cg + 1;
          if (ncg2 == wp.ncg) { ncg2 = 0; npt = it + 1 < i1 ? (it + 1) / wp.ntn : -1; }
          if (npt >= 0 && (npt != res_pt || ncg2 != res_cg)) { res_pt = npt; res_cg = ncg2; } else npt = -1;
          if (npt >= 0) deposit(npt, ncg2);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                 // group g = column j of this wave's row: 16 MFMAs, the next group's operands in flight
          const int cur = g & 1;
          if (!(WINO_AB & 4)) {
            if (g < 3) load_b(cur ^ 1, g + 1, c16);
            else load_b(cur ^ 1, 0, c16 + 1 < NC16 ? c16 + 1 : 0);
          }
          if (g < 3) {
#pragma unroll
            for (int t8 = 0; t8 < 2; ++t8) aq[cur ^ 1][t8] = *reinterpret_cast<const f32x4*>(Vc + (4 * wi + g + 1) * 512 + afo[t8]);
          }
          // the transform of chunk kc + 1 into the other buffer, a third per group.  (kc = 7: a throw-away pass over chunk 0 — nobody reads that
          // buffer before the next group's first transform / the output rounds overwrite it — instead of a branch around each part.)
          if (WINO_AB & 8) {
          } else if (g == 0) {       // (first patch row under group 0, second row + the row transform under group 1: 16 registers less across the boundary)
            const int cb = (((kc + 1) & 7) * 4 + q);
#pragma unroll
            for (int c = 0; c < 4; ++c) da[c] = *reinterpret_cast<const f32x4*>(halo + hbase + c * 128 + ((cb ^ (c < 2 ? sw0 : sw1)) << 2));
          } else if (g == 1) {
            const int cb = (((kc + 1) & 7) * 4 + q);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 dbv = *reinterpret_cast<const f32x4*>(halo + hbase + hdelta + c * 128 + ((cb ^ (c < 2 ? sw0 : sw1)) << 2));
              T[c] = da[c] + sg * dbv;
            }
          } else if (g == 2) {
            float* dst = Vn + vdst;
            *reinterpret_cast<f32x4*>(dst) = T[0] - T[2];
            *reinterpret_cast<f32x4*>(dst + 1 * 32 * 16) = T[1] + T[2];
            *reinterpret_cast<f32x4*>(dst + 2 * 32 * 16) = T[2] - T[1];
            *reinterpret_cast<f32x4*>(dst + 3 * 32 * 16) = T[1] - T[3];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t8 = 0; t8 < 2; ++t8)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
#pragma unroll
              for (int nbi = 0; nbi < 2; ++nbi) {
                if (!(WINO_AB & 1)) acc[g][nbi] = MFMA(aq[cur][t8][ss], bq[cur][nbi][t8][ss], acc[g][nbi]);
                else acc[g][nbi][(t8 * 4 + ss) & 15] += aq[cur][t8][ss] * bq[cur][nbi][t8][ss];
              }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Columns first, in registers (this wave holds row wi of M): c_b = (M A)[wi][b];
    // then the rows meet through LDS — [wi][b][tile][64 channels] = 64 KB (both V buffers) per round, two rounds (blocks nbi = 0, 1 of every wave)
    // (the next item's halo, in flight since the last chunk: waited for HERE — behind the output rounds it would sit behind their stores, which
    //  count in vmcnt too, and the wait would be for the stores' write acknowledgements)
    WINO_STAMP(11);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WINO_STAMP(12);
#pragma unroll 1
    for (int rd = 0; rd < ((WINO_AB & 2) ? 1 : 2); ++rd) {
      const int tyo = to >> 3, txo = to & 7;
      const int co = nt * 128 + ((qo >> 3) * 2 + rd) * 32 + (qo & 7) * 4;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (co - nt * 128));
      __syncthreads();       // V / the previous round's image is free
      WINO_STAMP(13 + 3 * rd);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* dst = VV + ((wi * 2 + b) * 32) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = rd == 0 ? acc[0][0][r] : acc[0][1][r], m1 = rd == 0 ? acc[1][0][r] : acc[1][1][r];
          const float m2 = rd == 0 ? acc[2][0][r] : acc[2][1][r], m3 = rd == 0 ? acc[3][0][r] : acc[3][1][r];
          const float cv = b == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;           // (bit 2 of the row = h: the two half-waves land in different bank halves)
          dst[row * 64 + ((nh * 32 + li) ^ (h << 5))] = cv;
        }
      }
      __syncthreads();
      WINO_STAMP(14 + 3 * rd);
      {
        const int col = (qo << 2) ^ (((to >> 2) & 1) << 5);
        f32x4 y[2][2];       // [output row a][output column b]: rows a .. a + 2 of c, signs + + / - -
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x4 c[4];
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) c[ii] = *reinterpret_cast<const f32x4*>(VV + ((ii * 2 + b) * 32 + to) * 64 + col);
          y[0][b] = (c[0] + c[1]) + c[2];
          y[1][b] = (c[1] - c[2]) - c[3];
        }
        const int mrow0 = (n * p.OH + oy0 + 2 * tyo) * p.OW + ox0 + 2 * txo;
        if (lean) {
          if (!SHUF) {
#pragma unroll 1
            for (int a = 0; a < 2; ++a) {     // (a pixel pair at a time: all four would not fit beside the accumulators)
              const size_t pix = (size_t)(mrow0 + a * p.OW);
              const f32x4 ya = a ? y[1][0] : y[0][0], yb = a ? y[1][1] : y[0][1];
              const Lean4In ia = lean_load4(p, pix, co), ib = lean_load4(p, pix + 1, co);
              lean_finish4(p, bv, ya, ia, pix, co);
              lean_finish4(p, bv, yb, ib, pix + 1, co);
            }
          } else {
#pragma unroll 1
            for (int a = 0; a < 2; ++a) {
              const f32x4 ya = a ? y[1][0] : y[0][0], yb = a ? y[1][1] : y[0][1];
              lean_store_shuffle4(p, bv, ya, n, oy0 + 2 * tyo + a, ox0 + 2 * txo, co);
              lean_store_shuffle4(p, bv, yb, n, oy0 + 2 * tyo + a, ox0 + 2 * txo + 1, co);
            }
          }
        } else if (!SHUF) {
#pragma unroll 1
          for (int a = 0; a < 2; ++a) {
            const size_t pix = (size_t)(mrow0 + a * p.OW);
            const f32x4 ya = a ? y[1][0] : y[0][0], yb = a ? y[1][1] : y[0][1];
            const Epi4In ia = epilogue_load4(p, pix, co), ib = epilogue_load4(p, pix + 1, co);
            epilogue_finish4(p, bv, ya, ia, pix, co);
            epilogue_finish4(p, bv, yb, ib, pix + 1, co);
          }
        } else {
#pragma unroll 1
          for (int a = 0; a < 2; ++a)
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
              epilogue_store(p, a ? y[1][0][e] : y[0][0][e], bv[e], mrow0 + a * p.OW, co + e, p.OH, p.OW, 0, 0);
              epilogue_store(p, a ? y[1][1][e] : y[0][1][e], bv[e], mrow0 + a * p.OW + 1, co + e, p.OH, p.OW, 0, 0);
            }
        }
      }
      WINO_STAMP(15 + 3 * rd);
    }
    // (no barrier here: the next item waits for its halo and meets everyone at the barrier in front of its first transform, which is the
    //  first write to V after this round's reads)
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
  float* halo = smem;                  // [HPIX][64], 16-B chunk c of halo pixel (hy, hx) in slot c ^ (((hx >> 1) & 3) << 2)
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
  const int qo = tid & 15;             // (output role: chunk qo of the 64 channels, tiles (tid >> 4) and + 16)
  const bool lean = epilogue_is_lean(p);
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
        int hy = 0, hx = 4 * wave + (lane >> 4);    // (the halo pixel advances by 16 per step, HW = 18: no division in the loop)
#pragma unroll 1
        for (int pc = wave; pc < PIECES2; pc += 4) {
          const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const int c = (lane & 15) ^ (((hx >> 1) & 3) << 2);
          dma16(xr, halo + pc * 256, ok ? (unsigned)(org + (hy * p.W + hx) * p.ldx + c * 4) * 4u : kOOB);
          hx += 16;
          if (hx >= HW) { hx -= HW; ++hy; }
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
              d[r][c] = *reinterpret_cast<const f32x4*>(halo + (hy * HW + hx) * 64 + ((cidx ^ (((hx >> 1) & 3) << 2)) << 2));
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
    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Columns first, in registers (wave w holds row w of M): c_b = (M A)[w][b]; then the
    // rows meet through LDS in ONE round — [w][b][tile][64 channels] = 64 KB over the halo and V (the next item re-deposits the halo)
    cur_pt = -1;
    {
      const int co = nt * 64 + (qo >> 3) * 32 + (qo & 7) * 4;
      const f32x4 bv = epilogue_bias4(p.bias, co);
      __syncthreads();       // the last chunk's MFMAs have read V, its transform the halo
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* dst = S + ((wave * 2 + b) * 32) * 64;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float cv = b == 0 ? (acc[0][nb][r] + acc[1][nb][r]) + acc[2][nb][r] : (acc[1][nb][r] - acc[2][nb][r]) - acc[3][nb][r];
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;           // (bit 2 of the row = h: the two half-waves land in different bank halves)
            dst[row * 64 + ((nb * 32 + li) ^ (h << 5))] = cv;
          }
      }
      __syncthreads();
#pragma unroll 1
      for (int tt = 0; tt < 2; ++tt) {     // this thread's two tiles
        const int to = (tid >> 4) + 16 * tt, tyo = to >> 3, txo = to & 7;
        const int col = (qo << 2) ^ (((to >> 2) & 1) << 5);
        f32x4 y[2][2];       // [output row a][output column b]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x4 c[4];
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) c[ii] = *reinterpret_cast<const f32x4*>(S + ((ii * 2 + b) * 32 + to) * 64 + col);
          y[0][b] = (c[0] + c[1]) + c[2];
          y[1][b] = (c[1] - c[2]) - c[3];
        }
        const int mrow0 = (n * p.OH + oy0 + 2 * tyo) * p.OW + ox0 + 2 * txo;
#pragma unroll 1
        for (int a = 0; a < 2; ++a) {
          const size_t pix = (size_t)(mrow0 + a * p.OW);
          const f32x4 ya = a ? y[1][0] : y[0][0], yb = a ? y[1][1] : y[0][1];
          if (lean) {
            if (!SHUF) {
              const Lean4In ia = lean_load4(p, pix, co), ib = lean_load4(p, pix + 1, co);
              lean_finish4(p, bv, ya, ia, pix, co);
              lean_finish4(p, bv, yb, ib, pix + 1, co);
            } else {
              lean_store_shuffle4(p, bv, ya, n, oy0 + 2 * tyo + a, ox0 + 2 * txo, co);
              lean_store_shuffle4(p, bv, yb, n, oy0 + 2 * tyo + a, ox0 + 2 * txo + 1, co);
            }
          } else if (!SHUF) {
            const Epi4In ia = epilogue_load4(p, pix, co), ib = epilogue_load4(p, pix + 1, co);
            epilogue_finish4(p, bv, ya, ia, pix, co);
            epilogue_finish4(p, bv, yb, ib, pix + 1, co);
          } else {
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
              epilogue_store(p, ya[e], bv[e], (int)pix, co + e, p.OH, p.OW, 0, 0);
              epilogue_store(p, yb[e], bv[e], (int)pix + 1, co + e, p.OH, p.OW, 0, 0);
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
  // Which instantiation: the 128-wide kernel (one workgroup per CU) where the launch has >= 192 of its items; else the 64-wide one (two per CU, half
  // the output channels per item -> twice the items) where the channels are multiples of 64 and that makes >= 128 items.  (The rule may look at the
  // batch: these kernels serve training launches only — an image's bits in a recorded pass need not be the bits of another batch.)
  const int ptiles = p.N * (p.H / TH) * (p.W / TW);
  const bool can128 = p.Cin % 128 == 0 && p.Cout % 128 == 0 && !(mode & 8);   // (bit 3: the 64-wide kernel on every layer — an experiment switch)
  const bool wide = can128 && ptiles * (p.Cout / 128) >= 192;
  if (!wide && (!(mode & 4) || ptiles * (p.Cout / 64) < 128)) return 0;        // bit 2: the 64-wide kernel
  const int nw = wide ? 128 : 64;
  WinoParams wp;
  wp.c = p;
  wp.u = u;
  wp.ntn = p.Cout / nw;
  wp.ncg = p.Cin / nw;
  wp.items = ptiles * wp.ntn;
  wp.u_bytes = (unsigned)ub;
  static PerDeviceOnce once[2];
  const int k = shuf ? 1 : 0;
  if (wide) {
    const int grid = wp.items < 256 ? wp.items : 256;
    const int lds = (HALO_FLOATS + 2 * VB_FLOATS + 128) * 4;
    static PerDeviceOnce oncep[2];
    if (shuf) {
      if (oncep[0].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(conv_wino_kernel<true>, dim3(grid), dim3(512), lds, st, wp);
    } else {
      if (oncep[1].first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
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

#if WINO_AB & 32
extern "C" int clc_wino_debug(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(wino_dbg), sizeof(unsigned long long) * 32);
}
#endif
