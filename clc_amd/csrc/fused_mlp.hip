// fused_mlp.hip — the MLP of a Swin block (Linear(64 -> 256), GELU, Linear(256 -> 64), + residual) as ONE launch forward and ONE launch
// for its whole data gradient, with the 256-channel hidden tensor never written by the forward pass (gfx950).
//
//   y = x_res + W2 . gelu(W1 . x + b1) + b2                      /root/reference/models/CLC_run.py:185-187, 190-192 (Block.mlp, `x + mlp(ln2(x))`)
//
// As two launches of the 1x1 convolution kernels a block on a 8 x 128 x 128 map moved 1.07 GB through HBM in the forward pass alone, 402 MB
// of it the hidden tensor and its stored GELU derivative (written by fc1, re-read by fc2 and by three gradient kernels): the layers sit at
// the chip's FLOP : byte balance point and ran at 47 % of the MFMA peak / 67 % of the achievable HBM rate.  Here the hidden tensor lives in
// the accumulator registers:
//
//   * every wave owns 32 pixels and ALL channels; there is no barrier and no LDS traffic for activations in the main loop — the waves of a
//     workgroup drift apart, so one wave's GELU (VALU) and stores run under another wave's MFMAs;
//   * the MFMA roles are swapped with respect to conv_igemm.hip: A = filter rows (channels), B = pixels.  The 32 x 32 result block of lane
//     (pixel j = lane & 31, half h = lane >> 5) then holds channels i = (r & 3) + 8 (r >> 2) + 4 h in register r, i.e. FOUR CONSECUTIVE
//     channels in four consecutive registers:
//       - the epilogues load / store 16 B per lane (dwordx4) instead of 16 dword accesses per block;
//       - register r of a finished block IS the B operand of the next GEMM's MFMA step that contracts the channel pair
//         {8 (r >> 2) + (r & 3), + 4}: the second GEMM runs straight out of the first one's accumulators (after bias + GELU in
//         registers) — no LDS round trip, no layout change;
//   * that pair order is exactly the K order of the tiled kernels (K-tile, 8-group t, step s -> pair {8 t + s, 8 t + 4 + s}), and the
//     same epilogue expressions are used, so the fused launch produces THE SAME BITS as the two-launch chain (checked by the tests): the
//     codec may use either, whatever the batch size;
//   * both filters (2 x 64 KB) are resident in LDS for the lifetime of a persistent workgroup, as slot-swizzled [K-tile][row][32] images
//     (conv_igemm_dma_kernel's layout: conflict-free ds_read_b128 fragments), deposited once by LDS-DMA.
//
// Backward (`clc_mlp_bwd`), per 32-pixel wave tile and 32-channel block hb of the hidden layer:
//   h  = W1[hb] x + b1          recomputed from the saved LayerNorm output (+ 8.6 GFLOP per 8 x 128 x 128 block; - 670 MB of traffic)
//   u  = W2^T[hb] dy
//   dh = u . gelu'(h),  g = gelu(h)      -> HBM (the dy operand of fc1's and the x operand of fc2's filter gradient, which stay on the
//                                           grouped stream-K kernels)
//   dx += W1^T[:, hb] dh                  straight from the dh registers; W1^T is read column-wise (ds_read_b32) from the W1 image
// Same K orders and epilogue expressions as the unfused data-gradient kernels -> same bits again.
#include "common.h"

namespace {

constexpr unsigned kOOB = 0x80000000u;
constexpr int CI = 64, CH = 256, CO = 64;   // the Swin blocks of the ConvTransBlocks: trans_dim 64, hidden 4 x 64

struct MlpParams {
  const float* x; const float* w1; const float* b1; const float* w2; const float* b2; const float* res; float* y;
  const float* dy; const float* w2t; float* dx; float* dh; float* g;
  float* hsave;                 // [M][256] fc1 pre-activation: written by the forward pass (optional), read instead of recomputed backward
  const float* ln_gamma; const float* ln_beta;   // LayerNorm in front of fc1 (LN instantiations): x is the block's raw input and the residual
  float* ln_out; float* ln_ws;                   // backward: LN(x) [M][64] for fc1's filter gradient; partial rows [grid][2][64] (dgamma, dbeta)
  int ldx, ldr, ldy, lddy, lddx;
  int M, tiles;
  unsigned x_bytes, res_bytes, y_bytes, dy_bytes, dx_bytes, hid_bytes;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Measured and NOT built in: starting the second wave of every SIMD a few microseconds late (s_sleep), so that a SIMD's two waves are out of step
// and one computes while the other is in its VALU / LDS / memory phase.  Under an eager kernel trace (every launch starts on an idle chip, all
// waves together) that took the LayerNorm backward launch at 8 x 128 x 128 from 164.5 to 145.8 us; replayed inside a hipGraph, back to back with
// its neighbours — how the step runs — offsets of 0 / 1.7 / 3.4 / 6.8 us all gave the same time (397 us per forward + backward of the block,
// tools/bench_mlp_ln.py) and the same step (27.26-27.30 ms): the previous launch's tail already spreads the workgroups' start times.  The same
// goes for the eager-trace ablations of the LayerNorm epilogue (column sums 11 us, x re-read 20 us per launch): 3-5 us in the graph.

// ---------------------------------------------------------------------------------------------------------------- LayerNorm in front (LN)
// `x + mlp(ln2(x))` from the block's RAW input: a lane (pixel li, half h) holds the 16-B chunks c = 8 kt + 2 t8 + h of its pixel's 64 channels
// (the B-operand layout above), i.e. the whole row of nn.LayerNorm(64) sits in two lanes.  The row sums follow layernorm_fwd_vec_kernel<16>'s
// xor-butterfly over its 16 chunk lanes (partners c ^ 8, c ^ 4, c ^ 2, c ^ 1 = kt, t8 bit 1, t8 bit 0 — all inside the lane — then h: one
// cross-lane add), and the same expressions are used, so LN(x), the row statistics and the LayerNorm gradient carry THE SAME BITS as the
// separate launches (elementwise.hip) — which batch sizes below the fusion threshold still use.
__device__ __forceinline__ float ln_row_sum(const float (&p)[2][4]) {
  float q[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) q[t8] = p[0][t8] + p[1][t8];
  float s = (q[0] + q[2]) + (q[1] + q[3]);
  s += __shfl_xor(s, 32, 64);
  return s;
}
// in place: xf <- x - mean; returns the row's mean and reciprocal standard deviation
__device__ __forceinline__ void ln_center(f32x4 (&xf)[2][4], float& mu, float& rs) {
  float pp[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) pp[kt][t8] = (xf[kt][t8][0] + xf[kt][t8][1]) + (xf[kt][t8][2] + xf[kt][t8][3]);
  mu = ln_row_sum(pp) / 64.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      const f32x4 d = xf[kt][t8] - mu;
      xf[kt][t8] = d;
      pp[kt][t8] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  rs = rsqrtf(ln_row_sum(pp) / 64.f + 1e-5f);
}
// in place: xf <- LN(xf); returns the row's mean and reciprocal standard deviation
__device__ __forceinline__ void ln_apply(f32x4 (&xf)[2][4], const float* lns, int h, float& mu, float& rs) {
  float pp[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) pp[kt][t8] = (xf[kt][t8][0] + xf[kt][t8][1]) + (xf[kt][t8][2] + xf[kt][t8][3]);
  mu = ln_row_sum(pp) / 64.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      const f32x4 d = xf[kt][t8] - mu;
      xf[kt][t8] = d;
      pp[kt][t8] = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  rs = rsqrtf(ln_row_sum(pp) / 64.f + 1e-5f);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      const int c0 = kt * 32 + 8 * t8 + 4 * h;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(lns + c0), bt = *reinterpret_cast<const f32x4*>(lns + 64 + c0);
      xf[kt][t8] = xf[kt][t8] * rs * gm + bt;
    }
}
// Column sums over a tile's 32 pixels of one 32-channel block held in the accumulator layout, added to *acc of lane (channel cc = lane & 31 of
// the block, pixel parity hh = lane >> 5): through the wave's 2 KB scratch, 16 pixels at a time, in a fixed order.  (A DPP butterfly per
// register — quad permutes, row mirrors, one swizzle: 320 VALU instructions per tile, no LDS round trip — measured SLOWER: 166 vs 156 us per
// launch at 8 x 128 x 128; f32 MFMAs do not hide VALU work and the waves of a launch run in step.)
__device__ __forceinline__ void colsum_block(float* scratch, const f32x4 (&v)[4], float* acc, int lane, int li, int h) {
  const int cc = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if ((li >> 4) == rd) {
      float* row = scratch + ((li & 15) << 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(row + ((((2 * q + h) ^ (li & 7))) << 2)) = v[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pl = 2 * k + hh;
      t += scratch[(pl << 5) + ((((cc >> 2) ^ (pl & 7))) << 2) + (cc & 3)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *acc += t;
  }
}


// The LayerNorm gradient as the epilogue of a data-gradient launch: dxacc = the gradient of LN's output in the accumulator layout, xh = the raw
// x of the tile's pixels in the same layout (destroyed), add = the residual branch's gradient.  layernorm_bwd_vec_kernel<16>'s expressions and
// row-sum order (-> its bits); the tile's dgamma / dbeta column sums are added to the lane-private slots `wacc`.
template <bool NO_COLSUM>
__device__ __forceinline__ void ln_bwd_epilogue(const f32x16 (&dxacc)[2], f32x4 (&xh)[2][4], const f32x4 (&add)[2][4], const float* lns, float* scratch,
                                                float* wacc, __amdgpu_buffer_rsrc_t dxr, unsigned pix, unsigned lddx, int lane, int h) {
  f32x4 dv[2][4];
  float p1[2][4], p2[2][4];
  float mu, rs;
  ln_center(xh, mu, rs);
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(lns + ib * 32 + 8 * q + 4 * h);
#pragma unroll
      for (int s = 0; s < 4; ++s) dv[ib][q][s] = dxacc[ib][4 * q + s] + 0.f;
      xh[ib][q] = xh[ib][q] * rs;   // (x - mean) * rstd
      const f32x4 g = dv[ib][q] * gm;
      p1[ib][q] = (g[0] + g[1]) + (g[2] + g[3]);
      p2[ib][q] = (g[0] * xh[ib][q][0] + g[1] * xh[ib][q][1]) + (g[2] * xh[ib][q][2] + g[3] * xh[ib][q][3]);
    }
  const float s1 = ln_row_sum(p1) / 64.f, s2 = ln_row_sum(p2) / 64.f;
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(lns + ib * 32 + 8 * q + 4 * h);
      const f32x4 g = dv[ib][q] * gm;
      f32x4 o = rs * (g - s1 - xh[ib][q] * s2);
      o += add[ib][q];   // the residual branch's gradient
      st4(o, dxr, (pix * lddx + (unsigned)(ib * 32 + 8 * q + 4 * h)) * 4u);
    }
  if (NO_COLSUM) return;
  int ln_ = lane;
  asm volatile("" : "+v"(ln_));   // (addresses derived from here are recomputed per tile: hoisted out of the tile loop they spill)
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    colsum_block(scratch, dv[ib], wacc + (2 + ib) * 64, ln_, ln_ & 31, ln_ >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q) xh[ib][q] = dv[ib][q] * xh[ib][q];
    colsum_block(scratch, xh[ib], wacc + ib * 64, ln_, ln_ & 31, ln_ >> 5);
  }
}
// the workgroup's row [2][64] of dgamma / dbeta partial sums from its waves' slots, in a fixed order (call after a barrier)
template <int NW>
__device__ __forceinline__ void ln_ws_row(const float* slots, float* ln_ws, int tid) {
  if (tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    const float* slot = slots + (2 * which + (c >> 5)) * 64 + (c & 31);   // wave 0, parity 0
    float tsum = 0.f;
    for (int w = 0; w < NW; ++w) {
      tsum += slot[w * 256];
      tsum += slot[w * 256 + 32];
    }
    ln_ws[((size_t)blockIdx.x * 2 + which) * 64 + c] = tsum;
  }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// Measured on the way (8 x 128 x 128, one launch, graph-replayed; tools/bench_mlp.py with CLC_TUNING=12:x):
//   * f32 MFMAs and VALU instructions do NOT overlap on this hardware (the f32 matrix rate IS the vector rate): the GELU's ~19 VALU
//     instructions per value add their full time (20 us of 111) whether they are interleaved with the MFMAs or not, with one or two
//     waves per SIMD.  A software pipeline that issued the next block's fc1 MFMAs between the GELU of the current one was SLOWER
//     (111.7 vs 98.6 us) than the plain loop: nothing to hide, more registers and hazards;
//   * consecutive MFMAs on ONE accumulator pay the dependent-issue latency (a block of 32 dependent 32x32x2 MFMAs ran at ~83 % of the
//     independent rate), so two hidden blocks are computed side by side and every MFMA alternates between two accumulators;
//   * LDS fragment reads cost nothing measurable (ablated: 105 vs 111 us).
// SAVE: store fc1's pre-activation (training, save mode).  PK: GELU on the packed-f32 instructions (gelu_parts2; same bits).
// LN: LayerNorm in front, computed in registers from the raw input, which is also the residual (p.res is not read).
template <int NW, int ABL = 0, bool SAVE = false, bool PK = true, bool LN = false>   // ABL: timing diagnostics of CLC_TUNE_ABLATE (results WRONG): 1 = no GELU arithmetic, 2 = no fc1 MFMAs, 4 = no LDS fragment reads
__global__ __launch_bounds__(64 * NW, 2) void mlp_fwd_kernel(const MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1s = smem;                    // [2][256][32]  fc1: rows = hidden, K = input channels
  float* W2s = W1s + 2 * CH * 32;       // [8][64][32]   fc2: rows = output channels, K = hidden
  float* b1s = W2s + 8 * CO * 32;       // [256]
  float* b2s = b1s + CH;                // [64]
  float* lns = b2s + CO;                // [2][64]  LayerNorm gamma, beta (LN)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = lns + 2 * CI + wave * 512;   // 2 KB per wave (LN, training): store_block_lines
  fill_image<NW>(W1s, p.w1, CH, CI, wave, lane);
  fill_image<NW>(W2s, p.w2, CO, CH, wave, lane);
  for (int i = tid; i < CH; i += 64 * NW) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  for (int i = tid; i < CO; i += 64 * NW) b2s[i] = p.b2 ? p.b2[i] : 0.f;
  if (LN) for (int i = tid; i < 2 * CI; i += 64 * NW) lns[i] = i < CI ? p.ln_gamma[i] : p.ln_beta[i - CI];

  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.x_bytes), rr = srd(p.res ? p.res : p.x, p.res ? p.res_bytes : p.x_bytes), yr = srd(p.y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t hr = srd(SAVE ? p.hsave : p.y, SAVE ? p.hid_bytes : p.y_bytes);
  const __amdgpu_buffer_rsrc_t lnr = srd(LN && p.ln_out ? p.ln_out : p.y, LN && p.ln_out ? (unsigned)p.M * CI * 4u : p.y_bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* W1l = W1s + (li << 5);   // this lane's row inside a 32-row block of an image
  const float* W2l = W2s + (li << 5);
  const float* b1l = b1s + 4 * h;
  auto load_x = [&](int t, f32x4 (&xf)[2][4]) {   // a tile past the end reads zeros through the SRD's range check
    const int p0 = (t * NW + wave) * 32;
    const unsigned xo = ((unsigned)(p0 + li) * (unsigned)p.ldx + 4u * h) * 4u;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = ld4(xr, p0 < p.M ? xo + (unsigned)(kt * 32 + 8 * t8) * 4u : kOOB);
  };
  f32x4 xf[2][4], xn[2][4];
  load_x(blockIdx.x, xn);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the filter images' DMA pieces and the first pixels)
  __syncthreads();

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
    const unsigned pix = (unsigned)(p0 + li);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = xn[kt][t8];
    load_x(t + gridDim.x, xn);          // the next tile's pixels: in flight under this tile's 512 MFMAs
    if (p0 >= p.M) continue;            // wave-uniform; there is no barrier below
    f32x4 xraw[2][4];
    if (LN) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) xraw[kt][t8] = xf[kt][t8];
      float mu, rs;
      ln_apply(xf, lns, h, mu, rs);
      if (p.ln_out) {   // training: LN(x) [M][64] dense for the backward launch (fc1's operand again there, and the x operand of its filter gradient)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) store_block_lines(scratch, xf[kt], lnr, (unsigned)p0, (unsigned)CI, (unsigned)(kt * 32), lane, li, h);
      }
    }
    f32x16 yacc[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.f;

#pragma unroll 1
    for (int hp = 0; hp < 4; ++hp) {    // two 32-channel blocks of the hidden layer side by side
      f32x16 hacc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[b][r] = 0.f;
      const float* w1b = W1l + ((hp * 64) << 5);
#pragma unroll
      for (int j = 0; j < 8; ++j) {       // K-steps (kt, t8) in the tiled kernels' order
        f32x4 a[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) a[b] = (ABL & 4) ? xf[b][j & 3] : *reinterpret_cast<const f32x4*>(w1b + (((j >> 2) * CH + b * 32) << 5) + fo[j & 3]);
        if (ABL & 2) { asm volatile("" ::"v"(a[0]), "v"(a[1])); continue; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) hacc[b] = MFMA(a[b][s], xf[j >> 2][j & 3][s], hacc[b]);
      }
      // bias + GELU in the accumulator registers (register 4 q + s of block b = hidden channel (2 hp + b) * 32 + 8 q + 4 h + s)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(b1l + (2 * hp + b) * 32 + 8 * q);
          f32x4 v;
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s] = hacc[b][4 * q + s] + bq[s];
          if (SAVE) st4(v, hr, (pix * (unsigned)CH + (unsigned)((2 * hp + b) * 32 + 8 * q + 4 * h)) * 4u);   // (training, save mode)
#pragma unroll
          for (int s = 0; s < 4; s += 2) {
            if (ABL & 1) { hacc[b][4 * q + s] = v[s]; hacc[b][4 * q + s + 1] = v[s + 1]; continue; }
            if (PK) {
              f32x2 cdf, pdf;
              const f32x2 vv = {v[s], v[s + 1]};
              gelu_parts2(vv, cdf, pdf);
              const f32x2 gg = vv * cdf;
              hacc[b][4 * q + s] = gg[0]; hacc[b][4 * q + s + 1] = gg[1];
            } else {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float cdf, pdf;
                gelu_parts(v[s + e], cdf, pdf);
                hacc[b][4 * q + s + e] = v[s + e] * cdf;
              }
            }
          }
        }
      // fc2: K-tiles 2 hp, 2 hp + 1 of the hidden layer; B operands = the registers just computed
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* w2b = W2l + (((2 * hp + b) * CO) << 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 a[2];
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) a[ob] = (ABL & 4) ? xf[ob][q] : *reinterpret_cast<const f32x4*>(w2b + ((ob * 32) << 5) + fo[q]);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) yacc[ob] = MFMA(a[ob][s], hacc[b][4 * q + s], yacc[ob]);
        }
      }
    }
    // epilogue: y = (acc + b2) + 1 * res, 16 B per lane
    f32x4 rv[2][4];
    if (LN) {   // (the accumulator layout of output block ob, quad q IS the operand layout of K-tile ob, group q: the residual is in registers)
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[ob][q] = xraw[ob][q];
    } else if (p.res) {
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[ob][q] = ld4(rr, (pix * (unsigned)p.ldr + (unsigned)(ob * 32 + 8 * q + 4 * h)) * 4u);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = ob * 32 + 8 * q + 4 * h;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(b2s + c0);
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = yacc[ob][4 * q + s] + bq[s];
        if (LN || p.res) {
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s] += 1.0f * rv[ob][q][s];
        }
        st4(v, yr, (pix * (unsigned)p.ldy + (unsigned)c0) * 4u);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------- backward
// LOADH: the forward pass saved the fc1 pre-activation (clc_mlp_desc.h): it is read back (134 MB per 8 x 128 x 128 block, on a kernel that is
// MFMA / VALU-bound either way) instead of recomputed (256 of a tile's 768 MFMAs).
// Two hidden blocks side by side, as forward: consecutive MFMAs never share an accumulator.
// LN: x is the block's raw input and ln_out the LN(x) the forward launch stored (fc1's operand here); dx is the gradient of the whole
// `x + mlp(LN(x))` — layernorm_bwd_vec_kernel<16>'s expressions on the accumulator registers (row statistics again from x), dy added as
// the residual's gradient — and the per-workgroup column sums for dgamma / dbeta go to ln_ws.
template <int NW, bool LOADH, bool LN = false, int ABL = 0>   // ABL (timing diagnostics, results WRONG): 1 = no dgamma / dbeta column sums, 4 = x not read again
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_kernel(const MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1s = smem;                    // [2][256][32]  fc1 filter: rows = hidden, K = input channels (also read column-wise for dx)
  float* Wts = W1s + 2 * CH * 32;       // [2][256][32]  fc2 filter transposed: rows = hidden, K = output channels
  float* b1s = Wts + 2 * CH * 32;       // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = b1s + CH + wave * 512;   // 2 KB per wave: store_block_lines
  float* lns = b1s + CH + NW * 512;         // [2][64]  LayerNorm gamma, beta (LN)
  fill_image<NW>(W1s, p.w1, CH, CI, wave, lane);
  fill_image<NW>(Wts, p.w2t, CH, CO, wave, lane);
  for (int i = tid; i < CH; i += 64 * NW) b1s[i] = p.b1 ? p.b1[i] : 0.f;
  if (LN) for (int i = tid; i < 2 * CI; i += 64 * NW) lns[i] = i < CI ? p.ln_gamma[i] : p.ln_beta[i - CI];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.x_bytes), dyr = srd(p.dy, p.dy_bytes), dxr = srd(p.dx, p.dx_bytes);
  const __amdgpu_buffer_rsrc_t dhr = srd(p.dh, p.hid_bytes), gr = srd(p.g, p.hid_bytes), hr = srd(LOADH ? p.hsave : p.g, p.hid_bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* W1l = W1s + (li << 5);
  const float* Wtl = Wts + (li << 5);
  // column reads of the W1 image for dx = W1^T dh: element (hidden row R = hb * 32 + 8 q + 4 h + s, input channel ib * 32 + li) sits at
  // ((ib * 256 + R) << 5) + (((li >> 2) ^ ((R >> 1) & 7)) << 2) + (li & 3), and (R >> 1) & 7 = (4 q + 2 h + (s >> 1)) & 7
  int co_[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      co_[q][s] = ((8 * q + 4 * h + s) << 5) + ((((li >> 2) ^ ((4 * q + 2 * h + (s >> 1)) & 7))) << 2) + (li & 3);
  const __amdgpu_buffer_rsrc_t lnr = srd(LN ? p.ln_out : p.x, p.x_bytes);   // fc1's operand
  // LN: lane-private LDS slots (the kernel has no register to spare): [0..1] this lane's share of dgamma (channel ib * 32 + (lane & 31), pixel
  // parity lane >> 5), [2..3] of dbeta
  float* wacc = lns + 2 * CI + wave * 256 + lane;
  if (LN) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wacc[k * 64] = 0.f;
  }

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
    if (p0 >= p.M) continue;            // wave-uniform; no barrier below
    const unsigned pix = (unsigned)(p0 + li);
    f32x4 xf[2][4], df[2][4];
    {
      const unsigned xo = (pix * (unsigned)p.ldx + 4u * h) * 4u, yo = (pix * (unsigned)p.lddy + 4u * h) * 4u;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) {
          if (!LOADH) xf[kt][t8] = ld4(lnr, xo + (unsigned)(kt * 32 + 8 * t8) * 4u);
          df[kt][t8] = ld4(dyr, yo + (unsigned)(kt * 32 + 8 * t8) * 4u);
        }
    }
    f32x16 dxacc[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) dxacc[ib][r] = 0.f;

    f32x4 xh[2][4];   // LN: the raw x again, for the LayerNorm gradient
    auto hp_iter = [&](const int hp, auto&& before_dx) __attribute__((always_inline)) {
      f32x16 hacc[2], uacc[2];
      const unsigned ho = (pix * (unsigned)CH + (unsigned)(hp * 64 + 4 * h)) * 4u;
      f32x4 hv[2][4];
      if (LOADH) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[b][q] = ld4(hr, ho + (unsigned)(b * 32 + 8 * q) * 4u);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hacc[b][r] = 0.f; uacc[b][r] = 0.f; }
      const int rowb = (hp * 64) << 5;
#pragma unroll
      for (int j = 0; j < 8; ++j) {       // K-steps (kt, t8) in the tiled kernels' order
        const int ko = (((j >> 2) * CH) << 5) + rowb + fo[j & 3];
        f32x4 a[2], c[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (!LOADH) a[b] = *reinterpret_cast<const f32x4*>(W1l + ko + ((b * 32) << 5));
          c[b] = *reinterpret_cast<const f32x4*>(Wtl + ko + ((b * 32) << 5));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (!LOADH) hacc[b] = MFMA(a[b][s], xf[j >> 2][j & 3][s], hacc[b]);
            uacc[b] = MFMA(c[b][s], df[j >> 2][j & 3][s], uacc[b]);
          }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 gq[4], dq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          if (LOADH) v = hv[b][q];
          else {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(b1s + hp * 64 + b * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = hacc[b][4 * q + s] + bq[s];
          }
          f32x4 gv, dv;
#pragma unroll
          for (int s = 0; s < 4; s += 2) {
            f32x2 cdf, pdf;
            const f32x2 vv = {v[s], v[s + 1]};
            gelu_parts2(vv, cdf, pdf);
            const f32x2 gg = vv * cdf, dd = cdf + vv * pdf;
            const f32x2 uu = {uacc[b][4 * q + s] + 0.f, uacc[b][4 * q + s + 1] + 0.f};   // (the unfused data-gradient epilogue adds its absent bias as 0.f first)
            const f32x2 dh2 = uu * dd;
            gv[s] = gg[0]; gv[s + 1] = gg[1];
            dv[s] = dh2[0]; dv[s + 1] = dh2[1];
            uacc[b][4 * q + s] = dh2[0]; uacc[b][4 * q + s + 1] = dh2[1];
          }
          gq[q] = gv; dq[q] = dv;
        }
        // g and dh to HBM as whole 128-B lines (the operands of the two filter gradients)
        store_block_lines(scratch, gq, gr, (unsigned)p0, (unsigned)CH, (unsigned)(hp * 64 + b * 32), lane, li, h);
        store_block_lines(scratch, dq, dhr, (unsigned)p0, (unsigned)CH, (unsigned)(hp * 64 + b * 32), lane, li, h);
      }
      before_dx();
      // dx += W1^T[:, hidden blocks 2 hp, 2 hp + 1] dh   (K-tiles in order, the two input-channel blocks alternate)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* colb = W1s + rowb + ((b * 32) << 5);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) dxacc[ib] = MFMA(colb[((ib * CH) << 5) + co_[q][s]], uacc[b][4 * q + s], dxacc[ib]);
      }
    };
    if constexpr (LN) {
      // the last round is peeled: the x loads for the LayerNorm gradient go out behind its dh / g stores and land under its 128 dx MFMAs
      // (issued after the loop they waited for the whole store queue to drain: 20 us of a 163-us launch at 8 x 128 x 128)
#pragma unroll 1
      for (int hp = 0; hp < 3; ++hp) hp_iter(hp, [] {});
      hp_iter(3, [&] {
        const unsigned xo = (pix * (unsigned)p.ldx + 4u * h) * 4u;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int q = 0; q < 4; ++q) xh[ib][q] = (ABL & 4) ? xf[ib][q] : ld4(xr, xo + (unsigned)(ib * 32 + 8 * q) * 4u);
      });
    } else {
#pragma unroll 1
      for (int hp = 0; hp < 4; ++hp) hp_iter(hp, [] {});
    }
    if (LN) {
      ln_bwd_epilogue<(ABL & 1) != 0>(dxacc, xh, df, lns, scratch, wacc, dxr, pix, (unsigned)p.lddx, lane, h);
      continue;
    }
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = dxacc[ib][4 * q + s] + 0.f;
        st4(v, dxr, (pix * (unsigned)p.lddx + (unsigned)(ib * 32 + 8 * q + 4 * h)) * 4u);
      }
  }
  if (LN) {   // the workgroup's row of partial sums, in a fixed order (every wave leaves the tile loop after the same number of rounds)
    __syncthreads();
    ln_ws_row<NW>(lns + 2 * CI, p.ln_ws, tid);
  }
}


// ---------------------------------------------------------------------------------------------------------------- LayerNorm + Linear
// ln1 and the attention's embedding (`self.embedding_layer(ln1(x))`: nn.LayerNorm(64) + nn.Linear(64, 192), /root/reference/models/CLC_run.py:120,
// 141, 180, 191) as ONE launch, and the Linear's data gradient + the LayerNorm's backward pass (+ the block's residual gradient) as one more —
// the same building blocks as above: a wave owns 32 pixels, the filter is a resident LDS image, LN in registers in the operand layout, the
// LayerNorm gradient as the epilogue on the dx accumulators, whole-line stores.  Same K order and expressions as the LayerNorm kernel followed
// by the tiled 1x1 kernels -> the same bits for qkv and dx.
struct LnLinParams {
  const float* x; const float* w; const float* b; const float* ln_gamma; const float* ln_beta;
  float* y; float* ln_out;
  const float* dy; const float* wt; float* dx; const float* dadd; float* ln_ws;
  int ldx, lddx, ldadd;
  int M, tiles;
  unsigned x_bytes, y_bytes, dx_bytes, add_bytes;
};

template <int NW, int CO_>   // CO_ output channels (a multiple of 64)
__global__ __launch_bounds__(64 * NW, 2) void lnlin_fwd_kernel(const LnLinParams p) {
  constexpr int NB = CO_ / 32;
  static_assert(NB % 2 == 0, "output blocks are computed in pairs");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [2][CO_][32]  rows = output channels, K = input channels
  float* bs = Ws + 2 * CO_ * 32;          // [CO_]
  float* lns = bs + CO_;                  // [2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = lns + 2 * CI + wave * 512;
  fill_image<NW>(Ws, p.w, CO_, CI, wave, lane);
  for (int i = tid; i < CO_; i += 64 * NW) bs[i] = p.b ? p.b[i] : 0.f;
  for (int i = tid; i < 2 * CI; i += 64 * NW) lns[i] = i < CI ? p.ln_gamma[i] : p.ln_beta[i - CI];
  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.x_bytes), yr = srd(p.y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t lnr = srd(p.ln_out ? p.ln_out : p.y, p.ln_out ? (unsigned)p.M * CI * 4u : p.y_bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* Wl = Ws + (li << 5);
  auto load_x = [&](int t, f32x4 (&xf)[2][4]) {   // a tile past the end reads zeros through the SRD's range check
    const int p0 = (t * NW + wave) * 32;
    const unsigned xo = ((unsigned)(p0 + li) * (unsigned)p.ldx + 4u * h) * 4u;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = ld4(xr, p0 < p.M ? xo + (unsigned)(kt * 32 + 8 * t8) * 4u : kOOB);
  };
  f32x4 xf[2][4], xn[2][4];
  load_x(blockIdx.x, xn);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = xn[kt][t8];
    load_x(t + gridDim.x, xn);
    if (p0 >= p.M) continue;            // wave-uniform; no barrier below
    float mu, rs;
    ln_apply(xf, lns, h, mu, rs);
    if (p.ln_out) {   // training: the x operand of the Linear's filter gradient
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) store_block_lines(scratch, xf[kt], lnr, (unsigned)p0, (unsigned)CI, (unsigned)(kt * 32), lane, li, h);
    }
#pragma unroll 1
    for (int op = 0; op < NB / 2; ++op) {
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      const float* wb = Wl + ((op * 64) << 5);
#pragma unroll
      for (int j = 0; j < 8; ++j) {       // K-steps (kt, t8) in the tiled kernels' order
        f32x4 a[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const f32x4*>(wb + (((j >> 2) * CO_ + b * 32) << 5) + fo[j & 3]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = MFMA(a[b][s], xf[j >> 2][j & 3][s], acc[b]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + (2 * op + b) * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int s = 0; s < 4; ++s) vq[q][s] = acc[b][4 * q + s] + bq[s];
        }
        store_block_lines(scratch, vq, yr, (unsigned)p0, (unsigned)CO_, (unsigned)((2 * op + b) * 32), lane, li, h);
      }
    }
  }
}

template <int NW, int CO_>
__global__ __launch_bounds__(64 * NW, 2) void lnlin_bwd_kernel(const LnLinParams p) {
  constexpr int NB = CO_ / 32;
  static_assert(NB % 2 == 0, "K-tiles are consumed in pairs");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wts = smem;                      // [NB][64][32]  the transposed filter: rows = input channels, K = output channels
  float* lns = Wts + NB * CI * 32;        // [2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = lns + 2 * CI + wave * 512;
  float* wacc = lns + 2 * CI + NW * 512 + wave * 256 + lane;   // lane-private dgamma / dbeta slots (ln_bwd_epilogue)
  fill_image<NW>(Wts, p.wt, CI, CO_, wave, lane);
  for (int i = tid; i < 2 * CI; i += 64 * NW) lns[i] = i < CI ? p.ln_gamma[i] : p.ln_beta[i - CI];
#pragma unroll
  for (int k = 0; k < 4; ++k) wacc[k * 64] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.x_bytes), dyr = srd(p.dy, p.y_bytes), dxr = srd(p.dx, p.dx_bytes);
  const __amdgpu_buffer_rsrc_t ar = srd(p.dadd ? p.dadd : p.x, p.dadd ? p.add_bytes : p.x_bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* Wtl = Wts + (li << 5);

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
    if (p0 >= p.M) continue;            // wave-uniform; no barrier below
    const unsigned pix = (unsigned)(p0 + li);
    f32x4 dq[2][4], dqn[2][4], xh[2][4], da[2][4];
    auto load_dq = [&](int kp, f32x4 (&dst)[2][4]) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) dst[b][t8] = ld4(dyr, (pix * (unsigned)CO_ + (unsigned)((2 * kp + b) * 32 + 8 * t8 + 4 * h)) * 4u);
    };
    load_dq(0, dq);
    {
      const unsigned xo = (pix * (unsigned)p.ldx + 4u * h) * 4u, ao = (pix * (unsigned)p.ldadd + 4u * h) * 4u;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xh[ib][q] = ld4(xr, xo + (unsigned)(ib * 32 + 8 * q) * 4u);
          if (p.dadd) da[ib][q] = ld4(ar, ao + (unsigned)(ib * 32 + 8 * q) * 4u);
          else da[ib][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x16 dxacc[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) dxacc[ib][r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < NB / 2; ++kp) {   // dx = W^T dy: K-tiles in order, the two input-channel blocks alternate
      if (kp + 1 < NB / 2) load_dq(kp + 1, dqn);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) {
          f32x4 a[2];
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) a[ib] = *reinterpret_cast<const f32x4*>(Wtl + (((2 * kp + b) * CI + ib * 32) << 5) + fo[t8]);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) dxacc[ib] = MFMA(a[ib][s], dq[b][t8][s], dxacc[ib]);
        }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) dq[b][t8] = dqn[b][t8];
    }
    ln_bwd_epilogue<false>(dxacc, xh, da, lns, scratch, wacc, dxr, pix, (unsigned)p.lddx, lane, h);
  }
  __syncthreads();   // (every wave leaves the tile loop after the same number of rounds)
  ln_ws_row<NW>(lns + 2 * CI + NW * 512, p.ln_ws, tid);
}


// ---------------------------------------------------------------------------------------------------------------- plain Linear / 1x1
// y = W x + b (+ res_scale * res) for the 128 -> 128 and 64 -> 64 1x1 layers of the ConvTransBlocks on large maps (conv1_1 / conv1_2 and the
// attention's output projection, forward and data gradient: /root/reference/models/CLC_run.py:205-206, 121; the data gradient is the same
// launch on the transposed filter), in the wave-private structure of the kernels above.  Round 4's first attempt at this (conv_w1x1_kernel,
// see conv_igemm.hip) stored 16-B quads straight from the accumulator layout and was 1.2-2.1x slower than the tiled kernels; with
// whole-line stores (store_block_lines) the same structure moves 3.7-3.9 TB/s (the LayerNorm + Linear launches above).
// Same K order and epilogue expression as the tiled kernels -> the same bits: clc_conv2d may pick either by the row count.
struct LinParams {
  const float* x; const float* w; const float* b; const float* res; float* y;
  int ldx, ldr, ldy;
  float res_scale;
  int M, tiles;
  unsigned x_bytes, res_bytes, y_bytes;
};

template <int NW, int KT, int NB>   // Cin = 32 KT, Cout = 32 NB
__global__ __launch_bounds__(64 * NW, 2) void lin_kernel(const LinParams p) {
  static_assert(NB % 2 == 0, "output blocks are computed in pairs");
  constexpr int CO_ = 32 * NB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [KT][CO_][32]
  float* bs = Ws + KT * CO_ * 32;         // [CO_]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = bs + CO_ + wave * 512;
  fill_image<NW>(Ws, p.w, CO_, 32 * KT, wave, lane);
  for (int i = tid; i < CO_; i += 64 * NW) bs[i] = p.b ? p.b[i] : 0.f;
  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.x_bytes), yr = srd(p.y, p.y_bytes), rr = srd(p.res ? p.res : p.x, p.res ? p.res_bytes : p.x_bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* Wl = Ws + (li << 5);
  auto load_x = [&](int t, f32x4 (&xf)[KT][4]) {   // a tile past the end reads zeros through the SRD's range check
    const int p0 = (t * NW + wave) * 32;
    const unsigned xo = ((unsigned)(p0 + li) * (unsigned)p.ldx + 4u * h) * 4u;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = ld4(xr, p0 < p.M ? xo + (unsigned)(kt * 32 + 8 * t8) * 4u : kOOB);
  };
  f32x4 xf[KT][4], xn[KT][4];
  load_x(blockIdx.x, xn);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
    const unsigned pix = (unsigned)(p0 + li);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) xf[kt][t8] = xn[kt][t8];
    load_x(t + gridDim.x, xn);
    if (p0 >= p.M) continue;            // wave-uniform; no barrier below
#pragma unroll 1
    for (int op = 0; op < NB / 2; ++op) {
      f32x4 rv[2][4];
      if (p.res) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q) rv[b][q] = ld4(rr, (pix * (unsigned)p.ldr + (unsigned)((2 * op + b) * 32 + 8 * q + 4 * h)) * 4u);
      }
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      const float* wb = Wl + ((op * 64) << 5);
#pragma unroll
      for (int j = 0; j < 4 * KT; ++j) {   // K-steps (kt, t8) in the tiled kernels' order
        f32x4 a[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const f32x4*>(wb + (((j >> 2) * CO_ + b * 32) << 5) + fo[j & 3]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = MFMA(a[b][s], xf[j >> 2][j & 3][s], acc[b]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + (2 * op + b) * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int s = 0; s < 4; ++s) vq[q][s] = acc[b][4 * q + s] + bq[s];
          if (p.res) {
            const f32x4 sc = p.res_scale * rv[b][q];
            vq[q] = vq[q] + sc;
          }
        }
        store_block_lines(scratch, vq, yr, (unsigned)p0, (unsigned)p.ldy, (unsigned)((2 * op + b) * 32), lane, li, h);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------- GDN / IGDN backward
// The data gradient of a GDN layer on a large 128-channel map (CompressAI GDN: y = x * (beta + gamma . x^2)^-+1/2; g_a / g_s of
// /root/reference/models/CLC_run.py:296-318, 337-353) as ONE launch.  Before: clc_gdn_bwd_elem (dy, x, v -> dx_direct, dv), then the 1x1 data-gradient
// convolution dx = dx_direct + 2 x (gamma^T dv) — 603 MB per 8 x 128 x 128 layer.  In the wave-private structure the elementwise part is an
// operand prologue and the epilogue reads the SAME registers (operand layout of K-tile ob, group q = accumulator layout of output block ob, quad q):
// x, dx_direct and dv of the tile's pixels fill 192 registers, dv is the B operand, dx_direct never leaves the chip: 335 MB.  dv still goes to HBM
// (whole lines): it is the dy operand of gamma's filter gradient.  Expressions of gdn_bwd_elem_kernel and of epilogue_math4 (norm = MUL2) -> the
// same bits for dv and dx.
struct GdnBwdParams {
  const float* dy; const float* x; const float* v; const float* wt; float* dv; float* dx;
  int M, tiles, inverse;
  unsigned bytes;
};

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void gdn_bwd_kernel(const GdnBwdParams p) {
  constexpr int KT = 4, NB = 4, CC = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [KT][CC][32]  gamma_eff^T: rows = input channels of the forward layer, K = its output channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  float* scratch = Ws + KT * CC * 32 + wave * 512;
  fill_image<NW>(Ws, p.wt, CC, CC, wave, lane);
  const __amdgpu_buffer_rsrc_t xr = srd(p.x, p.bytes), gr = srd(p.dy, p.bytes), vr = srd(p.v, p.bytes), dvr = srd(p.dv, p.bytes), dxr = srd(p.dx, p.bytes);
  const int sw = (li >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + h) ^ sw) << 2;
  const float* Wl = Ws + (li << 5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const int p0 = (t * NW + wave) * 32;
    if (p0 >= p.M) continue;            // wave-uniform; no barrier below
    const unsigned base = ((unsigned)(p0 + li) * (unsigned)CC + 4u * h) * 4u;
    f32x4 xf[KT][4], dd[KT][4], dv[KT][4];   // x, dx_direct (from dy), dv (from v)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8) {
        const unsigned off = base + (unsigned)(kt * 32 + 8 * t8) * 4u;
        xf[kt][t8] = ld4(xr, off);
        dd[kt][t8] = ld4(gr, off);
        dv[kt][t8] = ld4(vr, off);
      }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int t8 = 0; t8 < 4; ++t8)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = dd[kt][t8][e], xv = xf[kt][t8][e], vv = dv[kt][t8][e];
          if (!p.inverse) {
            const float rs = rsqrtf(vv);
            dd[kt][t8][e] = g * rs;
            dv[kt][t8][e] = g * xv * (-0.5f) * rs / vv;
          } else {
            const float sq = sqrtf(vv);
            dd[kt][t8][e] = g * sq;
            dv[kt][t8][e] = g * xv * 0.5f / sq;
          }
        }
      store_block_lines(scratch, dv[kt], dvr, (unsigned)p0, (unsigned)CC, (unsigned)(kt * 32), lane, li, h);
    }
#pragma unroll
    for (int op = 0; op < NB / 2; ++op) {   // (unrolled: xf / dd are indexed by op)
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      const float* wb = Wl + ((op * 64) << 5);
#pragma unroll
      for (int j = 0; j < 4 * KT; ++j) {   // K-steps (kt, t8) in the tiled kernels' order
        f32x4 a[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) a[b] = *reinterpret_cast<const f32x4*>(wb + (((j >> 2) * CC + b * 32) << 5) + fo[j & 3]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b] = MFMA(a[b][s], dv[j >> 2][j & 3][s], acc[b]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            float v = acc[b][4 * q + s] + 0.f;                          // (epilogue_math4: the absent bias is added as 0.f)
            v = 2.f * (xf[op * 2 + b][q][s] * v);                        // norm = MUL2, mul = x
            vq[q][s] = v + 1.0f * dd[op * 2 + b][q][s];                  // + res_scale * dx_direct
          }
        }
        store_block_lines(scratch, vq, dxr, (unsigned)p0, (unsigned)CC, (unsigned)((2 * op + b) * 32), lane, li, h);
      }
    }
  }
}

}  // namespace

static int mlp_fill(const clc_mlp_desc* d, MlpParams& p, bool bwd, const char* who) {
  CLC_CHECK(d && d->x && d->w1 && d->w2, "%s: null pointer", who);
  CLC_CHECK(d->Cin == CI && d->Chid == CH && d->Cout == CO, "%s: built for Linear(64 -> 256) GELU Linear(256 -> 64) (got %d -> %d -> %d)", who, d->Cin, d->Chid, d->Cout);
  CLC_CHECK(d->M > 0 && d->M % 32 == 0 && d->M < (1L << 24), "%s: the pixel count must be a positive multiple of 32 (got %ld)", who, d->M);
  auto ok = [](const void* ptr, int ld, int c) { return ptr == nullptr || (aligned16(ptr) && ld % 4 == 0 && ld >= c); };
  CLC_CHECK(ok(d->x, d->ldx, CI) && aligned16(d->w1) && aligned16(d->w2), "%s: operands must be 16-B aligned with leading dimensions that are multiples of 4", who);
  auto bytes = [](long M, int ld, int c) { return ((size_t)(M - 1) * ld + c) * 4; };
  CLC_CHECK(bytes(d->M, d->ldx, CI) < (1ull << 31) && bytes(d->M, CH, CH) < (1ull << 31), "%s: tensor larger than 2 GiB", who);
  p.x = d->x; p.ldx = d->ldx; p.w1 = d->w1; p.b1 = d->b1; p.w2 = d->w2; p.b2 = d->b2; p.M = (int)d->M;
  p.x_bytes = (unsigned)bytes(d->M, d->ldx, CI);
  p.res = nullptr; p.y = nullptr; p.dy = nullptr; p.w2t = nullptr; p.dx = nullptr; p.dh = nullptr; p.g = nullptr; p.hsave = d->h;
  CLC_CHECK(!d->h || aligned16(d->h), "%s: h unaligned", who);
  p.hid_bytes = (unsigned)bytes(d->M, CH, CH);
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.ln_out = nullptr; p.ln_ws = nullptr;
  CLC_CHECK((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "%s: ln_gamma / ln_beta must both be given or both NULL", who);
  if (d->ln_gamma) {
    CLC_CHECK(aligned16(d->ln_gamma) && aligned16(d->ln_beta), "%s: ln_gamma / ln_beta unaligned", who);
    CLC_CHECK(d->h == nullptr, "%s: the LayerNorm form recomputes the hidden tensor (h must be NULL)", who);
  }
  p.ldr = p.ldy = p.lddy = p.lddx = 0; p.res_bytes = p.y_bytes = p.dy_bytes = p.dx_bytes = 0;
  if (!bwd) {
    CLC_CHECK(d->y && ok(d->y, d->ldy, CO) && ok(d->res, d->ldr, CO), "%s: y / res missing or unaligned", who);
    CLC_CHECK(bytes(d->M, d->ldy, CO) < (1ull << 31) && (!d->res || bytes(d->M, d->ldr, CO) < (1ull << 31)), "%s: tensor larger than 2 GiB", who);
    p.y = d->y; p.ldy = d->ldy; p.y_bytes = (unsigned)bytes(d->M, d->ldy, CO);
    p.res = d->res; p.ldr = d->ldr; p.res_bytes = d->res ? (unsigned)bytes(d->M, d->ldr, CO) : 0;
    CLC_CHECK(!d->ln_gamma || !d->res, "%s: with ln_gamma the residual is x itself (res must be NULL)", who);
    CLC_CHECK(!d->ln_out || (d->ln_gamma && aligned16(d->ln_out)), "%s: ln_out needs ln_gamma and 16-B alignment", who);
    p.ln_out = d->ln_out;
  } else {
    CLC_CHECK(d->dy && d->w2t && d->dx && d->dh && d->g, "%s: dy / w2t / dx / dh / g missing", who);
    CLC_CHECK(ok(d->dy, d->lddy, CO) && ok(d->dx, d->lddx, CI) && aligned16(d->w2t) && aligned16(d->dh) && aligned16(d->g), "%s: gradient operands unaligned", who);
    CLC_CHECK(bytes(d->M, d->lddy, CO) < (1ull << 31) && bytes(d->M, d->lddx, CI) < (1ull << 31), "%s: tensor larger than 2 GiB", who);
    p.dy = d->dy; p.lddy = d->lddy; p.dy_bytes = (unsigned)bytes(d->M, d->lddy, CO);
    p.w2t = d->w2t; p.dx = d->dx; p.lddx = d->lddx; p.dx_bytes = (unsigned)bytes(d->M, d->lddx, CI);
    p.dh = d->dh; p.g = d->g;
    if (d->ln_gamma) {
      CLC_CHECK(d->ln_out && d->ln_ws && aligned16(d->ln_out) && aligned16(d->ln_ws), "%s: ln_out / ln_ws missing or unaligned", who);
      CLC_CHECK(d->ldx == CI, "%s: the LayerNorm form reads a dense x (ln_out has its size)", who);
      p.ln_out = d->ln_out; p.ln_ws = d->ln_ws;
    }
  }
  return 0;
}

// waves per workgroup: 32 pixels per wave; enough workgroups to give every CU one
static int mlp_waves(long M) { return M >= 65536 ? 8 : (M >= 32768 ? 4 : 2); }

static int mlp_grid(long M, int nw) {
  const long tiles = (M + 32 * nw - 1) / (32 * nw);
  return (int)(tiles < 256 ? tiles : 256);   // persistent: one workgroup per CU (129 KB of LDS), filters deposited once
}

template <int NW>
static int mlp_launch(MlpParams& p, bool bwd, hipStream_t st) {
  p.tiles = (p.M + 32 * NW - 1) / (32 * NW);
  const int grid = mlp_grid(p.M, NW);
  const size_t lds_f = (size_t)(2 * CH * 32 + 8 * CO * 32 + CH + CO + 2 * CI + NW * 512) * sizeof(float), lds_b = (size_t)(4 * CH * 32 + CH + NW * 512 + 2 * CI + NW * 256) * sizeof(float);
  const int abl = clc_tuning[CLC_TUNE_ABLATE];
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<NW, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    auto optin = [&](auto kern) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f); };
    optin(&mlp_fwd_kernel<NW, 0, true, true>); optin(&mlp_fwd_kernel<NW, 0, false, true>);
    optin(&mlp_fwd_kernel<NW, 0, true, false>); optin(&mlp_fwd_kernel<NW, 0, false, false>);
    optin(&mlp_fwd_kernel<NW, 0, false, true, true>);
    if (NW == 8) { optin(&mlp_fwd_kernel<8, 1>); optin(&mlp_fwd_kernel<8, 2>); optin(&mlp_fwd_kernel<8, 4>); optin(&mlp_fwd_kernel<8, 5>); }
  }
  auto fwd = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_f, st, p); };
  if (bwd && p.ln_gamma && NW == 8 && abl) {   // timing diagnostics (wrong results)
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_b, st, p);
    };
    if (abl == 1) go(&mlp_bwd_kernel<8, false, true, 1>);
    else if (abl == 4) go(&mlp_bwd_kernel<8, false, true, 4>);
    else go(&mlp_bwd_kernel<8, false, true, 5>);
  } else if (bwd && p.ln_gamma) hipLaunchKernelGGL((mlp_bwd_kernel<NW, false, true>), dim3(grid), dim3(64 * NW), lds_b, st, p);
  else if (!bwd && p.ln_gamma) fwd(&mlp_fwd_kernel<NW, 0, false, true, true>);
  else if (bwd && p.hsave) hipLaunchKernelGGL((mlp_bwd_kernel<NW, true>), dim3(grid), dim3(64 * NW), lds_b, st, p);
  else if (bwd) hipLaunchKernelGGL((mlp_bwd_kernel<NW, false>), dim3(grid), dim3(64 * NW), lds_b, st, p);
  else if (NW == 8 && abl) {   // timing diagnostics (wrong results): see the template argument
    if (abl == 1) fwd(&mlp_fwd_kernel<8, 1>);
    else if (abl == 2) fwd(&mlp_fwd_kernel<8, 2>);
    else if (abl == 4) fwd(&mlp_fwd_kernel<8, 4>);
    else fwd(&mlp_fwd_kernel<8, 5>);
  } else if (clc_tuning[CLC_TUNE_MLP_PK]) {
    if (p.hsave) fwd(&mlp_fwd_kernel<NW, 0, true, true>); else fwd(&mlp_fwd_kernel<NW, 0, false, true>);
  } else {
    if (p.hsave) fwd(&mlp_fwd_kernel<NW, 0, true, false>); else fwd(&mlp_fwd_kernel<NW, 0, false, false>);
  }
  CLC_LAUNCH_CHECK();
  return 0;
}

static int mlp_dispatch(const clc_mlp_desc* d, bool bwd, clc_stream_t stream, const char* who) {
  MlpParams p;
  if (mlp_fill(d, p, bwd, who) < 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  switch (mlp_waves(d->M)) {
    case 8: return mlp_launch<8>(p, bwd, st);
    case 4: return mlp_launch<4>(p, bwd, st);
    default: return mlp_launch<2>(p, bwd, st);
  }
}

extern "C" int clc_mlp_blocks(long M) { return M > 0 ? mlp_grid(M, mlp_waves(M)) : 0; }
extern "C" int clc_mlp_fwd(const clc_mlp_desc* d, clc_stream_t stream) { return mlp_dispatch(d, false, stream, "clc_mlp_fwd"); }
extern "C" int clc_mlp_bwd(const clc_mlp_desc* d, clc_stream_t stream) { return mlp_dispatch(d, true, stream, "clc_mlp_bwd"); }

// ---- LayerNorm + Linear(64 -> 192)
static int lnlin_dispatch(const clc_lnlin_desc* d, bool bwd, clc_stream_t stream, const char* who) {
  CLC_CHECK(d && d->x && d->ln_gamma && d->ln_beta, "%s: null pointer", who);
  CLC_CHECK(d->Cin == CI && d->Cout == 192, "%s: built for LayerNorm(64) + Linear(64 -> 192) (got %d -> %d)", who, d->Cin, d->Cout);
  CLC_CHECK(d->M > 0 && d->M % 32 == 0 && d->M < (1L << 24), "%s: the pixel count must be a positive multiple of 32 (got %ld)", who, d->M);
  auto ok = [](const void* ptr, int ld, int c) { return ptr == nullptr || (aligned16(ptr) && ld % 4 == 0 && ld >= c); };
  auto bytes = [](long M, int ld, int c) { return ((size_t)(M - 1) * ld + c) * 4; };
  CLC_CHECK(ok(d->x, d->ldx, CI) && aligned16(d->ln_gamma) && aligned16(d->ln_beta), "%s: operands must be 16-B aligned with leading dimensions that are multiples of 4", who);
  CLC_CHECK(bytes(d->M, d->ldx, CI) < (1ull << 31) && bytes(d->M, 192, 192) < (1ull << 31), "%s: tensor larger than 2 GiB", who);
  LnLinParams p;
  p.x = d->x; p.ldx = d->ldx; p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.M = (int)d->M;
  p.x_bytes = (unsigned)bytes(d->M, d->ldx, CI); p.y_bytes = (unsigned)bytes(d->M, 192, 192);
  p.w = nullptr; p.b = nullptr; p.y = nullptr; p.ln_out = nullptr; p.dy = nullptr; p.wt = nullptr; p.dx = nullptr; p.dadd = nullptr; p.ln_ws = nullptr;
  p.lddx = p.ldadd = 0; p.dx_bytes = p.add_bytes = 0;
  if (!bwd) {
    CLC_CHECK(d->w && d->y && aligned16(d->w) && aligned16(d->y) && (!d->ln_out || aligned16(d->ln_out)), "%s: w / y missing or unaligned", who);
    p.w = d->w; p.b = d->b; p.y = d->y; p.ln_out = d->ln_out;
  } else {
    CLC_CHECK(d->dy && d->wt && d->dx && d->ln_ws && aligned16(d->dy) && aligned16(d->wt) && aligned16(d->ln_ws), "%s: dy / wt / dx / ln_ws missing or unaligned", who);
    CLC_CHECK(ok(d->dx, d->lddx, CI) && ok(d->dadd, d->ldadd, CI), "%s: dx / dadd unaligned", who);
    CLC_CHECK(bytes(d->M, d->lddx, CI) < (1ull << 31) && (!d->dadd || bytes(d->M, d->ldadd, CI) < (1ull << 31)), "%s: tensor larger than 2 GiB", who);
    p.dy = d->dy; p.wt = d->wt; p.dx = d->dx; p.lddx = d->lddx; p.dx_bytes = (unsigned)bytes(d->M, d->lddx, CI);
    p.dadd = d->dadd; p.ldadd = d->ldadd; p.add_bytes = d->dadd ? (unsigned)bytes(d->M, d->ldadd, CI) : 0;
    p.ln_ws = d->ln_ws;
  }
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto nw_tag) {
    constexpr int NW = decltype(nw_tag)::value;
    p.tiles = (p.M + 32 * NW - 1) / (32 * NW);
    const int grid = mlp_grid(p.M, NW);
    const size_t lds_f = (size_t)(2 * 192 * 32 + 192 + 2 * CI + NW * 512) * sizeof(float), lds_b = (size_t)(6 * CI * 32 + 2 * CI + NW * 512 + NW * 256) * sizeof(float);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlin_fwd_kernel<NW, 192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lnlin_bwd_kernel<NW, 192>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    }
    if (bwd) hipLaunchKernelGGL((lnlin_bwd_kernel<NW, 192>), dim3(grid), dim3(64 * NW), lds_b, st, p);
    else hipLaunchKernelGGL((lnlin_fwd_kernel<NW, 192>), dim3(grid), dim3(64 * NW), lds_f, st, p);
  };
  switch (mlp_waves(d->M)) {
    case 8: go(std::integral_constant<int, 8>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    default: go(std::integral_constant<int, 2>{}); break;
  }
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_lnlin_fwd(const clc_lnlin_desc* d, clc_stream_t stream) { return lnlin_dispatch(d, false, stream, "clc_lnlin_fwd"); }
extern "C" int clc_lnlin_bwd(const clc_lnlin_desc* d, clc_stream_t stream) { return lnlin_dispatch(d, true, stream, "clc_lnlin_bwd"); }

// ---- plain 1x1 layers on large maps: called by clc_conv2d (conv_igemm.hip) for the shapes it checked; returns 0 when the shape is not built
template <int NW, int KT, int NB>
static void lin_go(LinParams& p, hipStream_t st) {
  p.tiles = (p.M + 32 * NW - 1) / (32 * NW);
  const int grid = mlp_grid(p.M, NW);
  const size_t lds = (size_t)(KT * NB * 32 * 32 + NB * 32 + NW * 512) * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_kernel<NW, KT, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((lin_kernel<NW, KT, NB>), dim3(grid), dim3(64 * NW), lds, st, p);
}
int clc_lin_launch(const float* x, int ldx, const float* w, const float* bias, const float* res, int ldr, float res_scale, float* y, int ldy, long M, int Cin,
                   int Cout, hipStream_t st) {
  if (!((Cin == 128 && Cout == 128) || (Cin == 64 && Cout == 64)) || M % 32 || M < 32768 || M >= (1L << 24)) return 0;
  auto bytes = [](long m, int ld, int c) { return ((size_t)(m - 1) * ld + c) * 4; };
  if (bytes(M, ldx, Cin) >= (1ull << 31) || bytes(M, ldy, Cout) >= (1ull << 31) || (res && bytes(M, ldr, Cout) >= (1ull << 31))) return 0;
  LinParams p;
  p.x = x; p.w = w; p.b = bias; p.res = res; p.y = y; p.ldx = ldx; p.ldr = ldr; p.ldy = ldy; p.res_scale = res_scale; p.M = (int)M;
  p.x_bytes = (unsigned)bytes(M, ldx, Cin); p.y_bytes = (unsigned)bytes(M, ldy, Cout); p.res_bytes = res ? (unsigned)bytes(M, ldr, Cout) : 0;
  const int nw = mlp_waves(M);
  if (Cin == 128) {
    if (nw == 8) lin_go<8, 4, 4>(p, st); else lin_go<4, 4, 4>(p, st);
  } else {
    if (nw == 8) lin_go<8, 2, 2>(p, st); else lin_go<4, 2, 2>(p, st);
  }
  return (11 << 20) | (nw << 16) | (Cin >> 5 << 8) | (Cout >> 5);   // family 11 = lin_kernel<NW, KT, NB>
}

// ---- GDN / IGDN data gradient, 128 channels, large maps
extern "C" int clc_gdn_bwd_fused(const float* dy, const float* x, const float* v, const float* gamma_eff_t, float* dv, float* dx, long M, int C, int inverse,
                                 clc_stream_t stream) {
  CLC_CHECK(dy && x && v && gamma_eff_t && dv && dx, "clc_gdn_bwd_fused: null pointer");
  CLC_CHECK(C == 128 && M >= 32768 && M % 32 == 0 && (size_t)M * C * 4 < (1ull << 31), "clc_gdn_bwd_fused: built for 128 channels on >= 32 768 rows (a multiple of 32); got C=%d M=%ld", C, M);
  CLC_CHECK(aligned16(dy) && aligned16(x) && aligned16(v) && aligned16(gamma_eff_t) && aligned16(dv) && aligned16(dx), "clc_gdn_bwd_fused: operands must be 16-B aligned");
  GdnBwdParams p;
  p.dy = dy; p.x = x; p.v = v; p.wt = gamma_eff_t; p.dv = dv; p.dx = dx; p.M = (int)M; p.inverse = inverse; p.bytes = (unsigned)((size_t)M * C * 4);
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto nw_tag) {
    constexpr int NW = decltype(nw_tag)::value;
    p.tiles = (p.M + 32 * NW - 1) / (32 * NW);
    const size_t lds = (size_t)(4 * 128 * 32 + NW * 512) * sizeof(float);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_bwd_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gdn_bwd_kernel<NW>), dim3(mlp_grid(p.M, NW)), dim3(64 * NW), lds, st, p);
  };
  if (mlp_waves(M) == 8) go(std::integral_constant<int, 8>{}); else go(std::integral_constant<int, 4>{});
  CLC_LAUNCH_CHECK();
  return 0;
}
