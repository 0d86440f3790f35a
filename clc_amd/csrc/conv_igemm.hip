// conv_igemm.hip — NHWC fp32 convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces the cuDNN forward / backward-data kernels behind every nn.Conv2d / nn.Linear of the
// CLC hot path (/root/reference/models/CLC_run.py:120,123,185-187,207-208,273-277,335-354,
// 412-481; CompressAI layers).
//
// GEMM view:  M = output pixels, N = output channels, K = (tap, input channel).
//   A[m][k] is gathered from the NHWC input (zero outside the image), B[n][k] is the filter
//   row [Cout][tap][Cin] — both K-contiguous, so one LDS image serves both: [rows][32+4] floats,
//   16-B loads -> ds_write_b128, fragments by ds_read_b128 (the +4 pad = 9 sixteen-byte slots
//   per row makes every 16-lane read group hit 16 distinct slots: conflict-free).
// Loads are `buffer_load_dwordx4` through an SRD with hardware range checking: halo pixels,
// rows past M and channels past Cin/Cout get an out-of-range offset and come back as zeros —
// no branch, no exec-mask juggling, no wait in front of the MFMAs (branches around loads make
// hipcc wait vmcnt(0) at every join: measured 3x slower).
// The f32 MFMA takes ONE float per lane per operand (lane l: A[i=l&31][k=l>>5]); a lane reads 4
// consecutive k with one ds_read_b128 and feeds 4 MFMAs from it, i.e. MFMA step (t,s) contracts
// the physical k pair {8t+s, 8t+4+s}.  A and B use the same pairing, so the sum is exact.
// Summation order per output element is a function of (ks, Cin) and of the kernel FAMILY only;
// the family is chosen from the per-image map size and channel counts — never from the batch
// size — so an image's result does not depend on what else is in the batch.
//
// Two families:
//   conv_igemm_kernel<BM,BN,WM,WN>   block-cooperative tiles, double-buffered LDS, register
//                                    prefetch of K-tile k+1 under the MFMAs of tile k, 2 WG/CU
//                                    (8 waves per 128-row tile: 4 waves per SIMD).
//   conv_igemm_splitk_kernel<BN>     small maps (<= 16x16 per image): 8 waves per 32xBN tile,
//                                    each wave takes every 8th K-tile and loads its MFMA
//                                    fragments straight from global memory (no LDS, no barrier
//                                    in the loop, all its tiles in flight at once), fixed-order
//                                    tree combine through LDS.
#include "common.h"
#ifndef CLC_PF32
#define CLC_PF32 3
#endif

namespace {
#include "conv_common.h"
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool TR>
__global__ __launch_bounds__(64 * WM * WN, (64 * WM * WN) >= 512 ? 4 : ((64 * WM * WN) >= 256 ? 2 : 4))
void conv_igemm_kernel(const ConvParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_P = BM * 8 / NT, B_P = BN * 8 / NT;
  static_assert(TM >= 1 && TN >= 1 && A_P * NT == BM * 8 && B_P * NT == BN * 8, "tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LD]
  float* Bs = smem + 2 * BM * LD;      // [2][BN][LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;   // parity class (transposed stride 2 only)
  const bool half = p.transposed && p.stride == 2;
  const int DH = half ? p.OH / 2 : p.OH, DW = half ? p.OW / 2 : p.OW;
  const TapGrid tg = make_taps(p, ph, pw);

  const int fset = p.group_rows ? min(m0 / p.group_rows, 3) : 0;   // block-uniform: tiles never straddle two filter sets
  const float* wsel = fset == 0 ? p.w : (fset == 1 ? p.w2 : (fset == 2 ? p.w3 : p.w4));
  const float* bsel = fset == 0 ? p.bias : (fset == 1 ? p.bias2 : (fset == 2 ? p.bias3 : p.bias4));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsel), 0, p.w_bytes, 0x00020000);

  RowState rows[A_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) rows[i] = make_row<TR>(p, m0 + ((tid + i * NT) >> 3), DH, DW, ph, pw);
  unsigned b_row_off[B_P];
  bool b_ok[B_P];
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    const int co = n0 + ((tid + i * NT) >> 3);
    b_ok[i] = co < p.Cout;
    b_row_off[i] = (unsigned)co * (unsigned)p.ldw;
  }
  const int c4 = (tid & 7) * 4;
  const bool sq = p.in_op == CLC_IN_SQUARE;

  const bool fuse_act = p.xs != nullptr;
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fuse_act ? p.xs : p.x), 0, fuse_act ? p.xs_bytes : p.x_bytes, 0x00020000);
  f32x4 a_reg[A_P], b_reg[B_P], s_reg[A_P];
  auto load_tile = [&](int kh, int kw, int kc) {
    const int c = kc * BK + c4;
    const bool c_ok = c < p.Cin;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const unsigned pix = a_pixel<TR>(p, rows[i], kh, kw);
      a_reg[i] = buf_load4(xr, pix_off(pix, p.ldx, c, c_ok));
      if (fuse_act) s_reg[i] = buf_load4(sr, pix_off(pix, p.ldxs, c, c_ok));    // block-uniform branch
    }
    const unsigned tap_off = (unsigned)((kh * p.ks + kw) * p.Cin + c);
#pragma unroll
    for (int i = 0; i < B_P; ++i) b_reg[i] = buf_load4(wr, (b_ok[i] && c_ok) ? (b_row_off[i] + tap_off) * 4u : kOOB);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      f32x4 v = sq ? a_reg[i] * a_reg[i] : a_reg[i];
      if (fuse_act) v = v * act_deriv4(s_reg[i], p.xs_act, p.xs_pre);
      *reinterpret_cast<f32x4*>(As + (buf * BM + ((tid + i * NT) >> 3)) * LD + c4) = v;
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) *reinterpret_cast<f32x4*>(Bs + (buf * BN + ((tid + i * NT) >> 3)) * LD + c4) = b_reg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K loop over (tap row j, tap col i, channel tile kc)
  const int total = tg.nkh * tg.nkw * p.kc_tiles;
  int tj = 0, ti = 0, kc = 0;
  auto advance = [&]() {
    if (++kc == p.kc_tiles) { kc = 0; if (++ti == tg.nkw) { ti = 0; ++tj; } }
  };
  load_tile(tg.kh0, tg.kw0, 0);
  store_tile(0);
  __syncthreads();
  const int frag_col = 4 * (lane >> 5);
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    advance();
    if (it + 1 < total) load_tile(tg.kh0 + tg.step * tj, tg.kw0 + tg.step * ti, kc);   // block-uniform branch
    const float* Ab = As + (buf * BM + wm * (BM / WM) + (lane & 31)) * LD + frag_col;
    const float* Bb = Bs + (buf * BN + wn * (BN / WN) + (lane & 31)) * LD + frag_col;
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LD + t8 * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LD + t8 * 8);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][ss], bf[j][ss], acc[i][j], 0, 0, 0);
    }
    if (it + 1 < total) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: stage the C tile in LDS (the operand tiles are dead now), then stream it out with a compact loop —
  // consecutive threads = consecutive channels of one pixel -> fully coalesced stores, no 64x unrolled flag checks.
  // C/D map of the MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel).
  constexpr int LDC = BN + 4;
  float* Cs = smem;   // [BM][LDC]  (fits: BM*LDC <= 2*(BM+BN)*LD for every tile variant)
  {
    const int col = lane & 31, rhalf = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf) * LDC + wn * (BN / WN) + j * 32 + col] = acc[i][j][r];
  }
  __syncthreads();
  if (p.vec_epi) {   // block-uniform
    for (int e = tid; e < BM * BN / 4; e += NT) {
      const int row = e / (BN / 4), cc = (e - row * (BN / 4)) * 4;
      const int m = m0 + row, co = n0 + cc;
      if (m < p.M && co < p.Cout) epilogue_store4(p, bsel, *reinterpret_cast<const f32x4*>(Cs + row * LDC + cc), m, co, DH, DW, ph, pw);
    }
    return;
  }
  for (int e = tid; e < BM * BN; e += NT) {
    const int row = e / BN, cc = e - row * BN;
    const int m = m0 + row, co = n0 + cc;
    if (m < p.M && co < p.Cout) epilogue_store(p, Cs[row * LDC + cc], bsel ? bsel[co] : 0.f, m, co, DH, DW, ph, pw);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of conv_igemm_kernel: `buffer_load_dwordx4 ... lds` deposits the operand tiles straight in LDS — no
// VGPR staging, no ds_write pass (measured: the staging writes cost ~15 % of the register-staged kernel, the loads ~8 %).
// One DMA wave-instruction writes 1 KiB linearly (lane l -> 16 B at base + 16 l) = 8 tile rows of 32 floats, so rows
// cannot be padded; instead the 16-B chunk c of tile row r lives in slot c ^ ((r >> 1) & 7): every lane FETCHES the chunk
// that belongs in its slot (the global source address is per lane) and the fragment reads apply the same XOR — the 16
// rows of each ds_read_b128 lane group then fall in 16 distinct 16-B bank groups.  Halo pixels / rows past M / channels
// past Cin go through the SRD's range check and deposit zeros.  K order and MFMA sequence are those of
// conv_igemm_kernel, so the results are the same bits.  No input prologue (square / fused activation derivative).
template <int BM, int BN, int WM, int WN, bool TR>
__global__ __launch_bounds__(64 * WM * WN, (64 * WM * WN) >= 512 ? 4 : ((64 * WM * WN) >= 256 ? 2 : 4))
void conv_igemm_dma_kernel(const ConvParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_P = BM * 8 / NT, B_P = BN * 8 / NT;
  static_assert(TM >= 1 && TN >= 1 && A_P * NT == BM * 8 && B_P * NT == BN * 8, "tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][BK], slot-swizzled
  float* Bs = smem + 2 * BM * BK;      // [2][BN][BK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;
  const bool half = p.transposed && p.stride == 2;
  const int DH = half ? p.OH / 2 : p.OH, DW = half ? p.OW / 2 : p.OW;
  const TapGrid tg = make_taps(p, ph, pw);

  const int fset = p.group_rows ? min(m0 / p.group_rows, 3) : 0;   // block-uniform: tiles never straddle two filter sets
  const float* wsel = fset == 0 ? p.w : (fset == 1 ? p.w2 : (fset == 2 ? p.w3 : p.w4));
  const float* bsel = fset == 0 ? p.bias : (fset == 1 ? p.bias2 : (fset == 2 ? p.bias3 : p.bias4));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsel), 0, p.w_bytes, 0x00020000);

  RowState rows[A_P];
  int a_c4[A_P], b_c4[B_P];          // channel offset (floats) of the chunk this lane fetches for its slot
  unsigned b_row_off[B_P];
  bool b_ok[B_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int r = (tid + i * NT) >> 3;
    rows[i] = make_row<TR>(p, m0 + r, DH, DW, ph, pw);
    a_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
  }
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    const int r = (tid + i * NT) >> 3;
    const int co = n0 + r;
    b_ok[i] = co < p.Cout;
    b_row_off[i] = (unsigned)co * (unsigned)p.ldw;
    b_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
  }
  const int wave_row = wave * 8;     // first tile row of this wave's DMA pieces (8 rows per wave-instruction)

  auto dma_tile = [&](int buf, int kh, int kw, int kc) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int c = kc * BK + a_c4[i];
      const unsigned pix = a_pixel<TR>(p, rows[i], kh, kw);
      float* dst = As + (buf * BM + wave_row + i * (NT / 8)) * BK;
      dma16(xr, dst, pix_off(pix, p.ldx, c, c < p.Cin));
    }
    const unsigned tap_off = (unsigned)((kh * p.ks + kw) * p.Cin + kc * BK);
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const bool ok = b_ok[i] && (kc * BK + b_c4[i] < p.Cin);
      float* dst = Bs + (buf * BN + wave_row + i * (NT / 8)) * BK;
      dma16(wr, dst, ok ? (b_row_off[i] + tap_off + (unsigned)b_c4[i]) * 4u : kOOB);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int total = tg.nkh * tg.nkw * p.kc_tiles;
  int tj = 0, ti = 0, kc = 0;
  auto advance = [&]() {
    if (++kc == p.kc_tiles) { kc = 0; if (++ti == tg.nkw) { ti = 0; ++tj; } }
  };
  dma_tile(0, tg.kh0, tg.kw0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment (row lr of a 32-row group, chunk 2*t8 + h) sits in slot (2*t8 + h) ^ ((lr >> 1) & 7)
  const int lr = lane & 31, hh = lane >> 5, sw = (lr >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + hh) ^ sw) * 4;
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    advance();
    if (it + 1 < total) dma_tile(buf ^ 1, tg.kh0 + tg.step * tj, tg.kw0 + tg.step * ti, kc);   // block-uniform branch
    const float* Ab = As + (buf * BM + wm * (BM / WM) + lr) * BK;
    const float* Bb = Bs + (buf * BN + wn * (BN / WN) + lr) * BK;
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * BK + fo[t8]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * BK + fo[t8]);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][ss], bf[j][ss], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this iteration's DMA pieces have landed
    __syncthreads();
  }

  constexpr int LDC = BN + 4;
  float* Cs = smem;   // [BM][LDC]
  {
    const int col = lane & 31, rhalf = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf) * LDC + wn * (BN / WN) + j * 32 + col] = acc[i][j][r];
  }
  __syncthreads();
  if (p.vec_epi) {   // block-uniform
    for (int e = tid; e < BM * BN / 4; e += NT) {
      const int row = e / (BN / 4), cc = (e - row * (BN / 4)) * 4;
      const int m = m0 + row, co = n0 + cc;
      if (m < p.M && co < p.Cout) epilogue_store4(p, bsel, *reinterpret_cast<const f32x4*>(Cs + row * LDC + cc), m, co, DH, DW, ph, pw);
    }
    return;
  }
  for (int e = tid; e < BM * BN; e += NT) {
    const int row = e / BN, cc = e - row * BN;
    const int m = m0 + row, co = n0 + cc;
    if (m < p.M && co < p.Cout) epilogue_store(p, Cs[row * LDC + cc], bsel ? bsel[co] : 0.f, m, co, DH, DW, ph, pw);
  }
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_dma_kernel with the K loop re-timed (same tiles, same K order, same MFMA sequence -> same bits):
//   * the gather address of a row is (origin + tap delta): origin byte offset and a 9-bit tap-validity mask are computed
//     once per row, the tap delta is a scalar -> ~4 VALU instructions per DMA piece instead of ~12 with two v_mul_lo_u32;
//   * the next tile's DMA pieces are issued BETWEEN the MFMA groups of the current tile (their address VALU executes in
//     the shadow of the MFMA that was just issued) instead of in a block of their own at the top of the iteration, and
//     without a branch around them (the last iteration deposits zeros in the idle buffer);
//   * group 0's fragments of the next tile are read right after the barrier, two fragment register sets are offered to the
//     scheduler (hipcc still sinks most reads to just before their first MFMA; pinning the order with sched_group_barrier
//     did not change that).
// KS = 1: the 1x1 / linear instantiation (one tap: no tap grid, no validity mask; its own symbol, so profiles tell the HBM-bound
// 1x1 layers from the MFMA-bound 3x3 ones); KS = 3: everything else.
// BF = true: the opt-in REDUCED-PRECISION mode (the reference's --use-mixed-precision branch, train_CLC.py:143-174):
// the same f32 tiles in LDS, but two fragment groups (2 x 4 consecutive k per lane) are rounded to bf16 and contracted by ONE
// v_mfma_f32_32x32x16_bf16 (f32 accumulate) instead of eight f32 MFMAs — 1/16 of the matrix-pipe time; A and B use the same
// lane -> k assignment, so the contraction is over the same 16 k.  Other bits than the f32 kernels by design.
template <int BM, int BN, int WM, int WN, bool TR, int KS, int OP = 0, bool BF = false>
__global__ __launch_bounds__(64 * WM * WN, (64 * WM * WN) >= 512 ? 4 : ((64 * WM * WN) >= 256 ? 2 : 4))
void conv_igemm_dma2_kernel(const ConvParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_P = BM * 8 / NT, B_P = BN * 8 / NT;
  static_assert(TM >= 1 && TN >= 1 && A_P * NT == BM * 8 && B_P * NT == BN * 8, "tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][BK], slot-swizzled
  float* Bs = smem + 2 * BM * BK;      // [2][BN][BK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroup -> tile.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2; in the
  // plain (x = pixel tile, y = channel tile) order the gridDim.y workgroups that read the same pixel tile are gridDim.x ids
  // apart: different times, different L2s, the operand fetched from HBM / Infinity Cache once per channel tile.  With xcd_map
  // (host: gridDim.x % 8 == 0) id L goes to XCD c = L % 8 as its j = L / 8-th workgroup and takes channel tile j % gridDim.y of
  // pixel tile (j / gridDim.y) * 8 + c: the channel tiles of one pixel tile run back to back on one XCD
  // (r2 PMC, 1x1 layers of the step: 8.9 -> 6.5 GB fetched).
  int bxm = blockIdx.x, bym = blockIdx.y;
  if (p.xcd_map) {
    const int L = blockIdx.x + gridDim.x * blockIdx.y, c = L & 7, j = L >> 3;
    bym = j % (int)gridDim.y;
    bxm = (j / (int)gridDim.y) * 8 + c;
  }
  const int m0 = bxm * BM, n0 = bym * BN;
  const int ksp = p.ksplit > 1 ? p.ksplit : 1;                // block-uniform
  const int cls = blockIdx.z / ksp, sk = blockIdx.z - cls * ksp, ph = cls >> 1, pw = cls & 1;
  const bool half = p.transposed && p.stride == 2;
  const int DH = half ? p.OH / 2 : p.OH, DW = half ? p.OW / 2 : p.OW;
  const TapGrid tg = KS == 1 ? TapGrid{0, 0, 1, 1, 1} : make_taps(p, ph, pw);   // (KS = 1 is dispatched for stride 1 only)

  const int fset = p.group_rows ? min(m0 / p.group_rows, 3) : 0;   // block-uniform: tiles never straddle two filter sets
  const float* wsel = fset == 0 ? p.w : (fset == 1 ? p.w2 : (fset == 2 ? p.w3 : p.w4));
  const float* bsel = fset == 0 ? p.bias : (fset == 1 ? p.bias2 : (fset == 2 ? p.bias3 : p.bias4));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsel), 0, p.w_bytes, 0x00020000);

  // Row r of the tile at tap-grid index (j, i) reads source pixel origin + sgn * (j * W + i): forward taps walk up/right from
  // (y0, x0); a data gradient walks down/left from (y0 - kh0, x0 - kw0) (>> 1 for stride 2: the tap grid has the class's parity).
  constexpr int sgn = TR ? -1 : 1;
  unsigned a_base[A_P], a_mask[A_P], b_base[B_P];
  int a_c4[A_P], b_c4[B_P];
  bool b_ok[B_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int r = (tid + i * NT) >> 3;
    const RowState rs = make_row<TR>(p, m0 + r, DH, DW, ph, pw);
    a_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
    int oy, ox;
    if (TR) { const int sh = p.stride - 1; oy = (rs.y0 - tg.kh0) >> sh; ox = (rs.x0 - tg.kw0) >> sh; }
    else { oy = rs.y0 + tg.kh0; ox = rs.x0 + tg.kw0; }
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int iy = oy + sgn * j, ix = ox + sgn * q;
        const bool ok = rs.ok && j < tg.nkh && q < tg.nkw && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        mask |= ok ? (1u << (j * 3 + q)) : 0u;
      }
    a_mask[i] = mask;
    a_base[i] = ((unsigned)(rs.base + oy * p.W + ox) * (unsigned)p.ldx + (unsigned)a_c4[i]) * 4u;   // (wraps harmlessly when invalid: never used then)
  }
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    const int r = (tid + i * NT) >> 3;
    const int co = n0 + r;
    b_ok[i] = co < p.Cout;
    b_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
    b_base[i] = ((unsigned)co * (unsigned)p.ldw + (unsigned)b_c4[i]) * 4u;
  }
  const int wave_row = wave * 8;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this workgroup's K-steps: all of them, or split sk's share [it_begin, it_begin + total) of a split launch
  const int total_all = tg.nkh * tg.nkw * p.kc_tiles;
  const int it_begin = (int)((long)total_all * sk / ksp), total = (int)((long)total_all * (sk + 1) / ksp) - it_begin;
  int tj, ti, kc;
  {
    const int tap = it_begin / p.kc_tiles;
    kc = it_begin - tap * p.kc_tiles;
    tj = tap / tg.nkw;
    ti = tap - tj * tg.nkw;
  }
  auto advance = [&]() {   // branch-free: the loop body stays ONE scheduling region
    const int kc1 = kc + 1;
    const bool w1 = kc1 == p.kc_tiles;
    kc = w1 ? 0 : kc1;
    const int ti1 = ti + (w1 ? 1 : 0);
    const bool w2 = ti1 == tg.nkw;
    ti = w2 ? 0 : ti1;
    tj += w2 ? 1 : 0;
  };
  // scalar state of the tile being fetched
  unsigned s_bit, s_adelta, s_bdelta; int s_cleft; bool s_en;
  auto set_fetch = [&](bool en) {
    s_en = en;
    s_bit = 1u << (tj * 3 + ti);
    s_cleft = p.Cin - kc * BK;
    s_adelta = (unsigned)(sgn * (tj * p.W + ti) * p.ldx + kc * BK) * 4u;
    const int kh = tg.kh0 + tg.step * tj, kw = tg.kw0 + tg.step * ti;
    s_bdelta = (unsigned)((kh * p.ks + kw) * p.Cin + kc * BK) * 4u;
  };
  auto dma_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const bool ok = s_en && (a_mask[i] & s_bit) != 0u && a_c4[i] < s_cleft;
      dma16(xr, As + (buf * BM + wave_row + i * (NT / 8)) * BK, ok ? a_base[i] + s_adelta : kOOB);
    }
  };
  auto dma_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const bool ok = s_en && b_ok[i] && b_c4[i] < s_cleft;
      dma16(wr, Bs + (buf * BN + wave_row + i * (NT / 8)) * BK, ok ? b_base[i] + s_bdelta : kOOB);
    }
  };
  set_fetch(true);
  dma_a(0);
  dma_b(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int lr = lane & 31, hh = lane >> 5, sw = (lr >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + hh) ^ sw) * 4;
  f32x4 af[2][TM], bf[2][TN];
  // OP = 1 (GDN's norm convolution, a 1x1 layer: its own instantiation): the operand is squared on its way OUT of LDS — DMA cannot.
  // (The same trick for data gradients with a fused activation derivative on the operand — the saved tile fetched by DMA beside
  // it, operand * act'(saved) at fragment read — was built and measured: -0.3 %, the second operand tile costs a resident workgroup.)
  auto read_frag = [&](int buf, int t8, f32x4 (&a)[TM], f32x4 (&b)[TN]) {
    const float* Ab = As + (buf * BM + wm * (BM / WM) + lr) * BK + fo[t8];
    const float* Bb = Bs + (buf * BN + wn * (BN / WN) + lr) * BK + fo[t8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * BK);
      if (OP == 1) a[i] = a[i] * a[i];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * BK);
  };
  auto mfma_group = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {
#pragma unroll
    for (int ss = 0; ss < 4; ++ss)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][ss], b[j][ss], acc[i][j], 0, 0, 0);
  };
  auto mfma_pair_bf16 = [&](const f32x4 (&a0)[TM], const f32x4 (&b0)[TN], const f32x4 (&a1)[TM], const f32x4 (&b1)[TN]) {
    bf16x8 pa[TM], pb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) pa[i] = pack_bf16(a0[i], a1[i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) pb[j] = pack_bf16(b0[j], b1[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i], pb[j], acc[i][j], 0, 0, 0);
  };
  read_frag(0, 0, af[0], bf[0]);
  if constexpr (BF) {
    for (int it = 0; it < total; ++it) {
      const int buf = it & 1;
      advance();
      set_fetch(it + 1 < total);
      dma_a(buf ^ 1); dma_b(buf ^ 1);
      read_frag(buf, 1, af[1], bf[1]);
      mfma_pair_bf16(af[0], bf[0], af[1], bf[1]);
      read_frag(buf, 2, af[0], bf[0]);
      read_frag(buf, 3, af[1], bf[1]);
      mfma_pair_bf16(af[0], bf[0], af[1], bf[1]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      read_frag(buf ^ 1, 0, af[0], bf[0]);
    }
  } else
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    advance();
    set_fetch(it + 1 < total);
    // The next tile's DMA pieces are issued at the TOP of the iteration: they have the whole iteration to land — what counts inside the
    // training step, where operands come from HBM / Infinity Cache (between the MFMA groups measured -1.7 % in the step, round 2).
    // No block-uniform branch in this loop (round 4 removed the ablation / placement switches: every one of them split the K step
    // into separately scheduled pieces).
    __builtin_amdgcn_s_setprio(2);   // the wave that has just passed the barrier issues its share of the next tile FIRST, ahead of the other waves' MFMAs
    dma_a(buf ^ 1); dma_b(buf ^ 1);   // (round 4, three interleaved builds on one box: 27.08 -> 26.92 ms per step; priority 3: 26.99; the window extended
    __builtin_amdgcn_s_setprio(0);   //  over the first fragment read: 27.05; priority 1 around the MFMA groups instead: 27.24 vs 27.09; the same in the
    read_frag(buf, 1, af[1], bf[1]);  //  filter-gradient kernels' K loops: no change)
    mfma_group(af[0], bf[0]);
    read_frag(buf, 2, af[0], bf[0]);
    mfma_group(af[1], bf[1]);
    read_frag(buf, 3, af[1], bf[1]);
    mfma_group(af[0], bf[0]);
    mfma_group(af[1], bf[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's DMA pieces have landed
    __syncthreads();
    read_frag(buf ^ 1, 0, af[0], bf[0]);                // (after the last tile: a harmless read of the zero-filled buffer)
  }

  if (ksp > 1) {   // raw partial tile -> scratch [class][split][M][Cout] (whole tiles: host-checked); conv_splitk_finish_kernel does the rest
    float* dst = p.partial + ((size_t)(cls * ksp + sk) * p.M) * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * (BN / WN) + j * 32 + lr;
        const int row0 = m0 + wm * (BM / WM) + i * 32 + 4 * hh;
        if (co >= p.Cout) continue;                       // (Cout a multiple of 32, not of the tile: the last tile's upper column blocks)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * p.Cout + co] = acc[i][j][r];
      }
    return;
  }
  if (p.reg_epi) {   // block-uniform (host: reg_epi_ok): every wave stores its own 32 x 32 blocks, no C tile in LDS, no barrier
    const RegEpi re = make_reg_epi(p);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * (BN / WN) + j * 32 + lr;
        epilogue_regs(p, re, acc[i][j], bsel ? bsel[co] : 0.f, (unsigned)(m0 + wm * (BM / WM) + i * 32 + 4 * hh), co);
      }
    return;
  }
  constexpr int LDC = BN + 4;
  float* Cs = smem;   // [BM][LDC]
  __syncthreads();    // every wave's trailing fragment read is done before the tile is overwritten
  {
    const int col = lane & 31, rhalf = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf) * LDC + wn * (BN / WN) + j * 32 + col] = acc[i][j][r];
  }
  __syncthreads();
  if (p.vec_epi) {   // block-uniform
    for (int e = tid; e < BM * BN / 4; e += NT) {
      const int row = e / (BN / 4), cc = (e - row * (BN / 4)) * 4;
      const int m = m0 + row, co = n0 + cc;
      if (m < p.M && co < p.Cout) epilogue_store4(p, bsel, *reinterpret_cast<const f32x4*>(Cs + row * LDC + cc), m, co, DH, DW, ph, pw);
    }
    return;
  }
  for (int e = tid; e < BM * BN; e += NT) {
    const int row = e / BN, cc = e - row * BN;
    const int m = m0 + row, co = n0 + cc;
    if (m < p.M && co < p.Cout) epilogue_store(p, Cs[row * LDC + cc], bsel ? bsel[co] : 0.f, m, co, DH, DW, ph, pw);
  }
}

// ------------------------------------------------------------------------------------------------
// <= 16 OUTPUT CHANNELS on large maps: the 12-channel tail of the synthesis transform (subpel_conv3x3(N, 3, 2): 128 -> 12 + PixelShuffle,
// /root/reference/models/CLC_run.py:351).  On the 32-column tiles 20 of 32 MFMA columns were padding (163 us for 9.7 padded GFLOP;
// 128 x 32 tiles in round 3: 147 -> 115 us).  Here the N = 16 shape of the f32 MFMA: v_mfma_f32_16x16x4_f32 (16 x 16 x 4 in 32 cycles:
// the same FLOP rate, half the columns) — 12 of 16 columns useful.  256 x 16 tiles on 4 waves; a wave owns 64 rows = four independent
// 16 x 16 accumulators (the 16x16x4 MFMA needs >= 2: 40-cycle dependent latency), staged by LDS-DMA like conv_igemm_dma2_kernel
// (same slot swizzle, same row-origin + scalar tap-delta addressing).  Lane (i = lane & 15, g = lane >> 4) reads the 16-B chunk 4 t + g
// of its row and feeds four MFMAs from it, so MFMA step (t, s) contracts k in {16 t + 4 g + s : g = 0..3} — ANOTHER summation order
// than the 32-column kernels; the layer takes this kernel by its shape alone (never by the batch), so an image's bits stay
// batch-independent, and the decoder's x_hat still equals the encoder-side reconstruction.
template <int KS>
__global__ __launch_bounds__(256, 2)
void conv_igemm_n16_kernel(const ConvParams p) {
  constexpr int BM = 256, NT = 256, A_P = 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][256][32], slot-swizzled
  float* Bs = smem + 2 * BM * BK;    // [2][16][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM;
  const int DH = p.OH, DW = p.OW;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  unsigned a_base[A_P], a_mask[A_P];
  int a_c4[A_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int r = (tid + i * NT) >> 3;
    const RowState rs = make_row<false>(p, m0 + r, DH, DW, 0, 0);
    a_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
    unsigned mask = 0;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const int iy = rs.y0 + j, ix = rs.x0 + q;
        mask |= (rs.ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (1u << (j * 3 + q)) : 0u;
      }
    a_mask[i] = mask;
    a_base[i] = ((unsigned)(rs.base + rs.y0 * p.W + rs.x0) * (unsigned)p.ldx + (unsigned)a_c4[i]) * 4u;
  }
  const int b_r = tid >> 3, b_c4 = ((tid & 7) ^ ((b_r >> 1) & 7)) * 4;     // threads 0..127: the 16 filter rows
  const bool b_ok = tid < 128 && b_r < p.Cout;
  const unsigned b_base = ((unsigned)b_r * (unsigned)p.ldw + (unsigned)b_c4) * 4u;
  const int wave_row = wave * 8;

  f32x4 acc[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int total = KS * KS * p.kc_tiles;
  int tj = 0, ti = 0, kc = 0;
  auto advance = [&]() {
    const int kc1 = kc + 1;
    const bool w1 = kc1 == p.kc_tiles;
    kc = w1 ? 0 : kc1;
    const int ti1 = ti + (w1 ? 1 : 0);
    const bool w2 = ti1 == KS;
    ti = w2 ? 0 : ti1;
    tj += w2 ? 1 : 0;
  };
  unsigned s_bit, s_adelta, s_bdelta; int s_cleft; bool s_en;
  auto set_fetch = [&](bool en) {
    s_en = en;
    s_bit = 1u << (tj * 3 + ti);
    s_cleft = p.Cin - kc * BK;
    s_adelta = (unsigned)((tj * p.W + ti) * p.ldx + kc * BK) * 4u;
    s_bdelta = (unsigned)((tj * KS + ti) * p.Cin + kc * BK) * 4u;
  };
  auto dma_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const bool ok = s_en && (a_mask[i] & s_bit) != 0u && a_c4[i] < s_cleft;
      dma16(xr, As + (buf * BM + wave_row + i * (NT / 8)) * BK, ok ? a_base[i] + s_adelta : kOOB);
    }
    if (wave < 2) dma16(wr, Bs + (buf * 16 + wave_row) * BK, (s_en && b_ok && b_c4 < s_cleft) ? b_base + s_bdelta : kOOB);   // wave-uniform branch
  };
  set_fetch(true);
  dma_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int li = lane & 15, g = lane >> 4, sw = (li >> 1) & 7;
  int fo[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) fo[t] = ((4 * t + g) ^ sw) * 4;
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    advance();
    set_fetch(it + 1 < total);
    dma_tile(buf ^ 1);
    const float* Ab = As + (buf * BM + wave * 64 + li) * BK;
    const float* Bb = Bs + (buf * 16 + li) * BK;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 a[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(Ab + rb * 16 * BK + fo[t]);   // (rows 16 rb + li: the swizzle term (row >> 1) & 7 is li's)
      const f32x4 b = *reinterpret_cast<const f32x4*>(Bb + fo[t]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][s], b[s], acc[rb], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // epilogue: lane (col = lane & 15, g) holds rows 4 g + r of each 16-row block
  const int co = li;
  if (co < p.Cout) {
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 64 + rb * 16 + 4 * g + r;
        if (m < p.M) epilogue_store(p, acc[rb][r], bv, m, co, DH, DW, 0, 0);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT kernel for the large-map 1x1 convolutions / linears (KS = 1, stride 1; 128 x 64 tiles, 8 waves as 4 x 2).
// Why (r3 ablation of conv_igemm_dma2_kernel<128,64,4,2,*,1,0> on 64 -> 192 @ 8x128x128, 70.8 us in all): without its MFMAs 54.8,
// without its result stores 41.0, without its operand DMA 61.2 — and with NONE of the three still 18.3 us.  A tile of these layers is
// 2-8 K-steps: the per-tile fixed work (row addressing, the first operand latency, the LDS round trip of the C tile, barriers, waiting
// for the stores to retire before the workgroup's slot is handed on) is as large as the arithmetic, and nothing overlaps with
// anything: all 768 resident workgroups load, then compute, then store, in step.  Here a workgroup stays resident and walks its tiles
//   (M-tile mt = blockIdx.x, + gridDim.x, ...) x (all N-tiles of that M-tile, so the A rows are re-read from L2) x (K-steps)
// as ONE flat sequence of K-steps through a 3-slot LDS ring: the DMA pieces of step s + 2 are issued at the top of step s whatever
// tile they belong to, so the next tile's operands arrive under this tile's MFMAs and epilogue; the epilogue is per WAVE, straight
// from the accumulator registers (a 32 x 32 block: 16 dword stores of two 128-B row pieces each) — no C tile in LDS, no barrier,
// so the two workgroups of a CU (72 KB each) and the waves inside one drift apart instead of marching in phase; result stores are
// never waited for while the operands they stand in front of (the vector-memory counter is in issue order) are not yet needed.
// Same K order and MFMA sequence per output element as the tiled kernels -> same bits.
// Host-side conditions: M % 128 == 0, Cout % 64 == 0, Cin % 4 == 0, kc_tiles >= 2, one filter set, no PixelShuffle store.
__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n) for the handful of counts the persistent kernel uses
  if (n >= 63) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");
  else if (n == 35) asm volatile("s_waitcnt vmcnt(35) lgkmcnt(0)" ::: "memory");
  else if (n == 19) asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
}
template <int OP>
__global__ __launch_bounds__(512, 4)
void conv_igemm_p1x1_kernel(const ConvParams p, int tiles_m, int tiles_n) {
  constexpr int BM = 128, BN = 64, NT = 512, NS = 3, SLOT = (BM + BN) * BK;
  constexpr int P = 3;   // DMA pieces per wave and K-step: 2 of the A tile (16 rows), 1 of the B tile (8 rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NS][BM + BN][BK], slot-swizzled (see conv_igemm_dma_kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  // per-lane parts of the DMA source offsets (bytes): row r of the tile, 16-B chunk chosen for this lane's LDS slot
  unsigned a_lane[2], b_lane;
  int a_c4[2], b_c4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (tid + i * NT) >> 3;
    a_c4[i] = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
    a_lane[i] = ((unsigned)r * (unsigned)p.ldx + (unsigned)a_c4[i]) * 4u;
  }
  {
    const int r = tid >> 3;
    b_c4 = ((tid & 7) ^ ((r >> 1) & 7)) * 4;
    b_lane = ((unsigned)r * (unsigned)p.ldw + (unsigned)b_c4) * 4u;
  }
  const int wave_row = wave * 8;
  // the bias vector sits in LDS behind the ring: an epilogue's bias read must not be a vector-memory load (its wait would drain the
  // DMA pieces in flight: the counter is in issue order)
  float* bias_s = smem + NS * SLOT;
  for (int i = tid; i < p.Cout; i += NT) bias_s[i] = p.bias ? p.bias[i] : 0.f;
  const int my_mt = (tiles_m - (int)blockIdx.x + G - 1) / G;
  const int total = my_mt * tiles_n * p.kc_tiles;

  struct Cur { int mt, nt, kt, n; };
  auto step = [&](Cur& c) {   // branch-free
    const int kt1 = c.kt + 1;
    const bool w1 = kt1 == p.kc_tiles;
    c.kt = w1 ? 0 : kt1;
    const int nt1 = c.nt + (w1 ? 1 : 0);
    const bool w2 = nt1 == tiles_n;
    c.nt = w2 ? 0 : nt1;
    c.mt += w2 ? G : 0;
    ++c.n;
  };
  auto dma = [&](int slot, const Cur& c) {
    const bool en = c.n < total;
    const int cleft = p.Cin - c.kt * BK;
    const unsigned ao = ((unsigned)(c.mt * BM) * (unsigned)p.ldx + (unsigned)(c.kt * BK)) * 4u;
    const unsigned bo = ((unsigned)(c.nt * BN) * (unsigned)p.ldw + (unsigned)(c.kt * BK)) * 4u;
    float* base = smem + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(xr, base + (wave_row + i * 64) * BK, (en && a_c4[i] < cleft) ? a_lane[i] + ao : kOOB);
    dma16(wr, base + (BM + wave_row) * BK, (en && b_c4 < cleft) ? b_lane + bo : kOOB);
  };

  const int lr = lane & 31, hh = lane >> 5, sw = (lr >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int t8 = 0; t8 < 4; ++t8) fo[t8] = ((2 * t8 + hh) ^ sw) * 4;
  f32x4 af[2], bf[2];
  auto read_frag = [&](int slot, int t8, f32x4& a, f32x4& b) {
    const float* base = smem + slot * SLOT;
    a = *reinterpret_cast<const f32x4*>(base + (wm * 32 + lr) * BK + fo[t8]);
    if (OP == 1) a = a * a;
    b = *reinterpret_cast<const f32x4*>(base + (BM + wn * 32 + lr) * BK + fo[t8]);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mfma4 = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ss], b[ss], acc, 0, 0, 0);
  };

  // per-wave epilogue straight from the accumulator registers (epilogue_regs)
  const RegEpi re = make_reg_epi(p);
  const int S = p.y_pre ? 32 : 16;   // store instructions of one epilogue
  auto epilogue = [&](int mt, int nt) {
    const int co = nt * BN + wn * 32 + lr;
    epilogue_regs(p, re, acc, bias_s[co], (unsigned)(mt * BM + wm * 32 + 4 * hh), co);
  };


  Cur cf{(int)blockIdx.x, 0, 0, 0}, cc = cf;
  dma(0, cf); step(cf);
  dma(1, cf); step(cf);
  asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");   // step 0 has landed (and the bias vector is written); step 1 stays in flight
  __builtin_amdgcn_s_barrier();
  int slot = 0, slot2 = 2, e_prev = 0;
  for (int s = 0; s < total; ++s) {
    dma(slot2, cf); step(cf);                         // step s + 2 into the slot step s - 1 was read from (everyone is past that barrier)
    read_frag(slot, 0, af[0], bf[0]);
    read_frag(slot, 1, af[1], bf[1]);
    mfma4(af[0], bf[0]);
    read_frag(slot, 2, af[0], bf[0]);
    mfma4(af[1], bf[1]);
    read_frag(slot, 3, af[1], bf[1]);
    mfma4(af[0], bf[0]);
    mfma4(af[1], bf[1]);
    const bool last = cc.kt == p.kc_tiles - 1;        // block-uniform
    if (last) {
      epilogue(cc.mt, cc.nt);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    // Step s + 1 (issued at the top of step s - 1) must have landed.  Issued after it: the pieces of step s + 2 and the stores of
    // the epilogues of steps s - 1 and s, which may stay in flight (an epilogue with operand loads has already waited for everything
    // older than them).
    const int e_cur = last ? 1 : 0;
    wait_vm(P + S * (e_prev + e_cur));
    __builtin_amdgcn_s_barrier();
    e_prev = e_cur;
    step(cc);
    slot = slot == 2 ? 0 : slot + 1;
    slot2 = slot2 == 2 ? 0 : slot2 + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-fill pieces past the last step must not land in a successor's LDS
}

// ------------------------------------------------------------------------------------------------
// Small maps (<= 16x16 per image): latency, not MFMA rate, is what matters (a 64->64 3x3 at 8x16x16 is 0.15 GFLOP).
// One 32 x BN output tile per workgroup of KW = 8 waves; wave w owns K-tiles w, w+8, ... and loads its MFMA fragments
// STRAIGHT from global memory into registers in operand layout (lane (i,h) needs 4 consecutive k of its own row i:
// one buffer_load_dwordx4) — no LDS staging, no barrier and no ds round trip in the loop, all of a wave's K-tiles in
// flight at once.  Partial tiles are combined through LDS in the fixed order ((w0+w1)+(w2+w3))+((w4+w5)+(w6+w7)).
// KW = waves per tile = K-split factor: 8, or 4 for layers of at most 4 K-tiles (1x1 convolutions with Cin <= 128: most of
// the SWAtten / Swin linears) — half of an 8-wave workgroup would idle there and the wide ones (128 -> 512 on 4096 stacked
// rows = 1024 tiles) would need 4 rounds of 512-thread workgroups instead of one round of 256-thread ones.
// PF = K-tiles a wave keeps in flight.  KW = 8: PF = 3 needs ~200-256 VGPRs (one 8-wave workgroup per CU); PF = 1 fits 128 (two per
// CU: the other workgroup's loads cover this one's latency instead of its own prefetch) — CLC_TUNE_SPLITK_PF picks.
// OPS: the launch has operand arithmetic (GDN's squared input, or a fused activation derivative on the gathered operand).  The plain
// instantiation carries none of that code: with the block-uniform `if (fuse_act)` + the activation switch inlined around every one of
// its MFMA groups, the K loop of this (latency-bound, 960-launches-per-step) kernel had compiled to ~180 branch instructions for 32 MFMAs.
template <int BN, bool TR, int KW, int PF, bool OPS>
__global__ __launch_bounds__(64 * KW, KW == 8 ? (PF >= 2 ? 1 : 4) : 4)
void conv_igemm_splitk_kernel(const ConvParams p) {
  constexpr int BM = 32, TN = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // combine buffer [KW][TN][16][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;
  const bool half = p.transposed && p.stride == 2;
  const int DH = half ? p.OH / 2 : p.OH, DW = half ? p.OW / 2 : p.OW;
  const TapGrid tg = make_taps(p, ph, pw);

  const int fset = p.group_rows ? min(m0 / p.group_rows, 3) : 0;   // block-uniform: tiles never straddle two filter sets
  const float* wsel = fset == 0 ? p.w : (fset == 1 ? p.w2 : (fset == 2 ? p.w3 : p.w4));
  const float* bsel = fset == 0 ? p.bias : (fset == 1 ? p.bias2 : (fset == 2 ? p.bias3 : p.bias4));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsel), 0, p.w_bytes, 0x00020000);

  const RowState row = make_row<TR>(p, m0 + li, DH, DW, ph, pw);
  unsigned b_row_off[TN];
  bool b_ok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + j * 32 + li;
    b_ok[j] = co < p.Cout;
    b_row_off[j] = (unsigned)co * (unsigned)p.ldw;
  }
  const bool sq = OPS && p.in_op == CLC_IN_SQUARE;
  const int total = tg.nkh * tg.nkw * p.kc_tiles;

  const bool fuse_act = OPS && p.xs != nullptr;
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fuse_act ? p.xs : p.x), 0, fuse_act ? p.xs_bytes : p.x_bytes, 0x00020000);
  f32x4 af[PF][4], sf[PF][4], bf[PF][TN][4];
  auto load_tile = [&](int slot, int it) {   // it = global K-tile index -> (tap, kc); fragment t8 covers k = 8*t8 + 4h .. +3
    const int t = it / p.kc_tiles, kc = it - t * p.kc_tiles;
    const int tj = t / tg.nkw, ti = t - tj * tg.nkw;
    const int kh = tg.kh0 + tg.step * tj, kw = tg.kw0 + tg.step * ti;
    const unsigned tap_off = (unsigned)((kh * p.ks + kw) * p.Cin);
    const unsigned pix = a_pixel<TR>(p, row, kh, kw);
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      const int c = kc * BK + t8 * 8 + 4 * h;
      const bool c_ok = c < p.Cin;
      af[slot][t8] = buf_load4(xr, pix_off(pix, p.ldx, c, c_ok));
      if (OPS && fuse_act) sf[slot][t8] = buf_load4(sr, pix_off(pix, p.ldxs, c, c_ok));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[slot][j][t8] = buf_load4(wr, (b_ok[j] && c_ok) ? (b_row_off[j] + tap_off + (unsigned)c) * 4u : kOOB);
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // all of this wave's first PF tiles go in flight at once; later ones (rare: K > 8*PF tiles) are loaded as slots free up
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (wave + s * KW < total) load_tile(s, wave + s * KW);
  int it = wave;
  for (int base = 0; it < total; base += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      if (it < total) {      // wave-uniform
#pragma unroll
        for (int t8 = 0; t8 < 4; ++t8) {
          f32x4 a = af[s][t8];
          if constexpr (OPS) {
            if (sq) a = a * a;
            if (fuse_act) a = a * act_deriv4(sf[s][t8], p.xs_act, p.xs_pre);
          }
#pragma unroll
          for (int ss = 0; ss < 4; ++ss)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ss], bf[s][j][t8][ss], acc[j], 0, 0, 0);
        }
        if (it + PF * KW < total) load_tile(s, it + PF * KW);
      }
      it += KW;
    }
  }

  // fixed-order combine through LDS, then a compact coalesced epilogue over the 32 x BN tile
  float* red = smem;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * TN + j) * 16 + r) * 64 + lane] = acc[j][r];
  __syncthreads();
  if (p.vec_epi) {   // block-uniform: 4 consecutive channels = 4 consecutive lanes of one (wave, j, r) row of the buffer
    for (int e = tid; e < BM * BN / 4; e += 64 * KW) {
      const int rowi = e / (BN / 4), cc = (e - rowi * (BN / 4)) * 4;
      const int j = cc >> 5, r = (rowi & 3) + 4 * (rowi >> 3), ln = (cc & 31) + 32 * ((rowi >> 2) & 1);
      f32x4 q[KW];
#pragma unroll
      for (int w = 0; w < KW; ++w) q[w] = *reinterpret_cast<const f32x4*>(red + ((w * TN + j) * 16 + r) * 64 + ln);
      f32x4 v = (q[0] + q[1]) + (q[2] + q[3]);
      if constexpr (KW == 8) v = v + ((q[4] + q[5]) + (q[6] + q[7]));
      const int m = m0 + rowi, co = n0 + cc;
      if (m < p.M && co < p.Cout) epilogue_store4(p, bsel, v, m, co, DH, DW, ph, pw);
    }
    return;
  }
  for (int e = tid; e < BM * BN; e += 64 * KW) {
    const int rowi = e / BN, cc = e - rowi * BN;         // rowi = (r&3) + 8*(r>>2) + 4*h ; cc = j*32 + (lane&31)
    const int j = cc >> 5, r = (rowi & 3) + 4 * (rowi >> 3), ln = (cc & 31) + 32 * ((rowi >> 2) & 1);
    float q[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) q[w] = red[((w * TN + j) * 16 + r) * 64 + ln];
    float v = (q[0] + q[1]) + (q[2] + q[3]);
    if constexpr (KW == 8) v = v + ((q[4] + q[5]) + (q[6] + q[7]));
    const int m = m0 + rowi, co = n0 + cc;
    if (m < p.M && co < p.Cout) epilogue_store(p, v, bsel ? bsel[co] : 0.f, m, co, DH, DW, ph, pw);
  }
}

// Finish of a K-split launch: the `ksplit` partial tiles of every output element added in split order (fixed -> reproducible), then the
// ordinary epilogue (epilogue_store4: bias, residual / gates, activation ...), one thread per 4 channels of a pixel.
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvParams p, int classes) {
  const int c4n = p.Cout >> 2;
  const long per_class = (long)p.M * c4n, total = per_class * classes;
  const bool half = p.transposed && p.stride == 2;
  const int DH = half ? p.OH / 2 : p.OH, DW = half ? p.OW / 2 : p.OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cls = (int)(i / per_class);
    const long r = i - (long)cls * per_class;
    const int m = (int)(r / c4n), co = (int)(r - (long)m * c4n) * 4;
    const float* src = p.partial + ((size_t)(cls * p.ksplit) * p.M + m) * p.Cout + co;
    f32x4 acc = *reinterpret_cast<const f32x4*>(src);
    for (int s2 = 1; s2 < p.ksplit; ++s2) acc = acc + *reinterpret_cast<const f32x4*>(src + (size_t)s2 * p.M * p.Cout);
    const int fset = p.group_rows ? min(m / p.group_rows, 3) : 0;
    const float* bsel = fset == 0 ? p.bias : (fset == 1 ? p.bias2 : (fset == 2 ? p.bias3 : p.bias4));
    epilogue_store4(p, bsel, acc, m, co, DH, DW, cls >> 1, cls & 1);
  }
}
// Long-K FORWARD 3x3 layers on <= 16x16 maps in training (the slice-parameter nets: 384..704 -> 224 on 2048 / 4096 rows): the split-K
// family re-fetches both operands per 32-row workgroup (1.1 GB through L2 for 640 -> 224, which is what bounds it); 128x128 LDS tiles
// move a third of that, with the K range split so that >= 256 workgroups exist.  The split looks at the row count, so the caller must
// allow a batch-dependent summation order (clc_conv_desc.batch_variant_ok: training forward passes; never the codec).  The same
// routing for the data gradients of these layers measured slower than their 64x64 tiles (121.6 -> 146.3 us).
// (heavy128_split > 1 <=> the layer takes this path.)
static int heavy128_split(const ConvParams& p, int classes) {
  if (!clc_tuning[CLC_TUNE_HEAVY128] || !p.batch_variant_ok || p.transposed || !p.vec_epi || p.in_op != CLC_IN_NONE || p.xs || p.shuffle || classes != 1) return 1;
  if (p.ks != 3 || p.stride != 1 || p.OH * p.OW > 256 || p.Cout < 128 || p.Cout % 32 || p.M % 128 || p.M < 2048) return 1;
  const int ksteps = 9 * p.kc_tiles;
  if (ksteps < 108) return 1;
  const long wgs = (long)(p.M / 128) * ((p.Cout + 127) / 128);
  int k = (int)((256 + wgs - 1) / wgs);
  if (k > 8) k = 8;
  while (k > 1 && ksteps / k < 16) --k;
  return k;
}
// how many ways clc_conv2d would split the K range of a 64x64-tile data-gradient launch (1: not at all)
static int conv_ksplit(const ConvParams& p, int classes) {
  { const int k128 = heavy128_split(p, classes); if (k128 > 1) return k128; }
  if (!clc_tuning[CLC_TUNE_DGRAD_SPLITK] || !p.transposed || !p.vec_epi || p.in_op != CLC_IN_NONE || p.xs || p.shuffle) return 1;
  if (p.M % 64 || p.Cout % 64) return 1;
  const long wgs = (long)(p.M / 64) * (p.Cout / 64) * classes;
  const int ksteps = (p.transposed && p.stride == 2 ? ((p.ks + 1) / 2) * ((p.ks + 1) / 2) : p.ks * p.ks) * p.kc_tiles;   // (upper bound for stride 2)
  // (measured, r3: 512 -> 128 @ 8x32x32 123 -> 102 us, 512 -> 320 @ 8x16x16 122 -> 81 us; a 36-step layer got slower: long K ranges only)
  // (r4: the 63-step data gradients of the un-paired lrp nets — 224 -> 448..704 on 2 048 rows, 288-352 tiles — 704 -> 224: 78.6 -> 65.1 us)
  const int min_steps = clc_tuning[CLC_TUNE_DGRAD_SPLITK] > 1 ? clc_tuning[CLC_TUNE_DGRAD_SPLITK] : 56;   // (key 11 > 1: the K-step threshold itself, for A/B)
  if (wgs >= 384 || ksteps < min_steps) return 1;
  int k = (int)((640 + wgs - 1) / wgs);
  if (k > 4) k = 4;
  while (k > 1 && ksteps / k < 24) --k;
  return k;
}

template <int BM, int BN, int WM, int WN, bool TR>
int launch_t(const ConvParams& p, int classes, hipStream_t st) {
  dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, classes);
  const size_t lds = (size_t)2 * (BM + BN) * LD * sizeof(float);
  static_assert(BM * (BN + 4) <= 2 * (BM + BN) * LD, "C tile must fit in the operand tiles' LDS");
  static PerDeviceOnce attr_once;  // >64 KiB of dynamic LDS needs an explicit opt-in (first call happens before any graph capture)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, TR>), grid, dim3(64 * WM * WN), lds, st, p);
  CLC_LAUNCH_CHECK();
  return (1 << 20) | (WM << 16) | (WN << 12) | (BM << 3) | (BN >> 5);  // kernel-variant id (> 1): family 1 = conv_igemm_kernel<BM,BN,WM,WN>
}
template <int BM, int BN, int WM, int WN, bool TR>
int launch_dma_t(const ConvParams& p, int classes, hipStream_t st) {
  dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, classes);
  constexpr size_t lds_ab = (size_t)2 * (BM + BN) * BK * sizeof(float), lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_kernel<BM, BN, WM, WN, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WM, WN, TR>), grid, dim3(64 * WM * WN), lds, st, p);
  CLC_LAUNCH_CHECK();
  return (2 << 20) | (WM << 16) | (WN << 12) | (BM << 3) | (BN >> 5);  // family 2 = conv_igemm_dma_kernel<BM,BN,WM,WN>
}
template <int BM, int BN, int WM, int WN, bool TR, int KS, int OP = 0, bool BF = false>
int launch_dma2_t(const ConvParams& p, int classes, hipStream_t st) {
  dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, classes);
  constexpr size_t lds_ab = (size_t)2 * (BM + BN) * BK * sizeof(float), lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma2_kernel<BM, BN, WM, WN, TR, KS, OP, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  ConvParams q = p;
  const int xm = clc_tuning[CLC_TUNE_XCD_MAP];
  q.xcd_map = grid.y > 1 && grid.x % 8 == 0 && (xm == 2 || (xm == 1 && KS == 1));
  q.reg_epi = clc_tuning[CLC_TUNE_REG_EPI] && reg_epi_ok(p, BM, BN);
  if (p.ksplit > 1) {   // (set by clc_conv2d for the 64x64 tile only)
    grid.z = classes * p.ksplit;
    q.xcd_map = 0;
  }
  hipLaunchKernelGGL((conv_igemm_dma2_kernel<BM, BN, WM, WN, TR, KS, OP, BF>), grid, dim3(64 * WM * WN), lds, st, q);
  CLC_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    const long total = (long)p.M * (p.Cout / 4) * classes;
    long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, q, classes);
    CLC_LAUNCH_CHECK();
  }
  return ((KS == 1 ? 5 : 4) << 20) | (OP << 24) | ((BF ? 1 : 0) << 26) | (WM << 16) | (WN << 12) | (BM << 3) | (BN >> 5);  // family 4 / 5 = conv_igemm_dma2_kernel<BM,BN,WM,WN,TR,3 / 1,OP>
}
// <= 16 output channels on large maps (forward, stride 1, one filter set): see conv_igemm_n16_kernel
int launch_n16(const ConvParams& p, int classes, hipStream_t st) {
  if (!clc_tuning[CLC_TUNE_N16] || classes != 1 || p.transposed || p.stride != 1 || p.Cout > 16 || p.group_rows || p.xs || p.in_op != CLC_IN_NONE || p.ksplit > 1) return 0;
  if (p.ks != 3) return 0;
  dim3 grid((p.M + 255) / 256);
  const size_t lds = (size_t)2 * (256 + 16) * BK * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_n16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm_n16_kernel<3>), grid, dim3(256), lds, st, p);
  CLC_LAUNCH_CHECK();
  return (10 << 20) | (4 << 16) | (256 << 3);   // family 10 = conv_igemm_n16_kernel<KS>
}
// persistent 1x1 kernel: eligibility + launch; returns 0 when the layer does not qualify (the caller falls through to the tiled kernels)
int launch_p1x1(const ConvParams& p, int classes, hipStream_t st) {
  if (!clc_tuning[CLC_TUNE_P1X1] || classes != 1 || p.ks != 1 || p.stride != 1 || p.shuffle || p.group_rows || p.xs) return 0;
  if (p.M % 128 || p.Cout % 64 || p.kc_tiles < 2 || !(p.in_op == CLC_IN_NONE || p.in_op == CLC_IN_SQUARE)) return 0;
  const int tiles_m = p.M / 128, tiles_n = p.Cout / 64;
  if (!reg_epi_ok(p, 128, 64)) return 0;
  if (tiles_m < (clc_tuning[CLC_TUNE_P1X1] >= 2 ? 256 : 512)) return 0;   // (a workgroup needs several M-tiles to pipeline across)
  const size_t lds = (size_t)3 * (128 + 64) * BK * sizeof(float) + (size_t)p.Cout * sizeof(float);
  if (lds > 80 * 1024) return 0;
  const int grid = tiles_m < 512 ? tiles_m : 512;   // two resident workgroups per CU
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_p1x1_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_p1x1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  }
  if (p.in_op == CLC_IN_SQUARE) hipLaunchKernelGGL((conv_igemm_p1x1_kernel<1>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n);
  else hipLaunchKernelGGL((conv_igemm_p1x1_kernel<0>), dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n);
  CLC_LAUNCH_CHECK();
  return (8 << 20) | ((p.in_op == CLC_IN_SQUARE ? 1 : 0) << 24) | (4 << 16) | (2 << 12) | (128 << 3) | (64 >> 5);   // family 8 = conv_igemm_p1x1_kernel<OP>
}
template <int BM, int BN, int WM, int WN>
int launch(const ConvParams& p, int classes, hipStream_t st) {
  static const int use_dma = getenv("CLC_DMA") ? atoi(getenv("CLC_DMA")) : 1;   // CLC_DMA=0: register staging everywhere (A/B knob)
  static const int dma_small = getenv("CLC_DMA_SMALL") ? atoi(getenv("CLC_DMA_SMALL")) : 1;
  if (use_dma && p.in_op == CLC_IN_SQUARE && p.xs == nullptr && p.ks == 1 && p.stride == 1 && (BM >= 128 || dma_small) && clc_tuning[CLC_TUNE_DMA_LOOP] == 2)
    return p.transposed ? launch_dma2_t<BM, BN, WM, WN, true, 1, 1>(p, classes, st) : launch_dma2_t<BM, BN, WM, WN, false, 1, 1>(p, classes, st);
  if (use_dma && p.in_op == CLC_IN_NONE && p.xs == nullptr && (BM >= 128 || dma_small) && clc_tuning[CLC_TUNE_DMA_LOOP] == 2)
    return p.ks == 1 && p.stride == 1
               ? (p.transposed ? launch_dma2_t<BM, BN, WM, WN, true, 1>(p, classes, st) : launch_dma2_t<BM, BN, WM, WN, false, 1>(p, classes, st))
               : (p.bf16   // reduced-precision mode: the 3x3 layers of the LDS-tiled family (maps larger than 16x16 = the transforms)
                      ? (p.transposed ? launch_dma2_t<BM, BN, WM, WN, true, 3, 0, true>(p, classes, st) : launch_dma2_t<BM, BN, WM, WN, false, 3, 0, true>(p, classes, st))
                      : (p.transposed ? launch_dma2_t<BM, BN, WM, WN, true, 3>(p, classes, st) : launch_dma2_t<BM, BN, WM, WN, false, 3>(p, classes, st)));
  if (use_dma && p.in_op == CLC_IN_NONE && p.xs == nullptr && (BM >= 128 || dma_small))   // no input prologue -> the tiles can go straight to LDS
    return p.transposed ? launch_dma_t<BM, BN, WM, WN, true>(p, classes, st) : launch_dma_t<BM, BN, WM, WN, false>(p, classes, st);
  return p.transposed ? launch_t<BM, BN, WM, WN, true>(p, classes, st) : launch_t<BM, BN, WM, WN, false>(p, classes, st);
}

template <int BN, bool TR, int KW, int PF, bool OPS>
int launch_splitk_t(const ConvParams& p, int classes, hipStream_t st) {
  dim3 grid((p.M + 31) / 32, (p.Cout + BN - 1) / BN, classes);
  const size_t lds = (size_t)KW * (BN / 32) * 16 * 64 * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_splitk_kernel<BN, TR, KW, PF, OPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_igemm_splitk_kernel<BN, TR, KW, PF, OPS>), grid, dim3(64 * KW), lds, st, p);
  CLC_LAUNCH_CHECK();
  return (3 << 20) | (KW << 16) | (32 << 3) | (BN >> 5);  // family 3 = conv_igemm_splitk_kernel<BN,TR,KW>
}
template <int BN, bool TR, int KW, int PF>
int launch_splitk_o(const ConvParams& p, int classes, hipStream_t st) {
  return (p.xs != nullptr || p.in_op == CLC_IN_SQUARE) ? launch_splitk_t<BN, TR, KW, PF, true>(p, classes, st) : launch_splitk_t<BN, TR, KW, PF, false>(p, classes, st);
}
template <int BN>
int launch_splitk(const ConvParams& p, int classes, hipStream_t st) {
  // the K-split factor is a function of the layer shape alone (never of the batch), like the family itself
  static const int kw4 = getenv("CLC_SPLITK_KW4") ? atoi(getenv("CLC_SPLITK_KW4")) : 1;   // 0: always 8 waves (A/B knob)
  if (kw4 && p.ks * p.ks * p.kc_tiles <= 4)
    return p.transposed ? launch_splitk_o<BN, true, 4, 1>(p, classes, st) : launch_splitk_o<BN, false, 4, 1>(p, classes, st);
  if (clc_tuning[CLC_TUNE_SPLITK_PF] == 1)
    return p.transposed ? launch_splitk_o<BN, true, 8, 1>(p, classes, st) : launch_splitk_o<BN, false, 8, 1>(p, classes, st);
  return p.transposed ? launch_splitk_o<BN, true, 8, 3>(p, classes, st) : launch_splitk_o<BN, false, 8, 3>(p, classes, st);
}

// small-Cin (image, Cin<=4, unaligned) direct convolution: one thread per (pixel, 4 output channels)
__global__ void conv_direct_small_kernel(const ConvParams p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cgroups = (p.Cout + 3) / 4;
  const long total = (long)p.M * cgroups;
  if (idx >= total) return;
  const int cg = (int)(idx % cgroups);
  const int m = (int)(idx / cgroups);
  const int n = m / (p.OH * p.OW), rr = m - n * (p.OH * p.OW);
  const int oy = rr / p.OW, ox = rr - oy * p.OW;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kh = 0; kh < p.ks; ++kh) {
    const int iy = oy * p.stride - p.pad + kh;
    if (iy < 0 || iy >= p.H) continue;
    for (int kw = 0; kw < p.ks; ++kw) {
      const int ix = ox * p.stride - p.pad + kw;
      if (ix < 0 || ix >= p.W) continue;
      const float* xp = p.x + (size_t)((n * p.H + iy) * p.W + ix) * p.ldx;
      for (int ci = 0; ci < p.Cin; ++ci) {
        float xv = xp[ci];
        if (p.in_op == CLC_IN_SQUARE) xv *= xv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cg * 4 + q;
          if (co < p.Cout) acc[q] = fmaf(xv, p.w[(size_t)co * p.ldw + (kh * p.ks + kw) * p.Cin + ci], acc[q]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int co = cg * 4 + q;
    if (co < p.Cout) epilogue_store(p, acc[q], p.bias ? p.bias[co] : 0.f, m, co, p.OH, p.OW, 0, 0);
  }
}

__global__ void filter_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int T, int Cin) {
  // w [Cout][T][Cin] -> wt [Cin][T][Cout]; 32x32 LDS tile per (t, co-tile, ci-tile)
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * T + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wt[((size_t)ci * T + t) * Cout + co] = tile[tx][r];
  }
}

// Batched form: one launch transposes every filter of the model (table resident on the device).
// Workgroup b looks up its entry by binary search over the per-entry first-tile index.
__global__ void filter_transpose_batched_kernel(const clc_transpose_entry* __restrict__ table, int n_entries) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_entries - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile_begin <= b) lo = mid; else hi = mid - 1;
  }
  const clc_transpose_entry e = table[lo];
  const int nci = (e.Cin + 31) / 32, nco = (e.Cout + 31) / 32;
  int local = b - e.tile_begin;
  const int t = local / (nci * nco); local -= t * nci * nco;
  const int co0 = (local / nci) * 32, ci0 = (local % nci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < e.Cout && ci < e.Cin) ? e.w[((size_t)co * e.T + t) * e.Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < e.Cin && co < e.Cout) e.wt[((size_t)ci * e.T + t) * e.Cout + co] = tile[tx][r];
  }
}

}  // namespace

extern "C" int clc_filter_transpose_batched(const clc_transpose_entry* table_dev, int n_entries, int total_tiles, clc_stream_t stream) {
  CLC_CHECK(table_dev && n_entries > 0 && total_tiles > 0, "clc_filter_transpose_batched: bad args");
  hipLaunchKernelGGL(filter_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table_dev, n_entries);
  CLC_LAUNCH_CHECK();
  return 0;
}

// turns the K split on for this launch when the caller provided the scratch for it
static void use_split(ConvParams& p, const clc_conv_desc* d, int classes) {
  const int k = conv_ksplit(p, classes);
  if (k > 1 && d->workspace && d->workspace_bytes >= (size_t)k * classes * p.M * p.Cout * sizeof(float) && aligned16(d->workspace) &&
      clc_tuning[CLC_TUNE_DMA_LOOP] == 2) {
    p.ksplit = k;
    p.partial = (float*)d->workspace;
  }
}

// validates a descriptor and fills the kernel parameters (shared by clc_conv2d and clc_conv2d_workspace_bytes)
static int fill_params(const clc_conv_desc* d, ConvParams& p, int& classes) {
  CLC_CHECK(d && d->x && d->w && d->y, "clc_conv2d: null pointer");
  CLC_CHECK(d->ks == 1 || d->ks == 3, "clc_conv2d: ks must be 1 or 3 (got %d)", d->ks);
  CLC_CHECK(d->stride == 1 || d->stride == 2, "clc_conv2d: stride must be 1 or 2 (got %d)", d->stride);
  CLC_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->OH > 0 && d->OW > 0, "clc_conv2d: bad dims");
  CLC_CHECK(d->ldx >= d->Cin, "clc_conv2d: ldx < Cin");
  if (!d->transposed) {
    CLC_CHECK(d->OH == (d->H + 2 * d->pad - d->ks) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->ks) / d->stride + 1,
              "clc_conv2d: output dims %dx%d inconsistent with input %dx%d ks=%d s=%d pad=%d", d->OH, d->OW, d->H, d->W, d->ks, d->stride, d->pad);
  } else {
    CLC_CHECK(d->H == (d->OH + 2 * d->pad - d->ks) / d->stride + 1 && d->W == (d->OW + 2 * d->pad - d->ks) / d->stride + 1,
              "clc_conv2d(transposed): dY dims %dx%d inconsistent with dX %dx%d", d->H, d->W, d->OH, d->OW);
    CLC_CHECK(d->stride == 1 || (d->OH % 2 == 0 && d->OW % 2 == 0), "clc_conv2d(transposed,s2): odd dX dims");
    CLC_CHECK(!d->shuffle, "clc_conv2d: shuffle with transposed");
  }
  CLC_CHECK(!d->shuffle || d->Cout % 4 == 0, "clc_conv2d: shuffle needs Cout %% 4 == 0");
  CLC_CHECK(d->norm == CLC_NORM_NONE || d->mul, "clc_conv2d: norm without mul");
  const int och = d->shuffle ? d->Cout / 4 : d->Cout;
  CLC_CHECK(d->ldy >= och, "clc_conv2d: ldy < channels");
  const size_t x_bytes = ((size_t)d->N * d->H * d->W - 1) * d->ldx * 4 + (size_t)d->Cin * 4;
  const size_t w_bytes = (size_t)d->Cout * d->ks * d->ks * d->Cin * 4;
  CLC_CHECK(x_bytes < (1ull << 31) && w_bytes < (1ull << 31), "clc_conv2d: tensor larger than 2 GiB");
  CLC_CHECK((size_t)d->N * d->OH * d->OW * (d->shuffle ? 4 : 1) < (1ull << 31), "clc_conv2d: too many pixels");

  p.x = d->x; p.w = d->w; p.bias = d->bias; p.y = d->y; p.mul = d->mul; p.res = d->res; p.y_pre = d->y_pre;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx;
  p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.ldy = d->ldy;
  p.ks = d->ks; p.stride = d->stride; p.pad = d->pad; p.transposed = d->transposed; p.in_op = d->in_op;
  p.act = d->act; p.norm = d->norm; p.shuffle = d->shuffle; p.res_first = d->res_first;
  p.ldm = d->ldm; p.ldr = d->ldr; p.ldp = d->ldp; p.ldw = d->ks * d->ks * d->Cin; p.res_scale = d->res_scale;
  p.kc_tiles = (d->Cin + BK - 1) / BK;
  p.x_bytes = (unsigned)x_bytes; p.w_bytes = (unsigned)w_bytes;
  p.xs = d->xs; p.ldxs = d->ldxs; p.xs_act = d->xs_act; p.xs_pre = d->xs_pre; p.xs_bytes = 0;
  if (d->xs) {
    CLC_CHECK(d->ldxs >= d->Cin && d->ldxs % 4 == 0 && aligned16(d->xs), "clc_conv2d: bad xs (fused activation backward operand)");
    const size_t sb = ((size_t)d->N * d->H * d->W - 1) * d->ldxs * 4 + (size_t)d->Cin * 4;
    CLC_CHECK(sb < (1ull << 31), "clc_conv2d: xs larger than 2 GiB");
    p.xs_bytes = (unsigned)sb;
  }
  {
    auto ok4 = [](const void* ptr, int ld) { return ptr == nullptr || (aligned16(ptr) && ld % 4 == 0); };
    p.vec_epi = !d->shuffle && d->Cout % 4 == 0 && ok4(d->y, d->ldy) && ok4(d->res, d->ldr) && ok4(d->mul, d->ldm) && ok4(d->y_pre, d->ldp) && ok4(d->res_gate, d->ldg) && ok4(d->out_gate, d->ldog);
  }
  classes = 1;
  p.M = d->N * d->OH * d->OW;
  if (d->transposed && d->stride == 2) { classes = 4; p.M = d->N * (d->OH / 2) * (d->OW / 2); }
  p.w2 = d->w2; p.bias2 = d->bias2; p.group_rows = 0; p.pre_deriv = d->pre_deriv;
  p.dma_place = clc_tuning[CLC_TUNE_DMA_PLACE]; p.xcd_map = 0; p.ablate = clc_tuning[CLC_TUNE_ABLATE]; p.reg_epi = 0; p.ksplit = 1; p.partial = nullptr; p.batch_variant_ok = d->batch_variant_ok; p.bf16 = 0;
  p.res_gate = d->res ? d->res_gate : nullptr; p.ldg = d->ldg; p.rg_act = d->res_gate_act; p.rg_pre = d->res_gate_pre;
  p.out_gate = d->out_gate; p.ldog = d->ldog; p.og_act = d->out_gate_act; p.og_pre = d->out_gate_pre;
  CLC_CHECK(!d->out_gate || !d->shuffle, "clc_conv2d: out_gate with shuffle");
  p.w3 = d->w3; p.bias3 = d->bias3; p.w4 = d->w4; p.bias4 = d->bias4;
  if (d->w2) {   // 2 or 4 filter sets on equal parts of the batch; rows are image-major, so the parts split at multiples of M / sets
    const int sets = d->w3 ? 4 : 2;
    CLC_CHECK((d->w3 != nullptr) == (d->w4 != nullptr), "clc_conv2d: w3 and w4 must be given together");
    CLC_CHECK(d->N % sets == 0, "clc_conv2d: %d filter sets need a batch that is a multiple of %d (got N=%d)", sets, sets, d->N);
    CLC_CHECK((d->bias == nullptr) == (d->bias2 == nullptr) && (!d->w3 || ((d->bias == nullptr) == (d->bias3 == nullptr) && (d->bias == nullptr) == (d->bias4 == nullptr))),
              "clc_conv2d: the bias must be given for every filter set or for none");
    CLC_CHECK(aligned16(d->w2) && (!d->w3 || (aligned16(d->w3) && aligned16(d->w4))) && (d->Cin % 4 == 0) && (d->ldx % 4 == 0),
              "clc_conv2d: filter sets need the aligned (Cin %% 4 == 0) path");
    p.group_rows = p.M / sets;
    CLC_CHECK(p.group_rows % 128 == 0, "clc_conv2d: filter sets need (N/sets)*rows-per-image to be a multiple of the largest tile (128), got %d", p.group_rows);
  } else {
    CLC_CHECK(!d->w3 && !d->w4, "clc_conv2d: w3 / w4 without w2");
  }

  return 0;
}

extern "C" size_t clc_conv2d_workspace_bytes(const clc_conv_desc* d) {
  if (!d) return 0;
  ConvParams p;
  int classes = 1;
  if (fill_params(d, p, classes) < 0) return 0;
  const bool vec_ok = (d->Cin % 4 == 0) && (d->ldx % 4 == 0) && aligned16(d->x) && aligned16(d->w);
  if (!vec_ok || d->Cout < 64) return 0;
  const int img_pix = (d->transposed && d->stride == 2) ? (d->OH / 2) * (d->OW / 2) : d->OH * d->OW;
  if (img_pix > 1024) return 0;   // (only the 64x64-tile family splits; whether THIS launch lands there is decided in clc_conv2d — an unused scratch is harmless)
  const int k = conv_ksplit(p, classes);
  return k > 1 ? (size_t)k * classes * p.M * p.Cout * sizeof(float) : 0;
}

extern "C" int clc_conv2d(const clc_conv_desc* d, clc_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  ConvParams p;
  int classes = 1;
  if (fill_params(d, p, classes) < 0) return -1;
  const bool vec_ok = (d->Cin % 4 == 0) && (d->ldx % 4 == 0) && aligned16(d->x) && aligned16(d->w);
  if (!vec_ok) {
    CLC_CHECK(!d->transposed && !d->xs, "clc_conv2d: unaligned/small-Cin path has no transposed / fused-activation mode (Cin=%d ldx=%d)", d->Cin, d->ldx);
    const int cgroups = (d->Cout + 3) / 4;
    const long total = (long)p.M * cgroups;
    hipLaunchKernelGGL(conv_direct_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    CLC_LAUNCH_CHECK();
    return 1;  // variant id of the direct small-Cin kernel
  }
  // Family / tile selection from the PER-IMAGE map and the channel counts only (never the batch size), so an image's
  // summation order — hence its bits — is the same in any batch.
  const int img_pix = (d->transposed && d->stride == 2) ? (d->OH / 2) * (d->OW / 2) : d->OH * d->OW;
  const int C = d->Cout;
  // Maps up to 32x32 with few channels are latency-bound the same way (a 64 -> 64 3x3 at 8x32x32 is 128 tiles of 64x64: half the
  // CUs, one 18-K-tile MFMA chain per wave): tuning key CLC_TUNE_SPLITK_PIX moves them to the split-K family too.
  const int splitk_pix = clc_tuning[CLC_TUNE_SPLITK_PIX] > 0 ? clc_tuning[CLC_TUNE_SPLITK_PIX] : 256;
  const bool small_map = img_pix <= 256 || (img_pix <= splitk_pix && C <= clc_tuning[CLC_TUNE_SPLITK_MAXC]);
  if (small_map) {
    if (heavy128_split(p, classes) > 1) {
      use_split(p, d, classes);
      if (p.ksplit > 1) return launch<128, 128, 4, 2>(p, classes, st);
    }
    // Heavy data gradients of the slice loop (3x3, 224 -> 448..768 on the stacked batch): enough 64x64 tiles to fill the
    // chip without splitting K, and the LDS-tiled kernel shares each operand tile among 4 waves where the split-K family
    // re-fetches fragments per wave (224 -> 704 @ 4096 rows: 192 -> 130 us).  Data gradients exist in training only, so the
    // codec path's batch-invariant family rule is untouched.  (The forward layers of the same nets measured slower this way.)
    static const int heavy_dgrad = getenv("CLC_HEAVY_DGRAD") ? atoi(getenv("CLC_HEAVY_DGRAD")) : 1;   // 0: A/B knob
    static const int heavy_min = getenv("CLC_HEAVY_MIN") ? atoi(getenv("CLC_HEAVY_MIN")) : 60;   // K-tiles (224-channel 3x3: 63); the 36-tile ones measured faster on split-K in the step
    if (heavy_dgrad && img_pix <= 256 && d->transposed && d->ks * d->ks * p.kc_tiles >= heavy_min && C >= 128 &&
        (long)((p.M + 63) / 64) * ((C + 63) / 64) * classes >= 128) {
      use_split(p, d, classes);
      return launch<64, 64, 2, 2>(p, classes, st);
    }
    // 32x32 tiles: these layers are bound by how many CUs get MFMA work (f32 MFMA = 64 cycles each), not by operand
    // re-use, so the smaller tile (2x the workgroups of 32x64) wins on every 16x16 shape measured
    // ... except the slice-parameter nets with multi-MB filters (448..704 -> 224), where halving the number of N tiles
    // halves the filter re-reads that dominate them
    static const int sk_rule = getenv("CLC_SPLITK_RULE") ? atoi(getenv("CLC_SPLITK_RULE")) : 1;   // 0: old rule (A/B knob)
    if (sk_rule == 0) {
      if ((long)d->Cout * d->ks * d->ks * d->Cin > 200000 && C > 32) return launch_splitk<64>(p, classes, st);
      return launch_splitk<32>(p, classes, st);
    }
    // 32x32 tiles while they fit the chip in one round (256 workgroups), 32x64 beyond that.  Both variants give every
    // output element the same K order (same wave / K-tile assignment, same combine tree), so the choice may look at M.
    const long wg32 = (long)((p.M + 31) / 32) * ((C + 31) / 32) * classes;
    if (C > 32 && wg32 > 256) return launch_splitk<64>(p, classes, st);
    return launch_splitk<32>(p, classes, st);
  }
  // Reduced-precision mode (CLC_TUNE_BF16): from here on only — the contract of set_precision("bf16") is "maps larger than 16x16; the
  // entropy-parameter nets, the context model and with it the codec stay f32".  Every small_map branch above (split-K family, the
  // heavy data gradients and the heavy128 forward of the slice-parameter nets, 3x3 layers with few channels on <= 32x32 maps) has
  // returned by now with p.bf16 = 0.
  p.bf16 = clc_tuning[CLC_TUNE_BF16] != 0 && img_pix > 256;
  // 3x3 / stride 1 / 128 input channels with the filter also supplied in fragment order: halo-resident tile, filter streamed from L2 into
  // registers, no barrier in the K loop (conv_halo.hip; same K order and epilogue arithmetic -> same bits)
  if ((clc_tuning[CLC_TUNE_WINO] & (d->transposed ? 2 : 1)) && d->w_wino && d->ks == 3 && d->stride == 1 && classes == 1) {
    const int v = clc_conv_wino_launch(&p, d->w_wino, st);   // Winograd F(2x2, 3x3): conv_wino.hip
    if (v) return v;
  }
  if (clc_tuning[CLC_TUNE_HALO] && d->w_packed && d->ks == 3 && d->stride == 1 && classes == 1) {
    const int v = clc_conv_halo_launch(&p, d->w_packed, st);
    if (v) return v;
  }
  if (img_pix <= 1024) {
    if (C >= 64) {
      // (128 x 64 tiles on 4 waves — a wave owns two accumulator blocks, the step that paid on the 256 x 64 / 128 x 32 tiles — were built
      //  for these 32x32-map layers in round 4 and measured in one process, forward / data gradient: 128 -> 512 92.7 / 99.2 us vs 93.6 /
      //  97.8, 128 -> 128 36.8 / 34.6 vs 36.7 / 34.6, 320 -> 320 172.6 / 169.3 vs 216.0 / 165.7: nothing, or worse.  Removed.)
      use_split(p, d, classes);
      return launch<64, 64, 2, 2>(p, classes, st);
    }
    return launch<64, 32, 2, 1>(p, classes, st);
  }
  // 8 waves per tile (4 per SIMD at 2 workgroups/CU): measured +3..15 % over 4 waves on every large-map shape
  // (1x1 128->128 @128^2: 63 -> 73 TF); the summation order per output element does not depend on the wave grid
  // (CLC_TUNE_1X1_TILE: HBM-bound 1x1 layers with few K-tiles on the narrower tile — 48 KB of LDS, three workgroups per CU)
  // (Round 4 built a WAVE-PRIVATE variant of this — conv_w1x1_kernel: a wave owns 32 pixels and all output channels, filter resident in
  // LDS, operands straight into registers, swapped MFMA roles with 16-B epilogue quads, no barrier — bit-identical, and 1.2-2.1x SLOWER:
  // 128->128 61 -> 103 us, 64->256 + stored derivative 82 -> 177, 64->192 47 -> 92.  Every access of that layout touches 32 B of a 128-B
  // line (1.5-2 TB/s where the row pieces of epilogue_regs / LDS-DMA reach 3.3-4.2).  Removed; the structure pays only where two GEMMs
  // are chained through the registers: csrc/fused_mlp.hip.)
  // (... and with whole-line stores it does pay on the plain layers too: lin_kernel, fused_mlp.hip — the 128 -> 128 / 64 -> 64 layers with a
  //  bias / residual epilogue, forward and data gradient alike: a 1x1 data gradient is the same launch on the transposed filter)
  if (clc_tuning[CLC_TUNE_LIN] && d->ks == 1 && d->stride == 1 && classes == 1 && vec_ok && !p.shuffle && !p.group_rows && !p.xs && p.in_op == CLC_IN_NONE &&
      p.act == CLC_ACT_NONE && p.norm == CLC_NORM_NONE && !p.y_pre && !p.res_gate && !p.out_gate && p.ldw == p.Cin && p.vec_epi && !p.bf16) {
    const int v = clc_lin_launch(p.x, p.ldx, p.w, p.bias, p.res, p.ldr, p.res_scale, p.y, p.ldy, p.M, p.Cin, p.Cout, st);
    if (v) { CLC_LAUNCH_CHECK(); return v; }
  }
  if (d->ks == 1 && C > 32 && vec_ok) {   // persistent pipelined kernel (CLC_TUNE_P1X1), where the layer qualifies
    const int v = launch_p1x1(p, classes, st);
    if (v) return v;
  }
  if (clc_tuning[CLC_TUNE_1X1_TILE] && d->ks == 1 && p.kc_tiles <= clc_tuning[CLC_TUNE_1X1_TILE] && C > 32) return launch<128, 64, 4, 2>(p, classes, st);
  // (256 x 128 tiles — 64 x 64 per wave, one workgroup per CU — were measured again in round 4 on the branch-free K loop: 128 -> 128 @ 8x128x128
  //  306.0 / 299.3 us (forward / data gradient) vs 310.1 / 308.1, 128 -> 512 @ 8x64x64 296.3 / 302.9 vs 300.7 / 303.2; step 27.57 vs 27.74 ms.)
  if (C % 128 == 0 || C >= 384) return launch<128, 128, 4, 2>(p, classes, st);
  // 64-channel 3x3 layers (the ResidualBlocks of the ConvTransBlocks) with enough rows for 512 tiles of 256: each wave then owns a
  // 64 x 32 block (two accumulators, 12 fragment reads and 6 DMA pieces per 32 MFMAs instead of 8 and 3 per 16): 98.1 -> 91.0 us on
  // 64 -> 64 @ 8x128x128, same bits (the K order of an output element does not depend on the tile, so the choice may look at M)
  // (filter sets: the set is chosen per workgroup, so a tile must not straddle two of them — fill_params guarantees multiples of 128 only)
  // (Round 4 also built the wave-private structure of fused_mlp.hip for these layers — conv_wp3x3_kernel: the whole 64 x 64 x 9 filter (144 KB) resident
  //  in LDS, a wave owns 32 pixels, the nine shifted source pixels loaded straight into registers three taps ahead, no barrier — bit-identical and
  //  NOT faster: 96.7 vs 95.6 us at 8x128x128 (graph-replayed), step 27.15 vs 27.06 ms.  Ablated: 85.5 us without operand traffic, 47.6 without
  //  MFMAs, 27.4 with neither — filter deposit + fragment reads + stores alone, amortised over only two tiles per wave (4096 tiles, 2048 waves);
  //  the MFMA floor is 61 us.  Removed.)
  if (clc_tuning[CLC_TUNE_TILE256] && d->ks == 3 && C > 32 && C <= 64 && p.M >= 256 * 512 && (!p.group_rows || p.group_rows % 256 == 0))
    return launch<256, 64, 4, 2>(p, classes, st);
  if (C > 32) return launch<128, 64, 4, 2>(p, classes, st);
  if (C <= 16 && vec_ok) {   // 16-column MFMAs for the 12-channel tail (a rule on the layer's shape alone: its K order differs from the 32-column kernels')
    const int v = launch_n16(p, classes, st);
    if (v) return v;
  }
  // the 12-channel tail of the synthesis transform (and any <= 32-channel 3x3 layer on >= 65 536 rows): 128 x 32 tiles, a wave owns 64 x 32
  // (two accumulators per B fragment) — 149.5 -> 120.3 us on 128 -> 12 @ 8x128x128, same bits (the same reasoning as the 256 x 64 rule)
  if (clc_tuning[CLC_TUNE_TILE256] && d->ks == 3 && p.M % 128 == 0 && p.M >= 128 * 512) return launch<128, 32, 2, 1>(p, classes, st);
  return launch<64, 32, 2, 1>(p, classes, st);
}

extern "C" int clc_filter_transpose(const float* w, float* wt, int Cout, int T, int Cin, clc_stream_t stream) {
  CLC_CHECK(w && wt && Cout > 0 && T > 0 && Cin > 0, "clc_filter_transpose: bad args");
  dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, T);
  hipLaunchKernelGGL(filter_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, Cout, T, Cin);
  CLC_LAUNCH_CHECK();
  return 0;
}
