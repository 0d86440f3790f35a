// conv_wgrad.hip — filter gradient of the NHWC fp32 convolutions on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces cuDNN backward-weight behind loss.backward() (/root/reference/train_CLC.py:159) for
// every conv / linear of the path.
//
// GEMM view:  dW[co][tap][ci] = sum_pix dY[pix][co] * in_op(X[pix@tap][ci])
//   M = co, N = ci (one filter tap per workgroup column), K = output pixels.
// Both operands are K-major in memory (pixel rows, channel-contiguous), so the LDS images are
// [k][M] and [k][N]: 16-B coalesced loads -> ds_write_b128, fragments by conflict-free
// ds_read_b32 (lane i reads column i of row k).  K is split over blockIdx.z; every split writes
// its own partial slab and a second kernel sums the slabs in a fixed order -> bitwise
// reproducible (no float atomics; train_CLC.py:28-29 asks for deterministic kernels).
// The bias gradient (column sums of dY) rides along in the workgroups of the first column.
#include "common.h"

#include <type_traits>

namespace {

constexpr int BK = 32;

constexpr unsigned kOOB = 0x80000000u;  // >= num_records of every SRD -> buffer load returns 0 (no branch around loads)

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0));
}

// A kernel-argument scalar behind an opaque move.  The problem descriptors of a grouped launch are indexed dynamically, and the
// register allocator treats their (invariant) s_loads as free to re-issue: under SGPR pressure it reloaded H, W, ldx ... from
// memory in front of EVERY buffer load of the K loop, each with an s_waitcnt lgkmcnt(0) (ISA of the r2 stream-K kernels: 16
// scalar loads per K-tile).  A pinned value lives in an SGPR (or spills to a VGPR lane) instead.
__device__ __forceinline__ int pin_s(int x) {
  int r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "s"(x));
  return r;
}

// LDS-DMA: 64 lanes x 16 B from per-lane byte offsets of the buffer straight into LDS at dst (wave-uniform) + 16 * lane
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc cannot type-check the LDS address-space cast)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, byte_off, 0, 0, 0);
#endif
}

struct WgradParams {
  const float* x; const float* dy; float* partial; float* bias_partial;
  unsigned x_bytes, dy_bytes;
  const float* dys; int lddys, dys_act, dys_pre; unsigned dys_bytes;   // fused activation backward on dy
  int ow_shift, img_shift;   // log2(OW), log2(OH*OW) when both are powers of two (else -1): pixel decode without integer division
  int rmw;          // single-split direct mode with accumulation: slab[i] += acc (one writer per element -> deterministic)
  int N, H, W, Cin, ldx;
  int OH, OW, Cout, lddy;
  int ks, stride, pad, in_op;
  int K;            // total pixels N*OH*OW
  int k_per_split;  // multiple of BK
  int nci;          // ci tiles
};

// Where a workgroup's result goes.  mode 0: global [Cout][T][Cin] layout, plain store (a split's slab or the gradient itself);
// mode 1: same layout, accumulate (single writer per element); mode 2: compact tile image (stream-K partial slot).
struct OutSpec { float* w; float* b; int mode; };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF = 2: f32 products on the bf16 matrix cores.  An f32 value is the exact sum of three bf16 pieces (8 + 8 + 8 significand bits: v = h + m + l with
// h = bf16(v), m = bf16(v - h), l = bf16(v - h - m); the differences are exact in f32), so a product is the sum of nine exact piece products; the six
// largest — h h', h m', m h', h l', l h', m m' — are accumulated in f32 by six v_mfma_f32_32x32x16_bf16, the dropped three are <= 2^-23 of the product
// together (an f32 multiply-add's own rounding is 2^-24).  The bf16 matrix rate is 16x the f32 one: 6 / 16 of the MFMA time, plus the splitting.
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hb = (__bf16)v[e];
    const float r = v[e] - (float)hb;
    const __bf16 mb = (__bf16)r;
    const float r2 = r - (float)mb;
    h[e] = hb; m[e] = mb; l[e] = (__bf16)r2;
  }
}
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// (small terms first: they meet the running sum at their own magnitude)
#define MFMA_SPLIT6(ah, am, al, bh, bm, bl, c) do { \
    c = MFMA_BF(al, bh, c); c = MFMA_BF(ah, bl, c); c = MFMA_BF(am, bm, c); \
    c = MFMA_BF(am, bh, c); c = MFMA_BF(ah, bm, c); c = MFMA_BF(ah, bh, c); } while (0)

// BF (with DMA): reduced-precision mode (CLC_TUNE_BF16) — the f32 tiles in LDS are rounded to bf16 at fragment read and 16 k are
// contracted by one v_mfma_f32_32x32x16_bf16 (f32 accumulate): lane half h supplies k = 16 q + 2 j + h as element j for both operands.
template <int BM, int BN, int WM, int WN, bool DMA = false, int BF = 0>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int bx, const int by, const int k_begin, const int k_end, const OutSpec o, const int tid) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_P = (BK * BM / 4 + NT - 1) / NT, B_P = (BK * BN / 4 + NT - 1) / NT;
  constexpr int AQ = BM / 4, BQ = BN / 4;  // float4 pieces per row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BK][BM]
  float* Bs = smem + 2 * BK * BM;   // [2][BK][BN]

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tap = bx / p.nci, ci0 = (bx % p.nci) * BN;
  const int kh = tap / p.ks, kw = tap - kh * p.ks;
  const int co0 = by * BM;
  const int ntiles = (k_end - k_begin + BK - 1) / BK;
  const bool do_bias = (o.b != nullptr) && (bx == 0);
  // everything the K loop reads, pinned (see pin_s)
  const int pCout = pin_s(p.Cout), pCin = pin_s(p.Cin), pH = pin_s(p.H), pW = pin_s(p.W), pOH = pin_s(p.OH), pOW = pin_s(p.OW);
  const int plddy = pin_s(p.lddy), plddys = pin_s(p.lddys), pldx = pin_s(p.ldx), pstride = pin_s(p.stride), ppad = pin_s(p.pad);
  const int pows = pin_s(p.ow_shift), pimgs = pin_s(p.img_shift), pact = pin_s(p.dys_act), ppre = pin_s(p.dys_pre);

  f32x4 a_reg[A_P], b_reg[B_P], bias_acc[A_P], s_reg[A_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) bias_acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const bool sq = p.in_op == CLC_IN_SQUARE;
  const bool fuse_act = p.dys != nullptr;
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fuse_act ? p.dys : p.dy), 0, fuse_act ? p.dys_bytes : p.dy_bytes, 0x00020000);
  auto load_tile = [&](int kt) {
    const int kbase = k_begin + kt * BK;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int piece = tid + i * NT, row = piece / AQ, q = piece - row * AQ;
      const int pix = kbase + row, co = co0 + q * 4;
      const bool ok = row < BK && pix < k_end && co < pCout;
      a_reg[i] = buf_load4(dr, ok ? ((unsigned)pix * (unsigned)plddy + (unsigned)co) * 4u : kOOB);
      if (fuse_act) s_reg[i] = buf_load4(sr, ok ? ((unsigned)pix * (unsigned)plddys + (unsigned)co) * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const int piece = tid + i * NT, row = piece / BQ, q = piece - row * BQ;
      const int pix = kbase + row, ci = ci0 + q * 4;
      const int pp = pix < k_end ? pix : k_begin;
      int n, oy, ox;
      if (pows >= 0) {   // block-uniform: every map of a 2^k-sized image is a power of two
        n = pp >> pimgs;
        const int r = pp & ((1 << pimgs) - 1);
        oy = r >> pows; ox = r & ((1 << pows) - 1);
      } else {
        n = pp / (pOH * pOW);
        const int r = pp - n * (pOH * pOW);
        oy = r / pOW; ox = r - oy * pOW;
      }
      const int iy = oy * pstride - ppad + kh, ix = ox * pstride - ppad + kw;
      const bool ok = row < BK && pix < k_end && ci < pCin && (unsigned)iy < (unsigned)pH && (unsigned)ix < (unsigned)pW;
      b_reg[i] = buf_load4(xr, ok ? ((unsigned)((n * pH + iy) * pW + ix) * (unsigned)pldx + (unsigned)ci) * 4u : kOOB);
    }
  };
  // Operands that need no arithmetic on the way in (no fused activation derivative on dy, no squared input) go from global memory
  // STRAIGHT to LDS (LDS-DMA): piece p of a tile lands at float offset 4 p of the tile image, which is exactly where 64
  // consecutive lanes x 16 B of one DMA instruction go — no staging registers, no ds_write, and the next tile's pieces are in flight
  // from the top of the iteration.  Same LDS image, same MFMA order: the results do not change.
  // (DMA: the host put only such problems into this launch — dma_ok())
  static_assert(!DMA || ((BK * BM / 4) % NT == 0 && (BK * BN / 4) % NT == 0), "whole DMA instructions per tile");
  constexpr bool use_dma = DMA;
  const int wave_base = (tid >> 6) * 64;   // (first piece of this wave in a group of NT pieces)
  auto dma_tile = [&](int kt, int buf) {
    const int kbase = k_begin + kt * BK;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int piece = tid + i * NT, row = piece / AQ, q = piece - row * AQ;
      const int pix = kbase + row, co = co0 + q * 4;
      const bool ok = pix < k_end && co < pCout;
      dma16(dr, As + buf * BK * BM + (i * NT + wave_base) * 4, ok ? ((unsigned)pix * (unsigned)plddy + (unsigned)co) * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const int piece = tid + i * NT, row = piece / BQ, q = piece - row * BQ;
      const int pix = kbase + row, ci = ci0 + q * 4;
      const int pp = pix < k_end ? pix : k_begin;
      int n, oy, ox;
      if (pows >= 0) {
        n = pp >> pimgs;
        const int r = pp & ((1 << pimgs) - 1);
        oy = r >> pows; ox = r & ((1 << pows) - 1);
      } else {
        n = pp / (pOH * pOW);
        const int r = pp - n * (pOH * pOW);
        oy = r / pOW; ox = r - oy * pOW;
      }
      const int iy = oy * pstride - ppad + kh, ix = ox * pstride - ppad + kw;
      const bool ok = pix < k_end && ci < pCin && (unsigned)iy < (unsigned)pH && (unsigned)ix < (unsigned)pW;
      dma16(xr, Bs + buf * BK * BN + (i * NT + wave_base) * 4, ok ? ((unsigned)((n * pH + iy) * pW + ix) * (unsigned)pldx + (unsigned)ci) * 4u : kOOB);
    }
  };
  // LIN (stride 1, 'same' padding, power-of-two maps — every DMA-staged problem of the training step): the source pixel of output pixel
  // `pix` at this workgroup's tap is pix + (kh - pad) * W + (kw - pad), i.e. the operand address is LINEAR in pix.  A piece's part of it
  // is computed once; per K-tile it costs one add, and the border test two shifts / masks and two compares — no v_mul_lo_u32 (a
  // quarter-rate instruction) and no division in the loop.  f32 MFMAs and VALU work do not overlap on gfx950: on the 64 x 64 tile the
  // general form above spent ~400 VALU cycles per K-tile next to 1 024 MFMA cycles (conv_wgrad_sk_kernel<64, 64, 2, 2, true>: 43 %
  // MFMA-busy in round 3).
  int a_rel[A_P], b_rel[B_P], b_row[B_P];
  bool a_okc[A_P], b_okc[B_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int piece = tid + i * NT, row = piece / AQ, q = piece - row * AQ;
    a_rel[i] = (row * plddy + co0 + q * 4) * 4;
    a_okc[i] = co0 + q * 4 < pCout;
  }
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    const int piece = tid + i * NT, row = piece / BQ, q = piece - row * BQ;
    b_row[i] = row;
    b_rel[i] = ((row + (kh - ppad) * pW + (kw - ppad)) * pldx + ci0 + q * 4) * 4;
    b_okc[i] = ci0 + q * 4 < pCin;
  }
  auto dma_tile_lin = [&](int kt, int buf, bool en) {   // en = false: deposits zeros (no block-uniform branch in the K loop)
    const int kbase = k_begin + kt * BK;
    const int kleft = en ? k_end - kbase : 0;                       // rows of this tile that exist (scalar)
    const int a_org = kbase * plddy * 4, b_org = kbase * pldx * 4;   // scalars
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int row = (tid + i * NT) / AQ;
      dma16(dr, As + buf * BK * BM + (i * NT + wave_base) * 4, (row < kleft && a_okc[i]) ? (unsigned)(a_org + a_rel[i]) : kOOB);
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const int pix = kbase + b_row[i];
      const int oy = (pix >> pows) & (pOH - 1), ox = pix & (pOW - 1);
      const bool ok = b_row[i] < kleft && b_okc[i] && (unsigned)(oy + kh - ppad) < (unsigned)pH && (unsigned)(ox + kw - ppad) < (unsigned)pW;
      dma16(xr, Bs + buf * BK * BN + (i * NT + wave_base) * 4, ok ? (unsigned)(b_org + b_rel[i]) : kOOB);
    }
  };
  auto bias_from_lds = [&](int buf) {   // (the DMA path has no staging registers to sum: the dY tile is read back, 4 floats per piece)
#pragma unroll
    for (int i = 0; i < A_P; ++i) bias_acc[i] += *reinterpret_cast<const f32x4*>(As + buf * BK * BM + (tid + i * NT) * 4);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int piece = tid + i * NT, row = piece / AQ, q = piece - row * AQ;
      if (fuse_act) a_reg[i] = a_reg[i] * act_deriv4(s_reg[i], pact, ppre);
      if (row < BK) *reinterpret_cast<f32x4*>(As + (buf * BK + row) * BM + q * 4) = a_reg[i];
      if (do_bias) bias_acc[i] += a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      const int piece = tid + i * NT, row = piece / BQ, q = piece - row * BQ;
      if (row < BK) *reinterpret_cast<f32x4*>(Bs + (buf * BK + row) * BN + q * 4) = sq ? b_reg[i] * b_reg[i] : b_reg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool lin = use_dma && pows >= 0 && pstride == 1 && pH == pOH && pW == pOW;   // block-uniform: chosen ONCE, outside the K loop
  const int khalf = lane >> 5, li = lane & 31;
  auto kloop = [&](auto LIN) {
    constexpr bool kLin = decltype(LIN)::value;
    if (ntiles > 0) {
      if (kLin) {
        dma_tile_lin(0, 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (use_dma) {
        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        load_tile(0);
        store_tile(0);
      }
      __syncthreads();
    }
    for (int kt = 0; kt < ntiles; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < ntiles;
      if (kLin) dma_tile_lin(kt + 1, buf ^ 1, more);
      else if (more) { if (use_dma) dma_tile(kt + 1, buf ^ 1); else load_tile(kt + 1); }
      if (use_dma && do_bias) bias_from_lds(buf);
      const float* Ab = As + (buf * BK + khalf) * BM + wm * (BM / WM) + li;
      const float* Bb = Bs + (buf * BK + khalf) * BN + wn * (BN / WN) + li;
      if constexpr (BF == 2) {
#pragma unroll
        for (int q = 0; q < BK / 16; ++q) {
          bf16x8 pa[TM][3], pb[TN][3];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Ab[(16 * q + 2 * e) * BM + i * 32];
            split3(v, pa[i][0], pa[i][1], pa[i][2]);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Bb[(16 * q + 2 * e) * BN + j * 32];
            split3(v, pb[j][0], pb[j][1], pb[j][2]);
          }
#define CLC_TERM(ia, ib) _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MFMA_BF(pa[i][ia], pb[j][ib], acc[i][j]);
          CLC_TERM(2, 0) CLC_TERM(0, 2) CLC_TERM(1, 1) CLC_TERM(1, 0) CLC_TERM(0, 1) CLC_TERM(0, 0)   // term-major, small terms first
#undef CLC_TERM
        }
      } else if constexpr (BF == 1) {
#pragma unroll
        for (int q = 0; q < BK / 16; ++q) {
          bf16x8 pa[TM], pb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) pa[i][e] = (__bf16)Ab[(16 * q + 2 * e) * BM + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) pb[j][e] = (__bf16)Bb[(16 * q + 2 * e) * BN + j * 32];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i], pb[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int ss = 0; ss < BK / 2; ++ss) {
          float af[TM], bf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = Ab[(2 * ss) * BM + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = Bb[(2 * ss) * BN + j * 32];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
      if (use_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's pieces have landed
      else if (more) store_tile(buf ^ 1);
      __syncthreads();
    }
  };
  if (lin) kloop(std::true_type{}); else kloop(std::false_type{});

  // result: [Cout][T][Cin] (slab / gradient) or the compact [BM][BN] image of a stream-K partial
  const int T = p.ks * p.ks;
  const int rhalf = 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cil = wn * (BN / WN) + j * 32 + li, ci = ci0 + cil;
    if (o.mode != 2 && ci >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf, co = co0 + col;
        if (o.mode == 2) {
          o.w[col * BN + cil] = acc[i][j][r];
        } else if (co < p.Cout) {
          float* dst = o.w + ((size_t)co * T + tap) * p.Cin + ci;
          *dst = o.mode == 1 ? *dst + acc[i][j][r] : acc[i][j][r];
        }
      }
  }

  if (do_bias) {  // reduce the per-thread column sums over the BK rows through LDS (fixed order)
    __syncthreads();
    float* red = smem;  // [BK][BM]
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int piece = tid + i * NT, row = piece / AQ, q = piece - row * AQ;
      if (row < BK) *reinterpret_cast<f32x4*>(red + row * BM + q * 4) = bias_acc[i];
    }
    __syncthreads();
    for (int c = tid; c < BM; c += NT) {
      float s = 0.f;
      for (int r = 0; r < BK; ++r) s += red[r * BM + c];
      if (o.mode == 2) {
        o.b[c] = s;
      } else if (co0 + c < p.Cout) {
        float* dst = o.b + co0 + c;
        *dst = o.mode == 1 ? *dst + s : s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 on power-of-two maps: ALL NINE TAPS in one workgroup.
// The tap-per-workgroup kernel above re-reads dY and X nine times (PMC: 12.6 GB fetched per step by the 64-channel
// layers alone, HBM-bound at 58 TF).  Here a K-tile is a TH x TW block of output pixels (TH*TW = 32); its dY rows and
// the (TH+2) x (TW+2) input window are staged in LDS ONCE and the nine taps are nine shifted views of that window:
// 64 co x (9 taps x 64 ci) outputs per workgroup, wave (wm, wn) = 32 co x 32 ci x 9 taps = 9 MFMA blocks (144 acc
// registers), 10 ds_read_b32 per 9 MFMAs.  Same slab layout / fixed-order reduce as above.
template <int TW, bool DMA = false, int BF = 0>
__device__ __forceinline__ void wgrad_taps_body(const WgradParams& p, const int bx, const int by, const int t_begin, const int t_end, const OutSpec o, const int tid) {
  constexpr int TH = 32 / TW, XW = TW + 2, XH = TH + 2, XP = XH * XW;
  constexpr int LTW = TW == 32 ? 5 : (TW == 16 ? 4 : 3);
  constexpr int X_P = (XP * 16 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][32][64]   dY tile, [k][co]
  float* Xs = smem + 2 * 32 * 64;    // [2][XP][64]   input window, [pixel][ci]

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ci0 = bx * 64, co0 = by * 64;
  const int ntiles = t_end - t_begin;
  const bool do_bias = (o.b != nullptr) && (bx == 0);
  const int lcols = pin_s(p.ow_shift - LTW);                       // log2(tile columns per image row)
  const int lrows = pin_s((p.img_shift - p.ow_shift) - (5 - LTW)); // log2(tile rows per image)
  // everything the K loop reads, pinned (see pin_s)
  const int pCout = pin_s(p.Cout), pCin = pin_s(p.Cin), pH = pin_s(p.H), pW = pin_s(p.W), pOH = pin_s(p.OH), pOW = pin_s(p.OW);
  const int plddy = pin_s(p.lddy), plddys = pin_s(p.lddys), pldx = pin_s(p.ldx), pact = pin_s(p.dys_act), ppre = pin_s(p.dys_pre);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const bool fuse_act = p.dys != nullptr;
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fuse_act ? p.dys : p.dy), 0, fuse_act ? p.dys_bytes : p.dy_bytes, 0x00020000);

  f32x4 a_reg[2], s_reg[2], x_reg[X_P], bias_acc[2];
  bias_acc[0] = bias_acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_tile = [&](int t) {
    const int bc = t & ((1 << lcols) - 1), t2 = t >> lcols;
    const int br = t2 & ((1 << lrows) - 1), n = t2 >> lrows;
    const int oy0 = br * TH, ox0 = bc * TW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = tid + i * 256, row = piece >> 4, q = piece & 15;
      const int pix = (n * pOH + oy0 + (row >> LTW)) * pOW + ox0 + (row & (TW - 1));
      const int co = co0 + q * 4;
      const bool ok = co < pCout;
      a_reg[i] = buf_load4(dr, ok ? ((unsigned)pix * (unsigned)plddy + (unsigned)co) * 4u : kOOB);
      if (fuse_act) s_reg[i] = buf_load4(sr, ok ? ((unsigned)pix * (unsigned)plddys + (unsigned)co) * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < X_P; ++i) {
      const int piece = tid + i * 256, xp = piece >> 4, q = piece & 15;
      const int wy = xp / XW, wx = xp - wy * XW;
      const int iy = oy0 + wy - 1, ix = ox0 + wx - 1, ci = ci0 + q * 4;
      const bool ok = xp < XP && ci < pCin && (unsigned)iy < (unsigned)pH && (unsigned)ix < (unsigned)pW;
      x_reg[i] = buf_load4(xr, ok ? ((unsigned)((n * pH + iy) * pW + ix) * (unsigned)pldx + (unsigned)ci) * 4u : kOOB);
    }
  };
  // DMA (host: no fused activation derivative on dy): the dY rows and the input window go from global memory straight to LDS —
  // piece p lands at float offset 4 p of its tile image, where 64 consecutive lanes x 16 B of one DMA instruction go (the last
  // window instruction is partly masked) — which frees the 44 staging registers for a one-step-ahead fragment prefetch below.
  const int wave_base = (tid >> 6) * 64;
  // f32 MFMAs and VALU instructions do not overlap on gfx950, so the per-piece address arithmetic of the staging (9 pieces with three
  // v_mul_lo_u32 and a division each: ~700 VALU cycles per K-tile next to 9 216 MFMA cycles) came straight out of the MFMA rate.  A
  // piece's position inside the tile never changes: its element offset relative to the tile origin and its border flags are computed
  // ONCE; per K-tile the origin and the tile's border mask are scalars, and a piece costs an add, an and, a compare and a select.
  // (Tiles are aligned blocks of power-of-two maps: a window pixel leaves the image only on the side where its tile touches the border.)
  int a_rel[2], x_rel[X_P], x_flag[X_P];   // x_flag: bit 0-3 = the piece is the window's top / bottom / left / right halo, bit 4 = never fetched
  bool a_okc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = tid + i * 256, row = piece >> 4, q = piece & 15;
    a_rel[i] = ((row >> LTW) * pOW + (row & (TW - 1))) * plddy + co0 + q * 4;
    a_okc[i] = co0 + q * 4 < pCout;
  }
#pragma unroll
  for (int i = 0; i < X_P; ++i) {
    const int piece = tid + i * 256, xp = piece >> 4, q = piece & 15;
    const int wy = xp / XW, wx = xp - wy * XW, ci = ci0 + q * 4;
    x_rel[i] = ((wy - 1) * pW + (wx - 1)) * pldx + ci;
    x_flag[i] = (wy == 0 ? 1 : 0) | (wy == XH - 1 ? 2 : 0) | (wx == 0 ? 4 : 0) | (wx == XW - 1 ? 8 : 0) | ((piece < XP * 16 && ci < pCin) ? 0 : 16) | 32;   // (bit 5: always set, masked in only by a disabled tile)
  }
  auto dma_tile = [&](int t, int buf, bool en) {   // en = false: deposits zeros (the K loop stays free of a block-uniform branch)
    const int bc = t & ((1 << lcols) - 1), t2 = t >> lcols;
    const int br = t2 & ((1 << lrows) - 1), n = t2 >> lrows;
    const int oy0 = br * TH, ox0 = bc * TW;
    const int a_org = ((n * pOH + oy0) * pOW + ox0) * plddy, x_org = ((n * pH + oy0) * pW + ox0) * pldx;   // scalars
    const int tmask = (en ? 0 : 63) | (oy0 == 0 ? 1 : 0) | (oy0 + TH == pH ? 2 : 0) | (ox0 == 0 ? 4 : 0) | (ox0 + TW == pW ? 8 : 0) | 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      dma16(dr, As + buf * 32 * 64 + (i * 256 + wave_base) * 4, (en && a_okc[i]) ? (unsigned)(a_org + a_rel[i]) * 4u : kOOB);
#pragma unroll
    for (int i = 0; i < X_P; ++i) {
      if ((i + 1) * 256 <= XP * 16 || tid + i * 256 < XP * 16)   // (only the last instruction is partly masked: compile-time for the others)
        dma16(xr, Xs + buf * XP * 64 + (i * 256 + wave_base) * 4, (x_flag[i] & tmask) == 0 ? (unsigned)(x_org + x_rel[i]) * 4u : kOOB);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = tid + i * 256;
      if (fuse_act) a_reg[i] = a_reg[i] * act_deriv4(s_reg[i], pact, ppre);
      *reinterpret_cast<f32x4*>(As + buf * 32 * 64 + piece * 4) = a_reg[i];
      if (do_bias) bias_acc[i] += a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < X_P; ++i) {
      const int piece = tid + i * 256;
      if (piece < XP * 16) *reinterpret_cast<f32x4*>(Xs + buf * XP * 64 + piece * 4) = x_reg[i];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (ntiles > 0) {
    if (DMA) {
      dma_tile(t_begin, 0, true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      load_tile(t_begin);
      store_tile(0);
    }
    __syncthreads();
  }
  const int khalf = lane >> 5, li = lane & 31;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < ntiles;
    if (DMA) dma_tile(t_begin + kt + 1, buf ^ 1, more);
    else if (more) load_tile(t_begin + kt + 1);
    const float* Ab = As + buf * 32 * 64 + wm * 32 + li;
    const float* Xb = Xs + buf * XP * 64 + wn * 32 + li;
    if (DMA && BF == 2) {   // f32 products from bf16 pieces (split3), split in registers: 16 k per six v_mfma_f32_32x32x16_bf16.
      // Lane half h supplies k = 16 q + 8 h + j as element j (any k <-> slot bijection serves, as long as both operands use it): a lane's 8 k are
      // 8 CONSECUTIVE pixels of one row, so the three taps of a filter row need 10 consecutive window columns — 10 values read and split per
      // (q, kh) instead of 24, the fragments of kw = 0 / 1 / 2 being three shifted views of them.  Software-pipelined: the values of step
      // (q, kh) + 1 are read and split while the 18 MFMAs of step (q, kh) run (VALU issues between the 32-cycle MFMAs of the same wave).
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) bias_acc[i] += *reinterpret_cast<const f32x4*>(As + buf * 32 * 64 + (tid + i * 256) * 4);
      }
      struct Row { bf16x8 x[3][3]; };   // [kw][piece]
      auto split_a = [&](int q, bf16x8 (&pa)[3]) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = Ab[(16 * q + 8 * khalf + e) * 64];
        split3(v, pa[0], pa[1], pa[2]);
      };
      auto split_row = [&](int q, int kh, Row& r) {
        const int rowq = TW == 32 ? 0 : (TW == 16 ? q : 2 * q + khalf);
        const int colq = TW == 32 ? 16 * q + 8 * khalf : (TW == 16 ? 8 * khalf : 0);
        const float* xr = Xb + ((rowq + kh) * XW + colq) * 64;
        __bf16 ph[10], pm[10], pl[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) {   // truncating pieces (masks and subtractions only): v = h + m + l exactly
          const float v = xr[c * 64];
          const unsigned hb = __builtin_bit_cast(unsigned, v) & 0xFFFF0000u;
          const float r1 = v - __builtin_bit_cast(float, hb);
          const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
          const float r2 = r1 - __builtin_bit_cast(float, mb);
          ph[c] = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
          pm[c] = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
          pl[c] = __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16));
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int e = 0; e < 8; ++e) { r.x[kw][0][e] = ph[e + kw]; r.x[kw][1][e] = pm[e + kw]; r.x[kw][2][e] = pl[e + kw]; }
      };
      bf16x8 pa[2][3];
      Row rows[2];
      split_a(0, pa[0]);
      split_row(0, 0, rows[0]);
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        const int q = st / 3, kh = st - 3 * q, cur = st & 1;
        if (st + 1 < 6) {                                  // the next step's operands: independent of this step's MFMAs
          const int qn = (st + 1) / 3, khn = (st + 1) - 3 * qn;
          if (khn == 0) split_a(qn, pa[qn]);
          split_row(qn, khn, rows[cur ^ 1]);
        }
        const bf16x8 (&a)[3] = pa[q];
        const Row& r = rows[cur];
        // term-major over the three taps of the row: consecutive MFMAs go to different accumulators (small terms first)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[2], r.x[kw][0], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[0], r.x[kw][2], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[1], r.x[kw][1], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[1], r.x[kw][0], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[0], r.x[kw][1], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = MFMA_BF(a[0], r.x[kw][0], acc[kh * 3 + kw]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's pieces have landed
    } else if (DMA && BF == 1) {   // reduced-precision mode: 16 k per v_mfma_f32_32x32x16_bf16, lane half h supplies k = 16 q + 2 j + h as element j
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) bias_acc[i] += *reinterpret_cast<const f32x4*>(As + buf * 32 * 64 + (tid + i * 256) * 4);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 pa;
        const float* xk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * q + 2 * e + khalf;
          pa[e] = (__bf16)Ab[k * 64];
          xk[e] = Xb + ((k >> LTW) * XW + (k & (TW - 1))) * 64;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          bf16x8 px[3];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int e = 0; e < 8; ++e) px[kw][e] = (__bf16)xk[e][(kh * XW + kw) * 64];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, px[kw], acc[kh * 3 + kw], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's pieces have landed
    } else if (DMA) {
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) bias_acc[i] += *reinterpret_cast<const f32x4*>(As + buf * 32 * 64 + (tid + i * 256) * 4);
      }
      // the 10 operand values of k-step ss + 1 are read while the 9 MFMAs of step ss issue
      float a[2], xv[2][9];
      auto read_step = [&](int ss, float& av, float (&x9)[9]) {
        const int k = 2 * ss + khalf;
        av = Ab[k * 64];
        const float* xk = Xb + ((k >> LTW) * XW + (k & (TW - 1))) * 64;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) x9[kh * 3 + kw] = xk[(kh * XW + kw) * 64];
      };
      read_step(0, a[0], xv[0]);
#pragma unroll
      for (int ss = 0; ss < 16; ++ss) {
        const int cur = ss & 1;
        if (ss < 15) read_step(ss + 1, a[cur ^ 1], xv[cur ^ 1]);
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) acc[t9] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], xv[cur][t9], acc[t9], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's pieces have landed
    } else {
#pragma unroll
      for (int ss = 0; ss < 16; ++ss) {
        const int k = 2 * ss + khalf;
        const float a = Ab[k * 64];
        const float* xk = Xb + ((k >> LTW) * XW + (k & (TW - 1))) * 64;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xk[(kh * XW + kw) * 64], acc[kh * 3 + kw], 0, 0, 0);
      }
      if (more) store_tile(buf ^ 1);
    }
    __syncthreads();
  }

  // result: [Cout][9][Cin] (slab / gradient) or the compact [64][9][64] image of a stream-K partial
  const int cil = wn * 32 + li, ci = ci0 + cil;
  if (o.mode == 2 || ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf, co = co0 + col;
        if (o.mode == 2) {
          o.w[(col * 9 + t) * 64 + cil] = acc[t][r];
        } else if (co < p.Cout) {
          float* dst = o.w + ((size_t)co * 9 + t) * p.Cin + ci;
          *dst = o.mode == 1 ? *dst + acc[t][r] : acc[t][r];
        }
      }
  }

  if (do_bias) {  // column sums of the dY tiles: per-thread partial sums over this thread's rows, reduced through LDS
    __syncthreads();
    float* red = smem;  // [32][64]
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(red + (tid + i * 256) * 4) = bias_acc[i];
    __syncthreads();
    if (tid < 64) {
      float sum = 0.f;
      for (int r = 0; r < 32; ++r) sum += red[r * 64 + tid];
      if (o.mode == 2) {
        o.b[tid] = sum;
      } else if (co0 + tid < p.Cout) {
        float* dst = o.b + co0 + tid;
        *dst = o.mode == 1 ? *dst + sum : sum;
      }
    }
  }
}

// split-K form: split z of a problem covers K-range z * k_per_split .. and writes slab z (or, single split, the gradient itself)
__device__ __forceinline__ OutSpec split_out(const WgradParams& p, int split) {
  OutSpec o;
  o.w = p.partial + (size_t)split * p.Cout * p.ks * p.ks * p.Cin;
  o.b = p.bias_partial ? p.bias_partial + (size_t)split * p.Cout : nullptr;
  o.mode = p.rmw ? 1 : 0;
  return o;
}
__device__ __forceinline__ int taps_range_begin(const WgradParams& p, int split) { return split * (p.k_per_split >> 5); }
__device__ __forceinline__ int taps_range_end(const WgradParams& p, int split) { return min(p.K >> 5, (split + 1) * (p.k_per_split >> 5)); }

template <int TW>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_taps_kernel(const WgradParams p) {
  wgrad_taps_body<TW>(p, blockIdx.x, blockIdx.y, taps_range_begin(p, blockIdx.z), taps_range_end(p, blockIdx.z), split_out(p, blockIdx.z), threadIdx.x);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2)
void conv_wgrad_kernel(const WgradParams p) {
  const int kb = blockIdx.z * p.k_per_split;
  wgrad_body<BM, BN, WM, WN>(p, blockIdx.x, blockIdx.y, kb, min(p.K, kb + p.k_per_split), split_out(p, blockIdx.z), threadIdx.x);
}

// Grouped launch: up to kMaxGroup independent filter-gradient problems of one tile shape in ONE grid (descriptors in the
// kernel-argument segment, workgroup -> problem by a scalar search of the prefix table).  The small-map layers of the
// slice loop are 5-20 us kernels of 60-250 workgroups each; grouped they fill the chip and pay one launch.
constexpr int kMaxGroup = 64;
struct WgradGroup {
  int count;
  int wg_end[kMaxGroup];   // exclusive prefix of workgroups
  int gx[kMaxGroup], gy[kMaxGroup];
  WgradParams p[kMaxGroup];
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2)
void conv_wgrad_grouped_kernel(const WgradGroup g) {
  const int b = blockIdx.x;
  int idx = 0;
  while (idx + 1 < g.count && b >= g.wg_end[idx]) ++idx;
  const int l = b - (idx ? g.wg_end[idx - 1] : 0);
  const int gx = g.gx[idx], gy = g.gy[idx];
  const int bx = l % gx, t = l / gx;
  const WgradParams& p = g.p[idx];
  const int kb = (t / gy) * p.k_per_split;
  wgrad_body<BM, BN, WM, WN>(p, bx, t % gy, kb, min(p.K, kb + p.k_per_split), split_out(p, t / gy), threadIdx.x);
}

template <int TW>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_taps_grouped_kernel(const WgradGroup g) {
  const int b = blockIdx.x;
  int idx = 0;
  while (idx + 1 < g.count && b >= g.wg_end[idx]) ++idx;
  const int l = b - (idx ? g.wg_end[idx - 1] : 0);
  const int gx = g.gx[idx], gy = g.gy[idx];
  const int bx = l % gx, t = l / gx;
  const WgradParams& p = g.p[idx];
  wgrad_taps_body<TW>(p, bx, t % gy, taps_range_begin(p, t / gy), taps_range_end(p, t / gy), split_out(p, t / gy), threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Stream-K form of the grouped launch.  The split-K form above sizes every problem as if it had the chip to itself (~256-512
// workgroups each), so a 64-problem group writes and re-reads thousands of partial slabs (8.5 GB per training step, r1 PMC
// profile).  Here ONE grid of G <= 512 workgroups (two per CU) covers the group: the K-units (32 output pixels) of all
// (problem, tile) pairs are laid end to end and workgroup w takes the contiguous range [w*U/G, (w+1)*U/G).  It keeps
// accumulating in registers while it stays inside a tile and writes
//   * straight into the gradient when it covers a tile's whole K range,
//   * else a compact partial image into slot 0 (it CONTINUES a tile begun by workgroup w-1) or slot 1 (it BEGINS a tile that
//     workgroup w+1 continues) of its own slot pair -> at most two partials per workgroup, ~G x |tile| bytes per launch.
// wgrad_sk_fixup_kernel then adds the partials of every split tile in workgroup order (fixed order -> run-to-run reproducible;
// the ranges depend only on the group's shapes, never on timing).
struct SKGroup {
  int count, G, slot_floats;
  long total;                    // K-units of the whole group
  long unit_end[kMaxGroup];      // exclusive prefix of K-units per problem
  int T[kMaxGroup];              // K-units per tile
  int gx[kMaxGroup], gy[kMaxGroup];
  float* slots;                  // [G][2][slot_floats]
  int* plan;                     // [G][4]: {problem, tile, first workgroup of the tile, 1} when workgroup w contributes a tile's LAST partial, else zeros
  WgradParams p[kMaxGroup];      // partial = the gradient itself, bias_partial = the bias gradient (or null), rmw = accumulate
};

struct SKSeg { int idx, tile, k0, k1; };
// threadIdx.x behind an opaque move: the per-thread index arithmetic of a tile body then cannot be hoisted out of the segment
// loop, where it would stay live in VGPRs across the whole K loop of every segment (measured: +50..120 VGPRs, spills)
__device__ __forceinline__ int sk_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ int sk_find(const SKGroup& g, long u, int lo) {   // smallest idx >= lo with u < unit_end[idx]
  int hi = g.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (u >= g.unit_end[mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ SKSeg sk_locate(const SKGroup& g, long u, long end, int& idx) {
  idx = sk_find(g, u, idx);
  const long base = idx ? g.unit_end[idx - 1] : 0;
  const int T = g.T[idx];
  const long local = u - base;
  SKSeg sgm;
  sgm.idx = idx;
  sgm.tile = (int)(local / T);
  sgm.k0 = (int)(local - (long)sgm.tile * T);
  const long k1 = (long)sgm.k0 + (end - u);
  sgm.k1 = k1 < T ? (int)k1 : T;
  return sgm;
}
// workgroup w's plan entry: its first segment continues a tile (k0 > 0) and ends it (k1 == T) -> w is the tile's last contributor
__device__ __forceinline__ void sk_write_plan(const SKGroup& g, int w, long u0, long end) {
  if (threadIdx.x != 0) return;
  int idx = 0;
  const SKSeg sgm = sk_locate(g, u0, end, idx);
  int e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  if (sgm.k0 > 0 && sgm.k1 == g.T[sgm.idx]) {
    const long ts = u0 - sgm.k0;                         // first K-unit of the tile
    long wf = ts * g.G / g.total;                        // its first workgroup: the largest w' with w' * U / G <= ts
    while ((wf + 1) * g.total / g.G <= ts) ++wf;
    while (wf * g.total / g.G > ts) --wf;
    e0 = sgm.idx; e1 = sgm.tile; e2 = (int)wf; e3 = 1;
  }
  int* dst = g.plan + 4 * w;
  dst[0] = e0; dst[1] = e1; dst[2] = e2; dst[3] = e3;
}

__device__ __forceinline__ OutSpec sk_out(const SKGroup& g, const WgradParams& p, const SKSeg& sgm, int T, int w, int tile_floats) {
  OutSpec o;
  if (sgm.k0 == 0 && sgm.k1 == T) {
    o.w = p.partial; o.b = p.bias_partial; o.mode = p.rmw ? 1 : 0;
  } else {
    float* slot = g.slots + ((size_t)w * 2 + (sgm.k0 > 0 ? 0 : 1)) * g.slot_floats;
    o.w = slot; o.b = p.bias_partial ? slot + tile_floats : nullptr; o.mode = 2;
  }
  return o;
}

template <int TW, bool DMA, int BF = 0>
__global__ __launch_bounds__(256, 2)
void conv_wgrad_taps_sk_kernel(const SKGroup g) {
  const int w = blockIdx.x;
  long u = (long)w * g.total / g.G;
  const long end = (long)(w + 1) * g.total / g.G;
  sk_write_plan(g, w, u, end);
  int idx = 0;
  bool first = true;
#pragma clang loop unroll(disable)
  while (u < end) {
    const SKSeg sgm = sk_locate(g, u, end, idx);
    const WgradParams& p = g.p[sgm.idx];
    const int gx = g.gx[sgm.idx];
    if (!first) __syncthreads();   // the previous segment's LDS tiles / bias reduction are done with
    wgrad_taps_body<TW, DMA, BF>(p, sgm.tile % gx, sgm.tile / gx, sgm.k0, sgm.k1, sk_out(g, p, sgm, g.T[sgm.idx], w, 64 * 9 * 64), sk_tid());
    first = false;
    u += sgm.k1 - sgm.k0;
  }
}

// resident workgroups per CU the register / LDS budget of each tile shape allows (= the split-K kernels' occupancy)
constexpr int sk_wg_per_cu(int bm, int bn) { return bm * bn >= 128 * 128 ? 2 : (bm * bn >= 128 * 64 ? 3 : 5); }

template <int BM, int BN, int WM, int WN, bool DMA, int BF = 0>
__global__ __launch_bounds__(64 * WM * WN, (BM * BN >= 128 * 128 ? 2 : (BM * BN >= 128 * 64 ? 4 : 6)))
void conv_wgrad_sk_kernel(const SKGroup g) {
  const int w = blockIdx.x;
  long u = (long)w * g.total / g.G;
  const long end = (long)(w + 1) * g.total / g.G;
  sk_write_plan(g, w, u, end);
  int idx = 0;
  bool first = true;
#pragma clang loop unroll(disable)
  while (u < end) {
    const SKSeg sgm = sk_locate(g, u, end, idx);
    const WgradParams& p = g.p[sgm.idx];
    const int gx = g.gx[sgm.idx];
    if (!first) __syncthreads();
    const int ke = sgm.k1 * BK;
    wgrad_body<BM, BN, WM, WN, DMA, BF>(p, sgm.tile % gx, sgm.tile / gx, sgm.k0 * BK, ke < p.K ? ke : p.K, sk_out(g, p, sgm, g.T[sgm.idx], w, BM * BN), sk_tid());
    first = false;
    u += sgm.k1 - sgm.k0;
  }
}

// Fix-up: adds the partials of every split tile in workgroup order.  grid (256-element chunks of the tile image, kFixY).
// Every block first packs the plan (one entry per workgroup, written by the stream-K kernel) into a list of the
// split tiles' last contributors, in workgroup order; block (x, y) of the fix-up takes list entries y, y + kFixY, ...
// Per entry the block is 64 float4 columns x 4 groups: group q sums partials q, q+4, ... (4 loads in flight), the groups are
// combined as (g0+g1)+(g2+g3): a fixed order -> run-to-run reproducible.
// TAPS = 9: tile image [64][9][64] (+64 bias sums); TAPS = 1: [BM][BN] of one tap (+BM bias sums).
constexpr int kFixY = 64;
template <int BM, int BN, int TAPS>
__global__ __launch_bounds__(256) void wgrad_sk_fixup_kernel(const SKGroup g) {
  constexpr int TILE = BM * TAPS * BN;
  __shared__ f32x4 sm[4][64];
  __shared__ int list[1280];
  __shared__ int wave_cnt[4];
  const int tid = threadIdx.x;
  // the plan's live entries (last-contributor workgroups), ascending: every block packs its own copy (G <= 1280 entries: five
  // ballot rounds — cheaper than the dispatch of a separate one-block pass, which costs ~4.6 us inside the graph)
  int n_list = 0;
  {
    const int lane = tid & 63, wv = tid >> 6;
    for (int base = 0; base < g.G; base += 256) {
      const int w = base + tid;
      const bool on = w < g.G && g.plan[4 * w + 3] != 0;
      const unsigned long long m = __ballot(on);
      if (lane == 0) wave_cnt[wv] = __popcll(m);
      __syncthreads();
      int off = n_list, tot = 0;
      for (int q = 0; q < 4; ++q) { if (q < wv) off += wave_cnt[q]; tot += wave_cnt[q]; }
      if (on) list[off + __popcll(m & ((1ull << lane) - 1ull))] = w;
      n_list += tot;
      __syncthreads();
    }
  }
  const int col = tid & 63, grp = tid >> 6;
  const int e = (blockIdx.x * 64 + col) * 4;              // element of the tile image (+ bias tail)
  for (int li = blockIdx.y; li < n_list; li += kFixY) {
    const int wl = list[li];
    const int idx = g.plan[4 * wl], tile = g.plan[4 * wl + 1], wf = g.plan[4 * wl + 2];
    const WgradParams& p = g.p[idx];
    const int gx = g.gx[idx];
    const int bx = tile % gx, by = tile / gx;
    const bool is_bias = e >= TILE;
    const bool live = e < TILE + BM && !(is_bias && !(p.bias_partial && bx == 0));
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const int n = wl - wf + 1;                            // partial i: slot 1 of workgroup wf (i = 0), slot 0 of workgroup wf + i
      auto src = [&](int i) { return reinterpret_cast<const f32x4*>(g.slots + ((size_t)(wf + i) * 2 + (i == 0 ? 1 : 0)) * g.slot_floats + e); };
      int i = grp;
      for (; i + 12 < n; i += 16) {
        const f32x4 a0 = *src(i), a1 = *src(i + 4), a2 = *src(i + 8), a3 = *src(i + 12);
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; i < n; i += 4) s += *src(i);
    }
    __syncthreads();                                        // (previous entry's combine has read sm)
    sm[grp][col] = s;
    __syncthreads();
    if (grp != 0 || !live) continue;
    s = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]);
    const int co0 = by * BM;
    if (is_bias) {
      const int c = e - TILE;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (co0 + c + q < p.Cout) { float* dst = p.bias_partial + co0 + c + q; *dst = (p.rmw ? *dst : 0.f) + s[q]; }
      continue;
    }
    int cl, tap, cil, ci0, T9;
    if (TAPS == 9) { cl = e / (9 * BN); const int r = e - cl * 9 * BN; tap = r / BN; cil = r - tap * BN; ci0 = bx * BN; T9 = 9; }
    else { cl = e / BN; cil = e - cl * BN; tap = bx / p.nci; ci0 = (bx - tap * p.nci) * BN; T9 = p.ks * p.ks; }
    const int co = co0 + cl, ci = ci0 + cil;
    if (co >= p.Cout || ci >= p.Cin) continue;              // (Cin % 4 == 0 on this path: a float4 is inside or outside as a whole)
    f32x4* dst = reinterpret_cast<f32x4*>(p.partial + ((size_t)co * T9 + tap) * p.Cin + ci);
    *dst = p.rmw ? *dst + s : s;
  }
}

// out[i] (+)= sum_k partial[k][i], fixed order: 4 interleaved groups (k mod 4) summed ascending, then ((g0+g1)+(g2+g3)).
// Block = 64 column-threads (float4 each) x 4 groups, so a 128-way split is a chain of 32 loads per thread, 4 in flight.
// One launch serves up to kMaxSlabs slab sets (the dW and dbias slabs of every problem of a grouped launch).
constexpr int kMaxSlabs = 2 * kMaxGroup;
struct SlabSet { const float* partial; float* out; long n; int splits, accumulate; };
struct SlabGroup {
  int count;
  int blk_end[kMaxSlabs];
  SlabSet e[kMaxSlabs];
};

__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabGroup sg) {
  __shared__ f32x4 sm[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  int idx = 0;
  while (idx + 1 < sg.count && (int)blockIdx.x >= sg.blk_end[idx]) ++idx;
  const long bx = (int)blockIdx.x - (idx ? sg.blk_end[idx - 1] : 0);
  const float* __restrict__ partial = sg.e[idx].partial;
  float* __restrict__ out = sg.e[idx].out;
  const long n = sg.e[idx].n;
  const int splits = sg.e[idx].splits, accumulate = sg.e[idx].accumulate;
  const long i = (bx * 64 + tx) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const bool vec = (i + 3 < n) && ((n & 3) == 0);
    int k = g;
    if (vec) {   // 4 slab rows in flight per thread (the loads are independent; the adds keep the ascending order)
      for (; k + 12 < splits; k += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(partial + (size_t)k * n + i);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + 4) * n + i);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + 8) * n + i);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(partial + (size_t)(k + 12) * n + i);
        s += a0; s += a1; s += a2; s += a3;
      }
    }
    for (; k < splits; k += 4) {
      const float* src = partial + (size_t)k * n + i;
      if (vec) s += *reinterpret_cast<const f32x4*>(src);
      else for (int e = 0; e < 4; ++e) if (i + e < n) s[e] += src[e];
    }
  }
  sm[g][tx] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    f32x4 t = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
    for (int e = 0; e < 4; ++e)
      if (i + e < n) out[i + e] = (accumulate ? out[i + e] : 0.f) + t[e];
  }
}

// small / unaligned Cin (the RGB input convs, T*Cin <= 32 columns): VALU kernel, LDS-tiled.
// One workgroup per (pixel range = split, tile of 128 output channels): per chunk of 32 pixels the dy rows and the
// gathered input patches are staged in LDS; thread (co, half) owns 16 of the <=32 (tap,ci) columns of filter row co.
__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgradParams p, int splits) {
  __shared__ float dyS[32][128 + 1];
  __shared__ float xS[32][32 + 1];
  const int tid = threadIdx.x, co_l = tid & 127, jh = tid >> 7;
  const int split = blockIdx.x, co0 = blockIdx.y * 128, co = co0 + co_l;
  const int T = p.ks * p.ks, ncol = T * p.Cin;
  const int k_begin = split * p.k_per_split, k_end = min(p.K, k_begin + p.k_per_split);
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  float bsum = 0.f;
  for (int kb = k_begin; kb < k_end; kb += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 128; e += 256) {
      const int k = e >> 7, c = e & 127, pix = kb + k;
      float gv = (pix < k_end && co0 + c < p.Cout) ? p.dy[(size_t)pix * p.lddy + co0 + c] : 0.f;
      if (p.dys && pix < k_end && co0 + c < p.Cout) gv *= act_deriv(p.dys[(size_t)pix * p.lddys + co0 + c], p.dys_act, p.dys_pre);
      dyS[k][c] = gv;
    }
    for (int e = tid; e < 32 * 32; e += 256) {
      const int k = e >> 5, j = e & 31, pix = kb + k;
      float v = 0.f;
      if (pix < k_end && j < ncol) {
        const int tap = j / p.Cin, ci = j - tap * p.Cin, kh = tap / p.ks, kw = tap - kh * p.ks;
        const int n = pix / (p.OH * p.OW), r = pix - n * (p.OH * p.OW);
        const int oy = r / p.OW, ox = r - oy * p.OW;
        const int iy = oy * p.stride - p.pad + kh, ix = ox * p.stride - p.pad + kw;
        if (iy >= 0 && ix >= 0 && iy < p.H && ix < p.W) {
          v = p.x[(size_t)((n * p.H + iy) * p.W + ix) * p.ldx + ci];
          if (p.in_op == CLC_IN_SQUARE) v *= v;
        }
      }
      xS[k][j] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 32; ++k) {
      const float g = dyS[k][co_l];
      bsum += g;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(g, xS[k][jh * 16 + j], acc[j]);
    }
  }
  if (co < p.Cout) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (jh * 16 + j < ncol) {
        float* dst = p.partial + ((size_t)split * p.Cout + co) * ncol + jh * 16 + j;
        *dst = p.rmw ? *dst + acc[j] : acc[j];
      }
    if (jh == 0 && p.bias_partial) {
      float* dst = p.bias_partial + (size_t)split * p.Cout + co;
      *dst = p.rmw ? *dst + bsum : bsum;
    }
  }
}

struct Plan { int bm, bn, splits, k_per_split, nci; bool small; int taps; };   // taps = TW of the all-taps kernel, or 0

int taps_mode() {   // CLC_WGRAD_TAPS: 0 = never, 1 = where the tap-per-workgroup plan is not 128x128, 2 = wherever eligible (default)
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("CLC_WGRAD_TAPS"); mode = e ? atoi(e) : 2; }
  return mode;
}

Plan make_plan(const clc_wgrad_desc* d) {
  Plan pl;
  pl.taps = 0;
  const long K = (long)d->N * d->OH * d->OW;
  const int T = d->ks * d->ks;
  pl.small = !((d->Cin % 4 == 0) && (d->ldx % 4 == 0) && (d->Cout % 4 == 0) && (d->lddy % 4 == 0));
  if (pl.small) {
    pl.bm = pl.bn = 0; pl.nci = 1;
    int splits = (int)((K + 255) / 256);
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    long kps = (K + splits - 1) / splits;
    pl.k_per_split = (int)kps; pl.splits = (int)((K + kps - 1) / kps);
    return pl;
  }
  // Tile and K-split choice by a small cost model (cycles): time = max(longest MFMA chain of one workgroup,
  // total MFMA work / 256 CUs) + slab traffic of the fixed-order reduce (launch + 2 x splits x |dW| bytes at ~4 TB/s).
  const int bms[2] = {128, 64}, bns[2] = {128, 64};
  const long wsz = (long)d->Cout * T * d->Cin;
  long best_cost = -1, splits = 1;
  pl.bm = 64; pl.bn = 64;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int bm = bms[a], bn = bns[b];
      if (bm == 128 && d->Cout < 128) continue;
      if (bn == 128 && (d->Cin < 128 || d->Cin % 128 != 0)) continue;
      const long tiles = (long)((d->Cout + bm - 1) / bm) * ((d->Cin + bn - 1) / bn) * T;
      // cycles per K-tile per workgroup, de-rated by the measured efficiency of the tile shape (operand re-use:
      // 128x128 reaches ~80 TF, 64x64 ~45 TF on large layers)
      const long eff_pct = (bm == 128 && bn == 128) ? 100 : ((bm == 128 || bn == 128) ? 80 : 55);
      const long mfma_per_ktile = (long)(bm / 32) * (bn / 32) / 4 * 16 * 64 * 100 / eff_pct;
      const long max_sp = K / 64 > 0 ? K / 64 : 1;
      for (long sp = 1; sp <= 256 && sp <= max_sp; ++sp) {
        const long ktiles = (K / sp + BK - 1) / BK;
        const long chain = ktiles * mfma_per_ktile + 6000;
        // a CU retires one workgroup's MFMA chain per chain-time (two co-resident workgroups share the matrix pipes), so
        // the kernel takes ceil(workgroups / 256) chains: 288 workgroups cost as much as 512 — pick splits that fill whole rounds
        const long rounds = (tiles * sp + 255) / 256;
        const long slab = sp > 1 ? 12000 + sp * wsz / 200 : 0;
        const long cost = rounds * chain + slab;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; pl.bm = bm; pl.bn = bn; splits = sp; }
      }
    }
  {
    auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    const int la = lg(d->OW), lb = lg(d->OH * d->OW), tw = d->OW < 32 ? d->OW : 32;
    const bool eligible = d->ks == 3 && d->stride == 1 && d->pad == 1 && la >= 3 && lb >= 0 && d->OH >= 32 / tw;
    const int mode = taps_mode();
    if (eligible && (mode == 2 || (mode == 1 && !(pl.bm == 128 && pl.bn == 128)))) {
      pl.taps = tw; pl.bm = 64; pl.bn = 64; pl.nci = (d->Cin + 63) / 64;
      const long ktiles = K / 32, tiles = (long)((d->Cout + 63) / 64) * pl.nci;
      long sp = (ktiles + 15) / 16;                    // <= ~16 K-tiles (a ~60 us MFMA chain) per workgroup,
      const long fill = (256 + tiles - 1) / tiles;     // enough workgroups for one per CU while K-tiles last (>= 2 each),
      if (sp < fill) sp = fill;
      if (sp > ktiles / 2) sp = ktiles / 2 > 0 ? ktiles / 2 : 1;
      const long cap = 512 / tiles > 0 ? 512 / tiles : 1;
      if (sp > cap) sp = cap;                          // and no more slabs than two workgroups per CU need
      const long kt = (ktiles + sp - 1) / sp;
      pl.k_per_split = (int)(kt * 32);
      pl.splits = (int)((ktiles + kt - 1) / kt);
      return pl;
    }
  }
  pl.nci = (d->Cin + pl.bn - 1) / pl.bn;
  long kps = (K + splits - 1) / splits;
  kps = (kps + BK - 1) / BK * BK;
  pl.k_per_split = (int)kps;
  pl.splits = (int)((K + kps - 1) / kps);
  return pl;
}

}  // namespace

extern "C" size_t clc_conv2d_wgrad_workspace_bytes(const clc_wgrad_desc* d) {
  if (!d) return 0;
  Plan pl = make_plan(d);
  const size_t wsz = (size_t)d->Cout * d->ks * d->ks * d->Cin;
  return ((size_t)pl.splits * (wsz + d->Cout) + 64) * sizeof(float);
}

namespace {

// validates one descriptor and fills the kernel parameters for its plan
// sk: the problem is part of a stream-K group (partial tiles live in the group workspace: no per-problem slabs, except on the
// small-Cin split path)
int prepare(const clc_wgrad_desc* d, Plan& pl, WgradParams& p, bool sk = false) {
  CLC_CHECK(d && d->x && d->dy && d->dw, "clc_conv2d_wgrad: null pointer");
  CLC_CHECK(d->ks == 1 || d->ks == 3, "clc_conv2d_wgrad: ks must be 1 or 3");
  CLC_CHECK(d->stride == 1 || d->stride == 2, "clc_conv2d_wgrad: stride must be 1 or 2");
  CLC_CHECK(d->OH == (d->H + 2 * d->pad - d->ks) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->ks) / d->stride + 1,
            "clc_conv2d_wgrad: output dims inconsistent");
  CLC_CHECK(d->ldx >= d->Cin && d->lddy >= d->Cout, "clc_conv2d_wgrad: ld too small");
  CLC_CHECK((long)d->N * d->H * d->W < (1l << 31) && (long)d->N * d->OH * d->OW < (1l << 31), "clc_conv2d_wgrad: too many pixels");
  pl = make_plan(d);
  if (!(sk && clc_tuning[CLC_TUNE_WGRAD_STREAMK] && !pl.small))
    CLC_CHECK(d->workspace && d->workspace_bytes >= clc_conv2d_wgrad_workspace_bytes(d), "clc_conv2d_wgrad: workspace too small");
  const int T = d->ks * d->ks;
  const size_t wsz = (size_t)d->Cout * T * d->Cin;
  p.x = d->x; p.dy = d->dy;
  const bool direct = (pl.splits == 1);   // one slab: write / accumulate the result in place, no reduce launch
  p.rmw = direct && d->accumulate;
  p.partial = direct ? d->dw : (float*)d->workspace;
  p.bias_partial = d->dbias ? (direct ? d->dbias : (float*)d->workspace + (size_t)pl.splits * wsz) : nullptr;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx;
  p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.lddy = d->lddy;
  p.ks = d->ks; p.stride = d->stride; p.pad = d->pad; p.in_op = d->in_op;
  p.K = d->N * d->OH * d->OW; p.k_per_split = pl.k_per_split; p.nci = pl.nci;
  const size_t xb = ((size_t)d->N * d->H * d->W - 1) * d->ldx * 4 + (size_t)d->Cin * 4;
  const size_t db = ((size_t)d->N * d->OH * d->OW - 1) * d->lddy * 4 + (size_t)d->Cout * 4;
  CLC_CHECK(xb < (1ull << 31) && db < (1ull << 31), "clc_conv2d_wgrad: tensor larger than 2 GiB");
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db;
  auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  const int la = lg(d->OW), lb = lg(d->OH * d->OW);
  p.ow_shift = (la >= 0 && lb >= 0) ? la : -1; p.img_shift = (la >= 0 && lb >= 0) ? lb : -1;
  p.dys = d->dys; p.lddys = d->lddys; p.dys_act = d->dys_act; p.dys_pre = d->dys_pre; p.dys_bytes = 0;
  if (d->dys) {
    CLC_CHECK(d->lddys >= d->Cout, "clc_conv2d_wgrad: bad dys");
    const size_t sb = ((size_t)d->N * d->OH * d->OW - 1) * d->lddys * 4 + (size_t)d->Cout * 4;
    CLC_CHECK(sb < (1ull << 31), "clc_conv2d_wgrad: dys larger than 2 GiB");
    p.dys_bytes = (unsigned)sb;
  }
  if (pl.small) CLC_CHECK(T * d->Cin <= 32, "clc_conv2d_wgrad: small/unaligned path needs ks*ks*Cin <= 32 (got %d)", T * d->Cin);
  else CLC_CHECK(aligned16(d->x) && aligned16(d->dy), "clc_conv2d_wgrad: unaligned pointers");
  return 0;
}

struct Pending { const clc_wgrad_desc* d; Plan pl; WgradParams p; };

template <int BM, int BN>
int launch_variant(const Pending* pend, int n, hipStream_t st) {
  WgradGroup g;
  g.count = 0;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const Pending& e = pend[i];
    if (e.pl.small || e.pl.taps || e.pl.bm != BM || e.pl.bn != BN) continue;
    const int T = e.d->ks * e.d->ks;
    const int gx = e.pl.nci * T, gy = (e.d->Cout + BM - 1) / BM;
    total += gx * gy * e.pl.splits;
    g.gx[g.count] = gx; g.gy[g.count] = gy; g.wg_end[g.count] = total; g.p[g.count] = e.p;
    ++g.count;
  }
  if (g.count == 0) return 0;
  const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
  if (g.count == 1) {
    const int last = g.wg_end[0] / (g.gx[0] * g.gy[0]);
    hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, 2, 2>), dim3(g.gx[0], g.gy[0], last), dim3(256), lds, st, g.p[0]);
  } else {
    hipLaunchKernelGGL((conv_wgrad_grouped_kernel<BM, BN, 2, 2>), dim3(total), dim3(256), lds, st, g);
  }
  CLC_LAUNCH_CHECK();
  return 0;
}

template <int TW>
int launch_taps(const Pending* pend, int n, hipStream_t st) {
  WgradGroup g;
  g.count = 0;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const Pending& e = pend[i];
    if (e.pl.small || e.pl.taps != TW) continue;
    const int gx = e.pl.nci, gy = (e.d->Cout + 63) / 64;
    total += gx * gy * e.pl.splits;
    g.gx[g.count] = gx; g.gy[g.count] = gy; g.wg_end[g.count] = total; g.p[g.count] = e.p;
    ++g.count;
  }
  if (g.count == 0) return 0;
  constexpr int XP = (32 / TW + 2) * (TW + 2);
  const size_t lds = (size_t)2 * (32 * 64 + XP * 64) * sizeof(float);
  static PerDeviceOnce attr_once;   // > 64 KiB of dynamic LDS needs the opt-in (first call happens before any graph capture)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_grouped_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (g.count == 1) {
    const int last = g.wg_end[0] / (g.gx[0] * g.gy[0]);
    hipLaunchKernelGGL((conv_wgrad_taps_kernel<TW>), dim3(g.gx[0], g.gy[0], last), dim3(256), lds, st, g.p[0]);
  } else {
    hipLaunchKernelGGL((conv_wgrad_taps_grouped_kernel<TW>), dim3(total), dim3(256), lds, st, g);
  }
  CLC_LAUNCH_CHECK();
  return 0;
}

// launches everything pending: one grid per tile shape, then one fixed-order reduce over every slab set
int flush(Pending* pend, int& n, hipStream_t st) {
  if (n == 0) return 0;
  for (int i = 0; i < n; ++i)
    if (pend[i].pl.small) {
      hipLaunchKernelGGL(wgrad_small_kernel, dim3(pend[i].pl.splits, (pend[i].d->Cout + 127) / 128), dim3(256), 0, st, pend[i].p, pend[i].pl.splits);
      CLC_LAUNCH_CHECK();
    }
  int rc;
  if ((rc = launch_variant<128, 128>(pend, n, st)) < 0) return rc;
  if ((rc = launch_variant<128, 64>(pend, n, st)) < 0) return rc;
  if ((rc = launch_variant<64, 128>(pend, n, st)) < 0) return rc;
  if ((rc = launch_variant<64, 64>(pend, n, st)) < 0) return rc;
  if ((rc = launch_taps<32>(pend, n, st)) < 0) return rc;
  if ((rc = launch_taps<16>(pend, n, st)) < 0) return rc;
  if ((rc = launch_taps<8>(pend, n, st)) < 0) return rc;
  SlabGroup sg;
  sg.count = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const Pending& e = pend[i];
    if (e.pl.splits == 1) continue;
    const long wsz = (long)e.d->Cout * e.d->ks * e.d->ks * e.d->Cin;
    blocks += (int)((wsz + 255) / 256);
    sg.e[sg.count] = SlabSet{e.p.partial, e.d->dw, wsz, e.pl.splits, e.d->accumulate};
    sg.blk_end[sg.count++] = blocks;
    if (e.d->dbias) {
      blocks += (e.d->Cout + 255) / 256;
      sg.e[sg.count] = SlabSet{e.p.bias_partial, e.d->dbias, (long)e.d->Cout, e.pl.splits, e.d->accumulate};
      sg.blk_end[sg.count++] = blocks;
    }
  }
  if (sg.count) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, sg);
    CLC_LAUNCH_CHECK();
  }
  n = 0;
  return 0;
}

// ---- stream-K launches ---------------------------------------------------------------------------
constexpr int kCUs = 256;
constexpr size_t sk_slot_floats(int tile_floats, int bm) { return (size_t)((tile_floats + bm + 255) / 256 * 256); }
constexpr size_t kSKPlanFloats = 4 * 1280 + 1536;   // plan entries + compact list (ints) in front of every family's slots
// grid = the workgroups that are resident at once (per tile shape) -> every workgroup starts immediately and runs the same number
// of K-units; workspace regions, one per kernel family (the families' launches and fix-ups run back to back)
constexpr size_t kSKRegion[7] = {
    (size_t)kCUs * sk_wg_per_cu(128, 128) * 2 * sk_slot_floats(128 * 128, 128), (size_t)kCUs * sk_wg_per_cu(128, 64) * 2 * sk_slot_floats(128 * 64, 128),
    (size_t)kCUs * sk_wg_per_cu(64, 128) * 2 * sk_slot_floats(64 * 128, 64),    (size_t)kCUs * sk_wg_per_cu(64, 64) * 2 * sk_slot_floats(64 * 64, 64),
    (size_t)kCUs * 2 * 2 * sk_slot_floats(64 * 9 * 64, 64), (size_t)kCUs * 2 * 2 * sk_slot_floats(64 * 9 * 64, 64),
    (size_t)kCUs * 2 * 2 * sk_slot_floats(64 * 9 * 64, 64)};
constexpr size_t sk_total_floats() { size_t t = 0; for (int i = 0; i < 7; ++i) t += kSKRegion[i] + kSKPlanFloats; return t; }

// fills the group of one family; returns false when no problem of the family is pending
template <typename Pred, typename Geo>
bool sk_collect(const Pending* pend, int n, SKGroup& g, float* region, int slot_floats, int max_g, int min_units, Pred pred, Geo geo) {
  g.count = 0; g.total = 0; g.plan = reinterpret_cast<int*>(region); g.slots = region + kSKPlanFloats; g.slot_floats = slot_floats;
  for (int i = 0; i < n; ++i) {
    const Pending& e = pend[i];
    if (e.pl.small || !pred(e)) continue;
    int gx, gy, T;
    geo(e, gx, gy, T);
    g.gx[g.count] = gx; g.gy[g.count] = gy; g.T[g.count] = T;
    g.total += (long)gx * gy * T;
    g.unit_end[g.count] = g.total;
    g.p[g.count] = e.p;
    g.p[g.count].partial = e.d->dw;                 // full tiles and the fix-up write the gradient itself
    g.p[g.count].bias_partial = e.d->dbias;
    g.p[g.count].rmw = e.d->accumulate;
    ++g.count;
  }
  if (g.count == 0) return false;
  long G = g.total / min_units;                     // a workgroup's chain of MFMAs must amortise its prologue / partial-tile write
  if (clc_tuning[CLC_TUNE_SK_HALF] && max_g > kCUs) max_g = kCUs;   // one workgroup per CU: leaves wave slots to a concurrent stream
  g.G = (int)(G < 1 ? 1 : (G > max_g ? max_g : G));
  return true;
}

// problems whose operands need no arithmetic on the way into LDS (no fused activation derivative on dy, no squared input): LDS-DMA staging
inline bool dma_ok(const Pending& e) { return clc_tuning[CLC_TUNE_WGRAD_DMA] && e.d->dys == nullptr && e.d->in_op == CLC_IN_NONE; }
// reduced-precision mode (CLC_TUNE_BF16): the LDS-DMA-staged problems of maps larger than 16x16 (= the analysis / synthesis transforms and
// the reference encoder; the entropy-parameter nets on the 16x16 latents stay f32)
inline int bf_ok(const Pending& e) {
  if (!dma_ok(e)) return 0;
  if (clc_tuning[CLC_TUNE_BF16] && (long)e.d->OH * e.d->OW > 256) return 1;
  // (key 24) f32 products from three-way bf16 splits on the bf16 matrix cores (BF = 2): bit 0 = the all-taps 3x3 kernels, bit 1 = the tiled kernels
  // (the 64 x 64 tiles split 16 values per 6 MFMAs — VALU-bound, measured 0.93-0.99x: they keep the f32 MFMAs)
  if (!e.pl.taps && e.pl.bm == 64 && e.pl.bn == 64) return 0;
  return (clc_tuning[CLC_TUNE_SPLIT] & (e.pl.taps ? 1 : 2)) ? 2 : 0;
}

template <int BM, int BN, bool DMA, int BF = 0>
int launch_variant_sk(const Pending* pend, int n, float* region, hipStream_t st) {
  static thread_local SKGroup g;   // (host staging of the kernel argument; launches are issued by one host thread per process)
  constexpr int slot = (int)sk_slot_floats(BM * BN, BM);
  if (!sk_collect(pend, n, g, region, slot, kCUs * sk_wg_per_cu(BM, BN), 16, [](const Pending& e) { return !e.pl.taps && e.pl.bm == BM && e.pl.bn == BN && dma_ok(e) == DMA && bf_ok(e) == BF; },
                  [](const Pending& e, int& gx, int& gy, int& T) {
                    gx = e.pl.nci * e.d->ks * e.d->ks; gy = (e.d->Cout + BM - 1) / BM; T = (e.p.K + BK - 1) / BK;
                  })) return 0;
  const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
  hipLaunchKernelGGL((conv_wgrad_sk_kernel<BM, BN, 2, 2, DMA, BF>), dim3(g.G), dim3(256), lds, st, g);
  CLC_LAUNCH_CHECK();
  if (g.G > 1) {
    hipLaunchKernelGGL((wgrad_sk_fixup_kernel<BM, BN, 1>), dim3((BM * BN + BM + 255) / 256, kFixY), dim3(256), 0, st, g);
    CLC_LAUNCH_CHECK();
  }
  return 0;
}

template <int TW, bool DMA, int BF = 0>
int launch_taps_sk(const Pending* pend, int n, float* region, hipStream_t st) {
  static thread_local SKGroup g;
  constexpr int slot = (int)sk_slot_floats(64 * 9 * 64, 64);
  if (!sk_collect(pend, n, g, region, slot, kCUs * 2, 4, [](const Pending& e) { return e.pl.taps == TW && dma_ok(e) == DMA && bf_ok(e) == BF; },
                  [](const Pending& e, int& gx, int& gy, int& T) { gx = e.pl.nci; gy = (e.d->Cout + 63) / 64; T = e.p.K >> 5; })) return 0;
  constexpr int XP = (32 / TW + 2) * (TW + 2);
  const size_t lds = (size_t)2 * (32 * 64 + XP * 64) * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_taps_sk_kernel<TW, DMA, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_wgrad_taps_sk_kernel<TW, DMA, BF>), dim3(g.G), dim3(256), lds, st, g);
  CLC_LAUNCH_CHECK();
  if (g.G > 1) {
    hipLaunchKernelGGL((wgrad_sk_fixup_kernel<64, 64, 9>), dim3((64 * 9 * 64 + 64 + 255) / 256, kFixY), dim3(256), 0, st, g);
    CLC_LAUNCH_CHECK();
  }
  return 0;
}

// stream-K flush: the RGB-head problems keep the split form (their own workspaces), everything else one grid per family
int flush_sk(Pending* pend, int& n, float* ws, hipStream_t st) {
  if (n == 0) return 0;
  SlabGroup sg;
  sg.count = 0;
  int blocks = 0;
  for (int i = 0; i < n; ++i)
    if (pend[i].pl.small) {
      const Pending& e = pend[i];
      hipLaunchKernelGGL(wgrad_small_kernel, dim3(e.pl.splits, (e.d->Cout + 127) / 128), dim3(256), 0, st, e.p, e.pl.splits);
      CLC_LAUNCH_CHECK();
      if (e.pl.splits == 1) continue;
      const long wsz = (long)e.d->Cout * e.d->ks * e.d->ks * e.d->Cin;
      blocks += (int)((wsz + 255) / 256);
      sg.e[sg.count] = SlabSet{e.p.partial, e.d->dw, wsz, e.pl.splits, e.d->accumulate};
      sg.blk_end[sg.count++] = blocks;
      if (e.d->dbias) {
        blocks += (e.d->Cout + 255) / 256;
        sg.e[sg.count] = SlabSet{e.p.bias_partial, e.d->dbias, (long)e.d->Cout, e.pl.splits, e.d->accumulate};
        sg.blk_end[sg.count++] = blocks;
      }
    }
  if (sg.count) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, sg);
    CLC_LAUNCH_CHECK();
  }
  int rc;
  float* r = ws;
  if ((rc = launch_variant_sk<128, 128, true>(pend, n, r, st)) < 0) return rc;    // (the two launches of a family share its region: stream order)
  if ((rc = launch_variant_sk<128, 128, true, 1>(pend, n, r, st)) < 0) return rc;    // (reduced-precision mode: bf16 MFMA)
  if ((rc = launch_variant_sk<128, 128, true, 2>(pend, n, r, st)) < 0) return rc;    // (f32 products from bf16 pieces)
  if ((rc = launch_variant_sk<128, 128, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[0] + kSKPlanFloats;
  if ((rc = launch_variant_sk<128, 64, true>(pend, n, r, st)) < 0) return rc;    // (the two launches of a family share its region: stream order)
  if ((rc = launch_variant_sk<128, 64, true, 1>(pend, n, r, st)) < 0) return rc;    // (reduced-precision mode: bf16 MFMA)
  if ((rc = launch_variant_sk<128, 64, true, 2>(pend, n, r, st)) < 0) return rc;    // (f32 products from bf16 pieces)
  if ((rc = launch_variant_sk<128, 64, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[1] + kSKPlanFloats;
  if ((rc = launch_variant_sk<64, 128, true>(pend, n, r, st)) < 0) return rc;    // (the two launches of a family share its region: stream order)
  if ((rc = launch_variant_sk<64, 128, true, 1>(pend, n, r, st)) < 0) return rc;    // (reduced-precision mode: bf16 MFMA)
  if ((rc = launch_variant_sk<64, 128, true, 2>(pend, n, r, st)) < 0) return rc;    // (f32 products from bf16 pieces)
  if ((rc = launch_variant_sk<64, 128, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[2] + kSKPlanFloats;
  if ((rc = launch_variant_sk<64, 64, true>(pend, n, r, st)) < 0) return rc;    // (the two launches of a family share its region: stream order)
  if ((rc = launch_variant_sk<64, 64, true, 1>(pend, n, r, st)) < 0) return rc;    // (reduced-precision mode: bf16 MFMA)
  if ((rc = launch_variant_sk<64, 64, true, 2>(pend, n, r, st)) < 0) return rc;    // (f32 products from bf16 pieces)
  if ((rc = launch_variant_sk<64, 64, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[3] + kSKPlanFloats;
  if ((rc = launch_taps_sk<32, true>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<32, true, 1>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<32, true, 2>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<32, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[4] + kSKPlanFloats;
  if ((rc = launch_taps_sk<16, true>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<16, true, 1>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<16, true, 2>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<16, false>(pend, n, r, st)) < 0) return rc;
  r += kSKRegion[5] + kSKPlanFloats;
  if ((rc = launch_taps_sk<8, true>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<8, true, 1>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<8, true, 2>(pend, n, r, st)) < 0) return rc;
  if ((rc = launch_taps_sk<8, false>(pend, n, r, st)) < 0) return rc;
  n = 0;
  return 0;
}

int variant_id(const Plan& pl) { return pl.small ? 1 : (pl.taps ? 64900 + pl.taps : pl.bm * 1000 + pl.bn); }

}  // namespace

extern "C" int clc_conv2d_wgrad(const clc_wgrad_desc* d, clc_stream_t stream) {
  Pending e;
  e.d = d;
  int rc = prepare(d, e.pl, e.p);
  if (rc < 0) return rc;
  int n = 1;
  if ((rc = flush(&e, n, (hipStream_t)stream)) < 0) return rc;
  return variant_id(e.pl);  // kernel-variant id
}

extern "C" int clc_conv2d_wgrad_batched(const clc_wgrad_desc* descs, int count, clc_stream_t stream) {
  CLC_CHECK(descs && count >= 0, "clc_conv2d_wgrad_batched: bad arguments");
  Pending pend[kMaxGroup];
  int n = 0, rc;
  for (int i = 0; i < count; ++i) {
    const clc_wgrad_desc* d = descs + i;
    // two problems that write the same gradient buffer (a filter applied twice) must not share a launch
    bool clash = n == kMaxGroup;
    for (int j = 0; j < n && !clash; ++j)
      clash = pend[j].d->dw == d->dw || (d->dbias && pend[j].d->dbias == d->dbias) || pend[j].d->workspace == d->workspace;
    if (clash && (rc = flush(pend, n, (hipStream_t)stream)) < 0) return rc;
    pend[n].d = d;
    if ((rc = prepare(d, pend[n].pl, pend[n].p)) < 0) return rc;
    ++n;
  }
  return flush(pend, n, (hipStream_t)stream);
}

extern "C" int clc_conv2d_wgrad_variant(const clc_wgrad_desc* d) {
  if (!d) return -1;
  return variant_id(make_plan(d));
}

extern "C" size_t clc_conv2d_wgrad_group_workspace_bytes(void) { return sk_total_floats() * sizeof(float); }

extern "C" size_t clc_conv2d_wgrad_sk_workspace_bytes(const clc_wgrad_desc* d) {
  if (!d) return 0;
  if (clc_tuning[CLC_TUNE_WGRAD_STREAMK] && !make_plan(d).small) return 0;   // partial tiles go to the group workspace
  return clc_conv2d_wgrad_workspace_bytes(d);
}

extern "C" int clc_conv2d_wgrad_batched_sk(const clc_wgrad_desc* descs, int count, void* group_workspace, size_t group_workspace_bytes,
                                          clc_stream_t stream) {
  CLC_CHECK(descs && count >= 0, "clc_conv2d_wgrad_batched_sk: bad arguments");
  CLC_CHECK(group_workspace && group_workspace_bytes >= clc_conv2d_wgrad_group_workspace_bytes() && aligned16(group_workspace),
            "clc_conv2d_wgrad_batched_sk: group workspace too small / unaligned");
  if (!clc_tuning[CLC_TUNE_WGRAD_STREAMK]) return clc_conv2d_wgrad_batched(descs, count, stream);
  Pending pend[kMaxGroup];
  int n = 0, rc;
  for (int i = 0; i < count; ++i) {
    const clc_wgrad_desc* d = descs + i;
    // two problems that write the same gradient buffer (a filter applied twice) must not share a launch
    bool clash = n == kMaxGroup;
    for (int j = 0; j < n && !clash; ++j)
      clash = pend[j].d->dw == d->dw || (d->dbias && pend[j].d->dbias == d->dbias) || (d->workspace && pend[j].d->workspace == d->workspace);
    if (clash && (rc = flush_sk(pend, n, (float*)group_workspace, (hipStream_t)stream)) < 0) return rc;
    pend[n].d = d;
    if ((rc = prepare(d, pend[n].pl, pend[n].p, true)) < 0) return rc;
    ++n;
  }
  return flush_sk(pend, n, (float*)group_workspace, (hipStream_t)stream);
}
