// winattn.hip — (shifted-)window multi-head self-attention core, forward and backward (gfx950).
//
// Replaces the einsum / softmax / rearrange / roll / mask / relative-embedding chain of
// WMSA.forward, /root/reference/models/CLC_run.py:142-169 (== models/tcm.py:173-200).
// Windows are tiny (ws=8 -> 64 tokens, ws=4 -> 16 tokens; head_dim 8/16/32; ~2 % of the path's
// FLOPs), so this is a VALU kernel organised for zero HBM waste instead of an MFMA kernel:
//   * one wave per (window, head) pass; lane = query token (forward / backward pass 1) or key
//     token (backward pass 2); K/V (and Q/dO) tiles staged once in LDS, read as broadcasts
//   * cyclic shift, window partition, head split and the relative-position gather are index
//     arithmetic on the NHWC qkv tensor: nothing is rolled, permuted or materialised
//   * the shift mask is evaluated from (window, token) coordinates, never allocated
//   * backward rebuilds the probabilities from Q, K and the saved per-row log-sum-exp; the
//     relative-bias gradient is accumulated per lane-owned row across the windows of a
//     workgroup and folded in a fixed order -> bitwise reproducible (no float atomics).
#include "common.h"

namespace {

// exp via the hardware exp2 (v_exp_f32, ~1 ulp): the probabilities' arguments are <= 0 and forward / backward use the same
// function, so the softmax stays self-consistent; OCML's expf costs ~15 VALU instructions per call, this one 2 — and these
// kernels are VALU-bound (64 exps per lane per window sweep).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

struct AttnParams {
  const float* qkv; const float* relbias; float* out; float* lse;
  const float* dout; float* dqkv; float* dbias_partial;
  int ldq, ldo, lddo, lddq;
  int B, H, W, C, heads, ws, shift;
  int nwin_y, nwin_x;      // windows per image
  int groups_total;        // total window groups (= ceil(B*nwin/G))
  int windows_total;       // B*nwin; the last group may be partial (its surplus slots recompute window 0 and store nothing)
  int groups_per_block;
  const float* relbias2; int half_windows;   // paired modules: windows >= half_windows use the second bias table
  int split;               // backward MFMA kernel on small grids: two workgroups per window group, one query / key tile each
};

// pixel index (in the un-rolled image) of token (ty,tx) of window (wy,wx)
__device__ __forceinline__ int token_pixel(const AttnParams& p, int b, int wy, int wx, int ty, int tx) {
  int y = wy * p.ws + ty, x = wx * p.ws + tx;
  if (p.shift) { y += p.ws / 2; if (y >= p.H) y -= p.H; x += p.ws / 2; if (x >= p.W) x -= p.W; }
  return (b * p.H + y) * p.W + x;
}
// shift mask: query token (qy,qx) vs key token (ky,kx) of window (wy,wx)
__device__ __forceinline__ bool masked(const AttnParams& p, int wy, int wx, int qy, int qx, int ky, int kx) {
  if (!p.shift) return false;
  const int s = p.ws - p.ws / 2;
  bool m = false;
  if (wy == p.nwin_y - 1) m = m || ((qy < s) != (ky < s));
  if (wx == p.nwin_x - 1) m = m || ((qx < s) != (kx < s));
  return m;
}

// T = tokens per window (64 or 16), HD = head dim. One wave per workgroup; G = 64/T window slots.
// Two sweeps over the keys (softmax statistics online, then P.V) instead of a T-long score
// array per lane: keeps the kernel in registers (no scratch) at the price of computing q.k twice.
template <int T, int HD>
__global__ __launch_bounds__(64) void winattn_fwd_kernel(const AttnParams p) {
  // K / V rows are read as BROADCASTS (every lane of a window reads the same key row), so rows are packed (LDK = HD,
  // 16-B aligned): one ds_read_b128 feeds 4 FMAs — the b32-per-FMA version was LDS-issue-bound (220 us at 128x128).
  constexpr int G = 64 / T, WS = (T == 64) ? 8 : 4, LDK = HD, NBW = 2 * WS - 1;
  __shared__ __attribute__((aligned(16))) float Ks[G][T][LDK], Vs[G][T][LDK];
  __shared__ float Bias[NBW * NBW];
  const int lane = threadIdx.x, g = lane / T, t = lane % T, ty = t / WS, tx = t % WS;
  const int head = blockIdx.y;
  const float scale = rsqrtf((float)HD);
  int cur_side = -1;
  const int nwin = p.nwin_y * p.nwin_x;
  const float* bias_t = Bias + (ty + WS - 1) * NBW + (tx + WS - 1);   // Bias[(ty-ky+WS-1)*NBW + tx-kx+WS-1] = bias_t[-(ky*NBW+kx)]
  const int sgap = WS - WS / 2;
  const bool q_lo_y = ty < sgap, q_lo_x = tx < sgap;
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = blockIdx.x * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const int side = (p.relbias2 != nullptr && grp * G >= p.half_windows) ? 1 : 0;   // block-uniform
    if (side != cur_side) {
      __syncthreads();
      const float* rb = (side ? p.relbias2 : p.relbias) + head * NBW * NBW;
      for (int i = lane; i < NBW * NBW; i += 64) Bias[i] = rb[i];
      cur_side = side;
    }
    const bool wvalid = grp * G + g < p.windows_total;
    const int widx = wvalid ? grp * G + g : 0;            // global window index (b, wy, wx)
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    const bool edge_y = p.shift && wy == p.nwin_y - 1, edge_x = p.shift && wx == p.nwin_x - 1;
    f32x4 q[HD / 4];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      q[c / 4] = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c) * scale;
      *reinterpret_cast<f32x4*>(&Ks[g][t][c]) = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
      *reinterpret_cast<f32x4*>(&Vs[g][t][c]) = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
    }
    __syncthreads();
    // keys as (ky, kx): kx is unrolled (compile-time), ky is a scalar loop counter -> the bias offset and the mask's key
    // side are scalar / immediate, and the register footprint stays that of one key row (full unrolling of all T keys
    // lets the scheduler hoist every LDS read: 250 VGPRs)
    auto score = [&](int ky, int kx) -> float {
      const int j = ky * WS + kx;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(&Ks[g][j][c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaf(q[c / 4][e], kv[e], a);
      }
      a += bias_t[-(ky * NBW + kx)];
      const bool m = (edge_y && (q_lo_y != (ky < sgap))) || (edge_x && (q_lo_x != (kx < sgap)));
      return m ? -INFINITY : a;
    };
    float mx = -INFINITY;
#pragma unroll 1
    for (int ky = 0; ky < WS; ++ky)
#pragma unroll
      for (int kx = 0; kx < WS; ++kx) mx = fmaxf(mx, score(ky, kx));
    float l = 0.f;
    f32x4 o[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < WS; ++ky)
#pragma unroll
      for (int kx = 0; kx < WS; ++kx) {
        const float e = exp_fast(score(ky, kx) - mx);
        l += e;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[g][ky * WS + kx][c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[c / 4][k] = fmaf(e, vv[k], o[c / 4][k]);
        }
      }
    const float inv = 1.f / l;
    if (wvalid) {
      float* op = p.out + (size_t)pix * p.ldo + head * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(op + c) = o[c / 4] * inv;
      if (p.lse) p.lse[(size_t)pix * p.heads + head] = mx + logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA form of the 8x8-window kernels (T = 64 tokens): the two contractions of a (window, head) run on
// v_mfma_f32_32x32x2_f32 instead of 64 x HD scalar FMA chains per lane (the VALU kernels above sit at their instruction
// bound: 11-38 k cycles per window-head).  One wave per (window, head) as before.
//   MFMA operand / result layout (32x32x2): A lane l -> A[m = l&31][k = l>>5]; B lane l -> B[k = l>>5][n = l&31];
//   D lane l, register r -> D[m = (r&3) + 8(r>>2) + 4(l>>5)][n = l&31].
//   Scores are produced TRANSPOSED, S'[j][i] = sum_c K[j][c] Q[i][c] (A = K rows, B = Q rows): lane (li, h) then holds, for
//   the queries i = li and 32 + li, the keys j = 32 tj + (r&3) + 8(r>>2) + 4h — a query's row is split over the two lanes
//   li and li + 32 (one cross-lane exchange for max / sum), and the normalised probabilities ARE the A operand of
//   O = P V (A[m = i][k = j]): step (tj, r) contracts the key pair {j_a, j_a + 4}, j_a = 32 tj + (r&3) + 8(r>>2), whose
//   B operand is V[j_a + 4h][c] straight from LDS.  No transposes, no score buffer.
//   The shift mask is compile-time / lane-constant in this layout: y-mask <=> query tile != key tile, x-mask <=>
//   ((li & 7) < 4) != (h == 0).
// The products whose N is the head dimension (O = P V; dQ = dS K, dV = P^T dO, dK = dS^T Q) have 8 or 16 useful columns: on 32x32x2 tiles
// 24 or 16 of the 32 columns are padding.  For head_dim <= 16 they run on v_mfma_f32_16x16x1_4b_f32 instead — FOUR independent 16x16 outer
// products per instruction (block b = lane / 16: A_b[i = lane % 16], B_b[n = lane % 16]; D_b[i][n] in VGPR 4 b + (i % 4) of lane
// 16 (i / 4) + n; layout checked on the hardware by tools/probes/mfma4b_probe.hip), half the cycles of a 32x32x2.  With the A operand in
// the S' register layout (lane (li, h): row 16 (b & 1) + i of the tile, contraction index offset 4 h) the SAME two operand registers
// serve: block b accumulates rows 16 (b & 1) .. + 15 over the contraction indices of half-wave b >> 1, so a row's result is
// D_b + D_{b+2} — one add at the end.
template <bool B4>
__device__ __forceinline__ f32x16 mfma_nhd(float a, float b, f32x16 c) {
  if constexpr (B4) return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// rows 32 tile .. 32 tile + 31 of an [T][HD] result held by mfma_nhd accumulators -> the LDS staging image (columns < HD only)
template <int HD, int LDQ, bool B4>
__device__ __forceinline__ void stage_rows(float (*St)[LDQ], const f32x16& acc, int tile, int lane, float mul) {
  if constexpr (B4) {
    const int n = lane & 15, g = lane >> 4;
    if (n < HD) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int v = 0; v < 4; ++v) St[32 * tile + 16 * half + 4 * g + v][n] = (acc[4 * half + v] + acc[8 + 4 * half + v]) * mul;
    }
  } else {
    const int li = lane & 31, h = lane >> 5;
    if (li < HD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) St[32 * tile + (r & 3) + 8 * (r >> 2) + 4 * h][li] = acc[r] * mul;
    }
  }
}

// NH > 1 (round 5): NH waves per workgroup, one HEAD each — the NH = 128 B / (4 HD) heads whose q (k, v) shares of a token lie in ONE 128-byte
// line.  With one head per single-wave workgroup every wave read 32 B (head_dim 8) of each of its 192 lines and the other three heads' workgroups
// — other times, other XCDs — fetched the same lines again: 219.8 MB at the L2s' memory side for the 100.7-MB qkv tensor of an 8 x 128 x 128 map,
// all in 128-byte requests (tools/pmc_attn_traffic.sh).  Running the four heads back to back in one wave made it worse (390 MB: an XCD's 4 MB
// L2 does not hold 256 resident workgroups' 24 KB across a head's compute).  Here the workgroup's NH x 64 threads fetch each line ONCE, whole
// (8 consecutive threads = one line), deposit every head's share in that head's LDS images, and store the result the same way.
// Everything between the load and the store is the single-wave kernel, per wave: same arithmetic, same bits.
template <int HD, bool B4, int NH = 1>
__global__ __launch_bounds__(64 * NH, NH == 1 ? 4 : 3) void winattn_fwd_mfma_kernel(const AttnParams p) {
  constexpr int T = 64, WS = 8, LDQ = HD + 4, NBW = 2 * WS - 1, NB = NBW * NBW;
  __shared__ __attribute__((aligned(16))) float QsA[NH][T][LDQ], KsA[NH][T][LDQ], VsA[NH][T][LDQ];
  __shared__ float BiasA[NH][NB];
  const int wave = NH == 1 ? 0 : (int)(threadIdx.x >> 6);
  float (*Qs)[LDQ] = QsA[wave];
  float (*Ks)[LDQ] = KsA[wave];
  float (*Vs)[LDQ] = VsA[wave];
  float* Bias = BiasA[wave];
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int ty = lane >> 3, tx = lane & 7;          // the token whose rows this lane loads / stores
  const int head = blockIdx.y * NH + wave;
  const float scale = rsqrtf((float)HD);
  const int nwin = p.nwin_y * p.nwin_x;
  int cur_side = -1;
  // bias index of element (tj, ti, r): (iy - jy + 7) * 15 + (ix - jx + 7), iy = 4 ti + (li >> 3), ix = li & 7,
  // jy = 4 tj + (r >> 2), jx = (r & 3) + 4 h
  const float* bias_l = Bias + ((li >> 3) + WS - 1) * NBW + ((li & 7) - 4 * h + WS - 1);
  const bool xmask_l = ((li & 7) < 4) != (h == 0);
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = blockIdx.x * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const int side = (p.relbias2 != nullptr && grp >= p.half_windows) ? 1 : 0;   // block-uniform
    if (side != cur_side) {
      __syncthreads();
      const float* rb = (side ? p.relbias2 : p.relbias) + head * NB;
      for (int i = lane; i < NB; i += 64) Bias[i] = rb[i];
      cur_side = side;
    }
    const int widx = grp;
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    const bool edge_y = p.shift && wy == p.nwin_y - 1, edge_x = p.shift && wx == p.nwin_x - 1;
    __syncthreads();
    if constexpr (NH == 1) {
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        *reinterpret_cast<f32x4*>(&Qs[lane][c]) = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c) * scale;
        *reinterpret_cast<f32x4*>(&Ks[lane][c]) = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
        *reinterpret_cast<f32x4*>(&Vs[lane][c]) = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
      }
    } else {
      // whole lines: thread t takes 16-B chunk t & 7 of the line of token (t >> 3) + k * (8 NH); chunk ch belongs to head ch / (HD / 4)
      constexpr int CPH = HD / 4;                                   // chunks per head
      const int t = threadIdx.x, ch = t & 7, hl = ch / CPH, cc = (ch % CPH) * 4;
      const int line0 = blockIdx.y * NH * HD;                       // first channel of this workgroup's heads
#pragma unroll
      for (int k = 0; k < 8 / NH; ++k) {
        const int tok = (t >> 3) + k * 8 * NH;
        const size_t rw = (size_t)token_pixel(p, b, wy, wx, tok >> 3, tok & 7) * p.ldq + line0 + ch * 4;
        *reinterpret_cast<f32x4*>(&QsA[hl][tok][cc]) = *reinterpret_cast<const f32x4*>(p.qkv + rw) * scale;
        *reinterpret_cast<f32x4*>(&KsA[hl][tok][cc]) = *reinterpret_cast<const f32x4*>(p.qkv + rw + p.C);
        *reinterpret_cast<f32x4*>(&VsA[hl][tok][cc]) = *reinterpret_cast<const f32x4*>(p.qkv + rw + 2 * p.C);
      }
    }
    __syncthreads();
    // ---- S'[j][i]: tiles s[tj][ti] ----
    f32x16 s[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bq = 0; bq < 2; ++bq)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[a][bq][r] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < HD; c0 += 8) {   // one b128 per operand row feeds 4 MFMAs: step ss contracts c in {c0+ss, c0+4+ss}
      f32x4 ka[2], qb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ka[t] = *reinterpret_cast<const f32x4*>(&Ks[32 * t + li][c0 + 4 * h]);
        qb[t] = *reinterpret_cast<const f32x4*>(&Qs[32 * t + li][c0 + 4 * h]);
      }
#pragma unroll
      for (int ss = 0; ss < 4; ++ss)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) s[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[tj][ss], qb[ti][ss], s[tj][ti], 0, 0, 0);
    }
    // ---- bias, mask, softmax over the keys of each query (two queries per lane: i = 32 ti + li) ----
    float lse_out = 0.f;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      float mx = -INFINITY;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const bool msk = (edge_y && ti != tj) || (edge_x && xmask_l);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = s[tj][ti][r] + bias_l[(4 * ti - 4 * tj - (r >> 2)) * NBW - (r & 3)];
          s[tj][ti][r] = msk ? -INFINITY : v;
          mx = fmaxf(mx, s[tj][ti][r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = exp_fast(s[tj][ti][r] - mx);
          s[tj][ti][r] = e;
          l += e;
        }
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.f / l;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tj][ti][r] *= inv;
      if (ti == h) lse_out = mx + logf(l);            // lane = li + 32 h stores the statistics of query i = lane
    }
    // ---- O[i][c] = sum_j P[i][j] V[j][c] ----
    f32x16 o[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[ti][r] = 0.f;
    const float* vcol = &Vs[4 * h][li & (HD - 1)];     // lanes with li >= HD re-read a valid column; their results are not stored
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float vb = vcol[(32 * tj + (r & 3) + 8 * (r >> 2)) * LDQ];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) o[ti] = mfma_nhd<B4>(s[tj][ti][r], vb, o[ti]);
      }
    // ---- rows out: stage through LDS (Q's image is dead), then one row per lane ----
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) stage_rows<HD, LDQ, B4>(Qs, o[ti], ti, lane, 1.f);
    __syncthreads();
    if constexpr (NH == 1) {
      float* op = p.out + (size_t)pix * p.ldo + head * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(op + c) = *reinterpret_cast<const f32x4*>(&Qs[lane][c]);
    } else {   // the NH heads' results of a token are one 128-byte line of `out`: stored whole
      constexpr int CPH = HD / 4;
      const int t = threadIdx.x, ch = t & 7, hl = ch / CPH, cc = (ch % CPH) * 4;
#pragma unroll
      for (int k = 0; k < 8 / NH; ++k) {
        const int tok = (t >> 3) + k * 8 * NH;
        float* op = p.out + (size_t)token_pixel(p, b, wy, wx, tok >> 3, tok & 7) * p.ldo + blockIdx.y * NH * HD + ch * 4;
        *reinterpret_cast<f32x4*>(op) = *reinterpret_cast<const f32x4*>(&QsA[hl][tok][cc]);
      }
    }
    if (p.lse) p.lse[(size_t)pix * p.heads + head] = lse_out;
  }
}

// Backward. Inputs: qkv, the forward output `out` (for D = dO.O), lse = log-sum-exp per (token, head).
template <int T, int HD>
__global__ __launch_bounds__(64) void winattn_bwd_kernel(const AttnParams p) {
  constexpr int G = 64 / T, WS = (T == 64) ? 8 : 4, LDK = HD, NBW = 2 * WS - 1, NB = NBW * NBW;   // packed rows: b128 broadcasts
  constexpr int UNR = HD == 8 ? 4 : (HD == 16 ? 2 : 1);   // keys per unrolled group (register pressure: each key holds 2*HD operand floats)
  __shared__ __attribute__((aligned(16))) float Qs[G][T][LDK], Ks[G][T][LDK], Vs[G][T][LDK], Ds[G][T][LDK];
  __shared__ float Lse[G][T], Dd[G][T], Bias[NB];
  // relative-bias gradient bins of this workgroup: for a fixed key the 64 query lanes hit 64 DISTINCT bins (bin = q - k), so a
  // plain LDS read-modify-write per key is race-free and its order (windows, then keys) is fixed -> reproducible without a
  // [T][T] dS buffer (16.6 KB per wave, which capped the kernel at 6 waves per CU)
  __shared__ float BinAcc[G][NB];
  const int lane = threadIdx.x, g = lane / T, t = lane % T, ty = t / WS, tx = t % WS;
  const int head = blockIdx.y;
  const float scale = rsqrtf((float)HD);
  {   // (paired modules: the launcher makes sure a workgroup's windows all belong to one module)
    const float* rb = ((p.relbias2 != nullptr && (int)blockIdx.x * p.groups_per_block * G >= p.half_windows) ? p.relbias2 : p.relbias) + head * NB;
    for (int i = lane; i < NB; i += 64) Bias[i] = rb[i];
  }
  for (int i = lane; i < G * NB; i += 64) (&BinAcc[0][0])[i] = 0.f;
  const int nwin = p.nwin_y * p.nwin_x;
  const float* bias_q = Bias + (ty + WS - 1) * NBW + (tx + WS - 1);   // lane = query: Bias[q - k] = bias_q[-(ky*NBW+kx)]
  const float* bias_k = Bias + (WS - 1 - ty) * NBW + (WS - 1 - tx);   // lane = key:   Bias[q - k] = bias_k[+(qy*NBW+qx)]
  // volatile: the read-modify-writes of successive keys alias ACROSS lanes (lane A's bin for key j is lane B's bin for key
  // j+1), which per-thread alias analysis cannot see — keep them in program order
  volatile float* bin_q = &BinAcc[g][(ty + WS - 1) * NBW + (tx + WS - 1)];
  const int sgap = WS - WS / 2;
  const bool lo_y = ty < sgap, lo_x = tx < sgap;
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = blockIdx.x * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const bool wvalid = grp * G + g < p.windows_total;
    const int widx = wvalid ? grp * G + g : 0;
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    const bool edge_y = p.shift && wy == p.nwin_y - 1, edge_x = p.shift && wx == p.nwin_x - 1;
    f32x4 q[HD / 4], d_o[HD / 4];
    float dsum = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c) * scale;
      const f32x4 kv = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dout + (size_t)pix * p.lddo + head * HD + c);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(p.out + (size_t)pix * p.ldo + head * HD + c);
      q[c / 4] = qv; d_o[c / 4] = dv;
#pragma unroll
      for (int e = 0; e < 4; ++e) dsum = fmaf(dv[e], ov[e], dsum);
      *reinterpret_cast<f32x4*>(&Qs[g][t][c]) = qv; *reinterpret_cast<f32x4*>(&Ks[g][t][c]) = kv;
      *reinterpret_cast<f32x4*>(&Vs[g][t][c]) = vv; *reinterpret_cast<f32x4*>(&Ds[g][t][c]) = dv;
    }
    const float lse = p.lse[(size_t)pix * p.heads + head];
    Lse[g][t] = lse; Dd[g][t] = dsum;
    __syncthreads();
    // ---- pass 1: lane = query row -> dQ and this row of dS (for the relative-bias gradient) ----
    f32x4 dq[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) dq[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < WS; ++ky)
#pragma unroll 1
      for (int kx0 = 0; kx0 < WS; kx0 += UNR)
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int kx = kx0 + u, j = ky * WS + kx;
        float a = 0.f, dp = 0.f;
        f32x4 kr[HD / 4];
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          kr[c / 4] = *reinterpret_cast<const f32x4*>(&Ks[g][j][c]);
          const f32x4 vr = *reinterpret_cast<const f32x4*>(&Vs[g][j][c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { a = fmaf(q[c / 4][e], kr[c / 4][e], a); dp = fmaf(d_o[c / 4][e], vr[e], dp); }
        }
        a += bias_q[-(ky * NBW + kx)];
        const bool m = (edge_y && (lo_y != (ky < sgap))) || (edge_x && (lo_x != (kx < sgap)));
        const float pj = m ? 0.f : exp_fast(a - lse);
        const float ds = pj * (dp - dsum);
        bin_q[-(ky * NBW + kx)] = bin_q[-(ky * NBW + kx)] + (wvalid ? ds : 0.f);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) dq[c][e] = fmaf(ds, kr[c][e], dq[c][e]);
      }
    if (wvalid) {
      float* dqp = p.dqkv + (size_t)pix * p.lddq + head * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(dqp + c) = dq[c / 4] * scale;
    }
    // ---- pass 2: lane = key column. dV = sum_i P[i][t] dO[i]; dK = sum_i dS[i][t] q_i ----
    f32x4 kk[HD / 4], vv2[HD / 4], dk[HD / 4], dvv[HD / 4];
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      kk[c] = *reinterpret_cast<const f32x4*>(&Ks[g][t][c * 4]); vv2[c] = *reinterpret_cast<const f32x4*>(&Vs[g][t][c * 4]);
      dk[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int qy = 0; qy < WS; ++qy)
#pragma unroll 1
      for (int qx0 = 0; qx0 < WS; qx0 += UNR)
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int qx = qx0 + u, i = qy * WS + qx;
        float a = 0.f, dpv = 0.f;
        f32x4 qr[HD / 4], dr[HD / 4];
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          qr[c / 4] = *reinterpret_cast<const f32x4*>(&Qs[g][i][c]);
          dr[c / 4] = *reinterpret_cast<const f32x4*>(&Ds[g][i][c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { a = fmaf(qr[c / 4][e], kk[c / 4][e], a); dpv = fmaf(dr[c / 4][e], vv2[c / 4][e], dpv); }
        }
        a += bias_k[qy * NBW + qx];
        const bool m = (edge_y && ((qy < sgap) != lo_y)) || (edge_x && ((qx < sgap) != lo_x));
        const float pij = m ? 0.f : exp_fast(a - Lse[g][i]);
        const float ds = pij * (dpv - Dd[g][i]);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) { dvv[c][e] = fmaf(pij, dr[c][e], dvv[c][e]); dk[c][e] = fmaf(ds, qr[c][e], dk[c][e]); }
      }
    float* dkp = p.dqkv + (size_t)pix * p.lddq + p.C + head * HD;
    float* dvp = p.dqkv + (size_t)pix * p.lddq + 2 * p.C + head * HD;
    if (wvalid)
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        // Qs already carries the hd^-1/2 factor, so dk is complete as is
        *reinterpret_cast<f32x4*>(dkp + c) = dk[c / 4];
        *reinterpret_cast<f32x4*>(dvp + c) = dvv[c / 4];
      }
  }
  // ---- per-workgroup partial of the relative-bias gradient (window slots summed in order) ----
  __syncthreads();
  for (int bin = lane; bin < NB; bin += 64) {
    float sum = 0.f;
    for (int gg = 0; gg < G; ++gg) sum += BinAcc[gg][bin];
    p.dbias_partial[((size_t)blockIdx.x * p.heads + head) * NB + bin] = sum;
  }
}

// MFMA backward for the 8x8 windows (layouts: winattn_fwd_mfma_kernel).  Two passes, as in the VALU kernel:
//   pass 1, lane = query: S' = K Q^T and dP' = V dO^T (transposed, so the per-query statistics are lane-held) ->
//           dS' = P' (dP' - D) is the A operand of dQ = dS K; the relative-bias bins take dS' with a plain read-modify-write
//           per (tile, register) step — inside a half-wave the 32 queries of one key hit 32 distinct bins, the two half-waves
//           (keys j and j + 4) own separate bin arrays, summed at the end: race-free and in a fixed order;
//   pass 2, lane = key:   S = Q K^T and dP = dO V^T -> P and dS are the A operands of dV = P^T dO and dK = dS^T Q.
template <int HD, bool B4>
__global__ __launch_bounds__(64, 2) void winattn_bwd_mfma_kernel(const AttnParams p) {
  constexpr int T = 64, WS = 8, LDQ = HD + 4, NBW = 2 * WS - 1, NB = NBW * NBW;
  __shared__ __attribute__((aligned(16))) float Qs[T][LDQ], Ks[T][LDQ], Vs[T][LDQ], Ds[T][LDQ], St[T][LDQ], St2[T][LDQ];
  __shared__ float Lse[T], Dd[T], Bias[NB];
  // Relative-bias gradient.  In the S' register layout the bin of element (tile pair (ti, tj), register r) of a lane is
  //   off_q(lane) + (4 (ti - tj)) * NBW - (r >> 2) * NBW - (r & 3)
  // — the SAME for every window this workgroup visits.  So dS is accumulated in REGISTERS per (ti - tj, r) (3 x 16 per lane) over all
  // the windows, and the bins in LDS are touched once per workgroup at the end (48 read-modify-write steps; inside a half-wave the 32
  // lanes of a step hit 32 distinct bins, the two half-waves own separate arrays).  Before (r2, early r3) every element of every window
  // was an LDS read-modify-write: one dependent chain per query tile (later 2-4 interleaved ones) that was longer than the MFMAs beside it.
  // (The bins live in the dV staging image, which is dead by then: with their own 1.8 KB the head_dim-8 kernel was 1 KB over the 20 KB that
  //  lets eight workgroups share a CU — two waves per SIMD, one round for the 2 048 workgroups of a 128x128 map.)
  static_assert(2 * NB <= T * LDQ, "the relative-bias bins must fit the dV staging image");
  float* const BinAcc = &St2[0][0];   // [2][NB]
  float binreg[3][16];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) binreg[d][r] = 0.f;
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  const int ty = lane >> 3, tx = lane & 7;
  const int head = blockIdx.y;
  const float scale = rsqrtf((float)HD);
  const int nwin = p.nwin_y * p.nwin_x;
  // Small grids (the 16x16 latent maps: 512 window-heads for 256 CUs) are latency-bound single-wave workgroups: there a window group
  // is shared by TWO workgroups — `half` does query tile `half` in pass 1 (its 32 rows of dQ) and key tile `half` in pass 2 (its 32 rows
  // of dK / dV) and contributes its own partial row of the relative-bias gradient.  Both load the whole window.
  const int bxi = p.split ? (int)blockIdx.x >> 1 : (int)blockIdx.x, half = p.split ? (int)blockIdx.x & 1 : -1;
  {   // (paired modules: the launcher makes sure a workgroup's windows all belong to one module)
    const float* rb = ((p.relbias2 != nullptr && bxi * p.groups_per_block >= p.half_windows) ? p.relbias2 : p.relbias) + head * NB;
    for (int i = lane; i < NB; i += 64) Bias[i] = rb[i];
  }
  // pass 1 (lane = query i = 32 ti + li; register r of key tile tj = key 32 tj + (r&3) + 8(r>>2) + 4h)
  const int off_q = ((li >> 3) + WS - 1) * NBW + ((li & 7) - 4 * h + WS - 1);
  const float* bias_q = Bias + off_q;
  // LDS byte address of this lane's bin in copy 0.  The updates go through ds_read_b32 / ds_write_b32 written out by hand: hipcc waits
  // for EVERY volatile LDS access before issuing the next one, which turns four interleaved chains back into one.
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned bin_q = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(&St2[0][0] + h * NB + off_q);
#else
  const unsigned bin_q = 0;
#endif
  // pass 2 (lane = key j = 32 tj + li; register r of query tile ti = query 32 ti + (r&3) + 8(r>>2) + 4h)
  const float* bias_k = Bias + (WS - 1 - (li >> 3)) * NBW + (WS - 1 - (li & 7) + 4 * h);
  const bool xmask_l = ((li & 7) < 4) != (h == 0);
  const int colc = li & (HD - 1);               // lanes with li >= HD re-read a valid column of the B operand; their results are not stored
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = bxi * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const int widx = grp;
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    const bool edge_y = p.shift && wy == p.nwin_y - 1, edge_x = p.shift && wx == p.nwin_x - 1;
    float dsum = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c) * scale;
      const f32x4 kv = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dout + (size_t)pix * p.lddo + head * HD + c);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(p.out + (size_t)pix * p.ldo + head * HD + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) dsum = fmaf(dv[e], ov[e], dsum);
      *reinterpret_cast<f32x4*>(&Qs[lane][c]) = qv; *reinterpret_cast<f32x4*>(&Ks[lane][c]) = kv;
      *reinterpret_cast<f32x4*>(&Vs[lane][c]) = vv; *reinterpret_cast<f32x4*>(&Ds[lane][c]) = dv;
    }
    Lse[lane] = p.lse[(size_t)pix * p.heads + head];
    Dd[lane] = dsum;
    __syncthreads();

    // ---------------- pass 1: dQ and the bias bins, one (query tile, key tile) pair at a time ----------------
    // (one pair's S' and dP' live at a time: the saved log-sum-exp makes the key tiles independent.  Holding both key tiles — and,
    //  in pass 2, both tiles' dK / dV until the end — took 316-400 registers: one wave per SIMD, every LDS / exp latency exposed.)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {   // (unrolled: binreg is indexed by ti - tj)
      if (half >= 0 && ti != half) continue;
      const float lse_i = Lse[32 * ti + li], dd_i = Dd[32 * ti + li];
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        f32x16 sp, dp;                // S'[j][i], dP'[j][i] for the queries i = 32 ti + li and the keys of tile tj
#pragma unroll
        for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int c0 = 0; c0 < HD; c0 += 8) {
          const f32x4 qb = *reinterpret_cast<const f32x4*>(&Qs[32 * ti + li][c0 + 4 * h]);
          const f32x4 db = *reinterpret_cast<const f32x4*>(&Ds[32 * ti + li][c0 + 4 * h]);
          const f32x4 ka = *reinterpret_cast<const f32x4*>(&Ks[32 * tj + li][c0 + 4 * h]);
          const f32x4 va = *reinterpret_cast<const f32x4*>(&Vs[32 * tj + li][c0 + 4 * h]);
#pragma unroll
          for (int ss = 0; ss < 4; ++ss) {
            sp = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[ss], qb[ss], sp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[ss], db[ss], dp, 0, 0, 0);
          }
        }
        const bool msk = (edge_y && ti != tj) || (edge_x && xmask_l);
        const int boff = (4 * ti - 4 * tj) * NBW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int bo = boff - (r >> 2) * NBW - (r & 3);
          const float a = sp[r] + bias_q[bo];
          const float pj = msk ? 0.f : exp_fast(a - lse_i);
          const float ds = pj * (dp[r] - dd_i);
          binreg[ti - tj + 1][r] += ds;
          const float kb = Ks[32 * tj + (r & 3) + 8 * (r >> 2) + 4 * h][colc];
          dq = mfma_nhd<B4>(ds, kb, dq);
          // (a fence per 4 registers: left alone the scheduler hoists all 16 steps' LDS reads and exps to the top of the pair, 380+
          //  live registers — and with the register budget of two waves per SIMD it spills instead of hoisting less)
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      stage_rows<HD, LDQ, B4>(St, dq, ti, lane, scale);
    }
    __syncthreads();
    {
      float* dqp = p.dqkv + (size_t)pix * p.lddq + head * HD;
      if (half < 0 || (lane >> 5) == half) {
#pragma unroll
        for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(dqp + c) = *reinterpret_cast<const f32x4*>(&St[lane][c]);
      }
    }

    // ---------------- pass 2: dK and dV, one (key tile, query tile) pair at a time ----------------
    __syncthreads();   // (the dQ rows have been read out of St: this pass stages dK there and dV in St2)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      if (half >= 0 && tj != half) continue;
      f32x16 dkt, dvt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        f32x16 s2, dp2;               // S[i][j], dP[i][j] for the keys j = 32 tj + li and the queries of tile ti
#pragma unroll
        for (int r = 0; r < 16; ++r) { s2[r] = 0.f; dp2[r] = 0.f; }
#pragma unroll
        for (int c0 = 0; c0 < HD; c0 += 8) {
          const f32x4 kb = *reinterpret_cast<const f32x4*>(&Ks[32 * tj + li][c0 + 4 * h]);
          const f32x4 vb = *reinterpret_cast<const f32x4*>(&Vs[32 * tj + li][c0 + 4 * h]);
          const f32x4 qa = *reinterpret_cast<const f32x4*>(&Qs[32 * ti + li][c0 + 4 * h]);
          const f32x4 da = *reinterpret_cast<const f32x4*>(&Ds[32 * ti + li][c0 + 4 * h]);
#pragma unroll
          for (int ss = 0; ss < 4; ++ss) {
            s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ss], kb[ss], s2, 0, 0, 0);
            dp2 = __builtin_amdgcn_mfma_f32_32x32x2f32(da[ss], vb[ss], dp2, 0, 0, 0);
          }
        }
        const bool msk = (edge_y && ti != tj) || (edge_x && xmask_l);
        const int boff = (4 * ti - 4 * tj) * NBW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int iq = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h;    // the query of this register (half-wave uniform)
          const float a = s2[r] + bias_k[boff + (r >> 2) * NBW + (r & 3)];
          const float pij = msk ? 0.f : exp_fast(a - Lse[iq]);
          const float ds = pij * (dp2[r] - Dd[iq]);
          const float dob = Ds[iq][colc], qb = Qs[iq][colc];
          dvt = mfma_nhd<B4>(pij, dob, dvt);
          dkt = mfma_nhd<B4>(ds, qb, dkt);
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      stage_rows<HD, LDQ, B4>(St, dkt, tj, lane, 1.f);
      stage_rows<HD, LDQ, B4>(St2, dvt, tj, lane, 1.f);
    }
    __syncthreads();
    {
      float* dst = p.dqkv + (size_t)pix * p.lddq + p.C + head * HD;
      if (half < 0 || (lane >> 5) == half) {
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
          *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(&St[lane][c]);
          *reinterpret_cast<f32x4*>(dst + p.C + c) = *reinterpret_cast<const f32x4*>(&St2[lane][c]);
        }
      }
    }
  }
  // ---- per-workgroup partial of the relative-bias gradient: registers -> bins (fixed order), then the two half-wave bin sets summed ----
  __syncthreads();
  for (int i = lane; i < 2 * NB; i += 64) BinAcc[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bo = (4 * (d - 1)) * NBW - (r >> 2) * NBW - (r & 3);
      float old;
      // (hand-written: the steps of different lanes alias — lane A's bin of step k is lane B's of step k + 1 — which the compiler cannot
      //  know; LDS executes a wave's accesses in issue order, so read -> wait -> add -> write per step is race-free)
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(bin_q + (unsigned)(bo * 4)) : "memory");
      const float upd = old + binreg[d][r];
      asm volatile("ds_write_b32 %0, %1" ::"v"(bin_q + (unsigned)(bo * 4)), "v"(upd) : "memory");
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int bin = lane; bin < NB; bin += 64)
    p.dbias_partial[((size_t)blockIdx.x * p.heads + head) * NB + bin] = BinAcc[bin] + BinAcc[NB + bin];
}

// out[i] (+)= sum_b partial[b][i]; 32 columns x 8 interleaved block groups per workgroup, combined in a fixed tree
__global__ __launch_bounds__(256) void dbias_reduce_kernel(const float* __restrict__ partial, int nblocks, int n, float* out, int accumulate) {
  __shared__ float sm[8][32];
  const int tx = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + tx;
  float s = 0.f;
  if (i < n)
    for (int b = g; b < nblocks; b += 8) s += partial[(size_t)b * n + i];
  sm[g][tx] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    const float t = ((sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx])) + ((sm[4][tx] + sm[5][tx]) + (sm[6][tx] + sm[7][tx]));
    out[i] = (accumulate ? out[i] : 0.f) + t;
  }
}

int check_geom(const char* who, int B, int H, int W, int C, int heads, int ws, int ld1, int ld3) {
  CLC_CHECK(ws == 8 || ws == 4, "%s: window size must be 8 or 4 (got %d)", who, ws);
  CLC_CHECK(H % ws == 0 && W % ws == 0 && H > ws && W > ws, "%s: %dx%d tokens not a multiple of / larger than the window %d", who, H, W, ws);
  CLC_CHECK(heads > 0 && C % heads == 0, "%s: C %% heads", who);
  const int hd = C / heads;
  CLC_CHECK(hd == 8 || hd == 16 || hd == 32, "%s: head_dim must be 8/16/32 (got %d)", who, hd);
  CLC_CHECK(ld3 >= 3 * C && ld1 >= C && ld3 % 4 == 0 && ld1 % 4 == 0, "%s: bad leading dims", who);
  CLC_CHECK((long)B * H * W < (1l << 31), "%s: too many tokens", who);
  return 0;
}

static bool attn_use_mfma() {
  static const int use_mfma = getenv("CLC_ATTN_MFMA") ? atoi(getenv("CLC_ATTN_MFMA")) : 1;   // 0: VALU kernels (A/B knob)
  return use_mfma != 0;
}
void fill(AttnParams& p, int B, int H, int W, int C, int heads, int ws, int shift, int target_blocks, bool paired = false, bool bwd = false) {
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.ws = ws; p.shift = shift;
  p.nwin_y = H / ws; p.nwin_x = W / ws;
  const int T = ws * ws, G = 64 / T;
  p.windows_total = B * p.nwin_y * p.nwin_x;
  p.groups_total = (p.windows_total + G - 1) / G;
  int per = (p.groups_total * heads + target_blocks - 1) / target_blocks;
  if (per < 1) per = 1;
  p.half_windows = 0;
  if (paired) {   // a workgroup's windows must all belong to one module: groups per block divides the groups of a half
    p.half_windows = p.windows_total / 2;
    const int half_groups = p.half_windows / G;
    while (half_groups % per) --per;
  }
  p.groups_per_block = per;
  // backward on a grid that leaves most wave slots empty: two workgroups per window group (winattn_bwd_mfma_kernel)
  const int nbx = (p.groups_total + per - 1) / per;
  p.split = (bwd && ws == 8 && attn_use_mfma() && clc_tuning[CLC_TUNE_ATTN_SPLIT] && (long)nbx * heads <= 1024) ? 1 : 0;
}

#define DISPATCH(KERNEL, T_, hd, grid, p, st)                                                          \
  do {                                                                                                 \
    if (hd == 8) hipLaunchKernelGGL((KERNEL<T_, 8>), grid, dim3(64), 0, st, p);                         \
    else if (hd == 16) hipLaunchKernelGGL((KERNEL<T_, 16>), grid, dim3(64), 0, st, p);                  \
    else hipLaunchKernelGGL((KERNEL<T_, 32>), grid, dim3(64), 0, st, p);                                \
  } while (0)

}  // namespace

static int bwd_blocks_x(int B, int H, int W, int heads, int ws, bool paired) {
  AttnParams p{};
  fill(p, B, H, W, heads /*C unused*/, heads, ws, 0, 2048, paired, true);
  return ((p.groups_total + p.groups_per_block - 1) / p.groups_per_block) * (p.split ? 2 : 1);
}

extern "C" int clc_winattn_bwd_blocks(int B, int H, int W, int heads, int ws, int paired) { return bwd_blocks_x(B, H, W, heads, ws, paired != 0); }
extern "C" size_t clc_winattn_bwd_workspace_bytes(int B, int H, int W, int heads, int ws) {
  const int nb = (2 * ws - 1) * (2 * ws - 1);
  const int bx = bwd_blocks_x(B, H, W, heads, ws, false), bp = (B % 2 == 0) ? bwd_blocks_x(B, H, W, heads, ws, true) : 0;
  return (size_t)(bx > bp ? bx : bp) * heads * nb * sizeof(float);
}

static int pair_ok(const char* who, int B, int H, int W, int ws) {
  const int T = ws * ws, G = 64 / T;
  CLC_CHECK(B % 2 == 0 && ((B / 2) * (H / ws) * (W / ws)) % G == 0, "%s: the paired form needs an even batch whose half holds whole window groups", who);
  return 0;
}

static int winattn_fwd_impl(const float* qkv, int ldq, const float* relbias, const float* relbias2, float* out, int ldo, float* lse, int B, int H,
                            int W, int C, int heads, int ws, int shift, clc_stream_t stream) {
  CLC_CHECK(qkv && relbias && out, "clc_winattn_fwd: null pointer");
  if (check_geom("clc_winattn_fwd", B, H, W, C, heads, ws, ldo, ldq)) return -1;
  CLC_CHECK(aligned16(qkv) && aligned16(out), "clc_winattn_fwd: unaligned");
  if (relbias2 && pair_ok("clc_winattn_fwd_pair", B, H, W, ws)) return -1;
  AttnParams p{};
  p.qkv = qkv; p.relbias = relbias; p.out = out; p.lse = lse; p.ldq = ldq; p.ldo = ldo;
  const int hd = C / heads;
  static const int use_mfma = getenv("CLC_ATTN_MFMA") ? atoi(getenv("CLC_ATTN_MFMA")) : 1;   // 0: VALU kernels (A/B knob)
  static const int nh_on = getenv("CLC_ATTN_NH") ? atoi(getenv("CLC_ATTN_NH")) : 1;          // 0: one head per single-wave workgroup (A/B knob)
  // forward MFMA kernel: the heads that share a 128-byte line of a token's row as the waves of ONE workgroup (whole-line loads and stores)
  const int nh = (ws == 8 && use_mfma && nh_on && (hd == 8 || hd == 16) && heads % (32 / hd) == 0 && ldq % 4 == 0 && ldo % 4 == 0 && (hd * (32 / hd)) % 32 == 0) ? 32 / hd : 1;
  fill(p, B, H, W, C, heads / nh, ws, shift, nh > 1 ? 4096 : 8192, relbias2 != nullptr);
  p.heads = heads;
  p.relbias2 = relbias2;
  dim3 grid((p.groups_total + p.groups_per_block - 1) / p.groups_per_block, heads / nh);
  if (nh > 1) {
    const int m4 = clc_tuning[CLC_TUNE_ATTN_4B];
    if (hd == 8) { if (m4 & 4) hipLaunchKernelGGL((winattn_fwd_mfma_kernel<8, true, 4>), grid, dim3(256), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_fwd_mfma_kernel<8, false, 4>), grid, dim3(256), 0, (hipStream_t)stream, p); }
    else { if (m4 & 1) hipLaunchKernelGGL((winattn_fwd_mfma_kernel<16, true, 2>), grid, dim3(128), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_fwd_mfma_kernel<16, false, 2>), grid, dim3(128), 0, (hipStream_t)stream, p); }
  } else if (ws == 8 && use_mfma) {
    // head_dim <= 16: the N = head_dim products on 4-block 16x16x1 MFMAs (key 16, bit mask: 1 = head_dim 16, 2 = head_dim-8 backward,
    // 4 = head_dim-8 forward).  The head_dim-8 FORWARD (the analysis / synthesis transforms) is off by default: another summation order
    // moves y by ~4e-6, and on the parity sample one hyper-latent sits that close to .5 — its flip costs 9e-4 bpp against the oracle
    // (bar: 1e-4).  Both orders are equally exact (3e-7 of fp64); the default keeps the order the parity numbers were taken with.
    const int m4 = clc_tuning[CLC_TUNE_ATTN_4B];
    if (hd == 8) { if (m4 & 4) hipLaunchKernelGGL((winattn_fwd_mfma_kernel<8, true>), grid, dim3(64), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_fwd_mfma_kernel<8, false>), grid, dim3(64), 0, (hipStream_t)stream, p); }
    else if (hd == 16) { if (m4 & 1) hipLaunchKernelGGL((winattn_fwd_mfma_kernel<16, true>), grid, dim3(64), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_fwd_mfma_kernel<16, false>), grid, dim3(64), 0, (hipStream_t)stream, p); }
    else hipLaunchKernelGGL((winattn_fwd_mfma_kernel<32, false>), grid, dim3(64), 0, (hipStream_t)stream, p);
  } else if (ws == 8) DISPATCH(winattn_fwd_kernel, 64, hd, grid, p, (hipStream_t)stream);
  else DISPATCH(winattn_fwd_kernel, 16, hd, grid, p, (hipStream_t)stream);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_winattn_fwd(const float* qkv, int ldq, const float* relbias, float* out, int ldo, float* lse, int B, int H, int W,
                               int C, int heads, int ws, int shift, clc_stream_t stream) {
  return winattn_fwd_impl(qkv, ldq, relbias, nullptr, out, ldo, lse, B, H, W, C, heads, ws, shift, stream);
}
extern "C" int clc_winattn_fwd_pair(const float* qkv, int ldq, const float* relbias, const float* relbias2, float* out, int ldo, float* lse,
                                    int B, int H, int W, int C, int heads, int ws, int shift, clc_stream_t stream) {
  CLC_CHECK(relbias2, "clc_winattn_fwd_pair: relbias2 missing");
  return winattn_fwd_impl(qkv, ldq, relbias, relbias2, out, ldo, lse, B, H, W, C, heads, ws, shift, stream);
}

static int winattn_bwd_impl(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* relbias2, const float* out,
                            int ldo, const float* lse, float* dqkv, int lddq, float* drelbias, float* drelbias2, int accumulate, int B, int H,
                            int W, int C, int heads, int ws, int shift, void* wsb, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(dout && qkv && relbias && out && lse && dqkv, "clc_winattn_bwd: null pointer");
  CLC_CHECK(ldo >= C && ldo % 4 == 0 && aligned16(out), "clc_winattn_bwd: bad out");
  if (check_geom("clc_winattn_bwd", B, H, W, C, heads, ws, lddo, ldq)) return -1;
  CLC_CHECK(lddq >= 3 * C && lddq % 4 == 0, "clc_winattn_bwd: bad lddq");
  CLC_CHECK(aligned16(qkv) && aligned16(dout) && aligned16(dqkv), "clc_winattn_bwd: unaligned");
  CLC_CHECK(wsb && ws_bytes >= clc_winattn_bwd_workspace_bytes(B, H, W, heads, ws), "clc_winattn_bwd: workspace too small");
  if (relbias2 && pair_ok("clc_winattn_bwd_pair", B, H, W, ws)) return -1;
  AttnParams p{};
  p.qkv = qkv; p.relbias = relbias; p.dout = dout; p.dqkv = dqkv; p.dbias_partial = (float*)wsb;
  p.out = const_cast<float*>(out); p.lse = const_cast<float*>(lse);
  p.ldq = ldq; p.lddo = lddo; p.lddq = lddq; p.ldo = ldo;
  fill(p, B, H, W, C, heads, ws, shift, 2048, relbias2 != nullptr, true);
  p.relbias2 = relbias2;
  const int nbx = ((p.groups_total + p.groups_per_block - 1) / p.groups_per_block) * (p.split ? 2 : 1);   // (= clc_winattn_bwd_blocks)
  dim3 grid(nbx, heads);
  const int hd = C / heads;
  const bool use_mfma = attn_use_mfma();
  if (ws == 8 && use_mfma) {
    const int m4 = clc_tuning[CLC_TUNE_ATTN_4B];
    if (hd == 8) { if (m4 & 2) hipLaunchKernelGGL((winattn_bwd_mfma_kernel<8, true>), grid, dim3(64), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_bwd_mfma_kernel<8, false>), grid, dim3(64), 0, (hipStream_t)stream, p); }
    else if (hd == 16) { if (m4 & 1) hipLaunchKernelGGL((winattn_bwd_mfma_kernel<16, true>), grid, dim3(64), 0, (hipStream_t)stream, p); else hipLaunchKernelGGL((winattn_bwd_mfma_kernel<16, false>), grid, dim3(64), 0, (hipStream_t)stream, p); }
    else hipLaunchKernelGGL((winattn_bwd_mfma_kernel<32, false>), grid, dim3(64), 0, (hipStream_t)stream, p);
  } else if (ws == 8) DISPATCH(winattn_bwd_kernel, 64, hd, grid, p, (hipStream_t)stream);
  else DISPATCH(winattn_bwd_kernel, 16, hd, grid, p, (hipStream_t)stream);
  CLC_LAUNCH_CHECK();
  if (drelbias == nullptr) return 0;   // partial rows [blocks][n] stay in wsb for clc_partial_reduce_batched
  const int n = heads * (2 * ws - 1) * (2 * ws - 1);
  const int nb1 = relbias2 ? nbx / 2 : nbx;
  hipLaunchKernelGGL(dbias_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)wsb, nb1, n, drelbias, accumulate);
  CLC_LAUNCH_CHECK();
  if (relbias2) {
    CLC_CHECK(drelbias2, "clc_winattn_bwd_pair: drelbias2 missing");
    hipLaunchKernelGGL(dbias_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)wsb + (size_t)nb1 * n, nbx - nb1, n, drelbias2, accumulate);
    CLC_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int clc_winattn_bwd(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* out, int ldo,
                               const float* lse, float* dqkv, int lddq, float* drelbias, int accumulate, int B, int H, int W, int C,
                               int heads, int ws, int shift, void* wsb, size_t ws_bytes, clc_stream_t stream) {
  return winattn_bwd_impl(dout, lddo, qkv, ldq, relbias, nullptr, out, ldo, lse, dqkv, lddq, drelbias, nullptr, accumulate, B, H, W, C, heads, ws, shift,
                          wsb, ws_bytes, stream);
}
extern "C" int clc_winattn_bwd_pair(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* relbias2,
                                    const float* out, int ldo, const float* lse, float* dqkv, int lddq, float* drelbias, float* drelbias2,
                                    int accumulate, int B, int H, int W, int C, int heads, int ws, int shift, void* wsb, size_t ws_bytes,
                                    clc_stream_t stream) {
  CLC_CHECK(relbias2, "clc_winattn_bwd_pair: relbias2 missing");
  return winattn_bwd_impl(dout, lddo, qkv, ldq, relbias, relbias2, out, ldo, lse, dqkv, lddq, drelbias, drelbias2, accumulate, B, H, W, C, heads, ws,
                          shift, wsb, ws_bytes, stream);
}
