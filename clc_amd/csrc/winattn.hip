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

struct AttnParams {
  const float* qkv; const float* relbias; float* out; float* lse;
  const float* dout; float* dqkv; float* dbias_partial;
  int ldq, ldo, lddo, lddq;
  int B, H, W, C, heads, ws, shift;
  int nwin_y, nwin_x;      // windows per image
  int groups_total;        // total window groups (= ceil(B*nwin/G))
  int windows_total;       // B*nwin; the last group may be partial (its surplus slots recompute window 0 and store nothing)
  int groups_per_block;
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
  constexpr int G = 64 / T, WS = (T == 64) ? 8 : 4, LDK = HD + 1;
  __shared__ float Ks[G][T][LDK], Vs[G][T][LDK], Bias[(2 * WS - 1) * (2 * WS - 1)];
  const int lane = threadIdx.x, g = lane / T, t = lane % T, ty = t / WS, tx = t % WS;
  const int head = blockIdx.y;
  const float scale = rsqrtf((float)HD);
  for (int i = lane; i < (2 * WS - 1) * (2 * WS - 1); i += 64) Bias[i] = p.relbias[head * (2 * WS - 1) * (2 * WS - 1) + i];
  const int nwin = p.nwin_y * p.nwin_x;
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = blockIdx.x * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const bool wvalid = grp * G + g < p.windows_total;
    const int widx = wvalid ? grp * G + g : 0;            // global window index (b, wy, wx)
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    float q[HD];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c);
      const f32x4 kv = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q[c + e] = qv[e] * scale; Ks[g][t][c + e] = kv[e]; Vs[g][t][c + e] = vv[e]; }
    }
    __syncthreads();
    auto score = [&](int j) -> float {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) a = fmaf(q[c], Ks[g][j][c], a);
      const int ky = j / WS, kx = j % WS;
      a += Bias[(ty - ky + WS - 1) * (2 * WS - 1) + (tx - kx + WS - 1)];
      return masked(p, wy, wx, ty, tx, ky, kx) ? -INFINITY : a;
    };
    float mx = -INFINITY;
#pragma unroll 4
    for (int j = 0; j < T; ++j) mx = fmaxf(mx, score(j));
    float l = 0.f, o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = 0.f;
#pragma unroll 4
    for (int j = 0; j < T; ++j) {
      const float e = expf(score(j) - mx);
      l += e;
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = fmaf(e, Vs[g][j][c], o[c]);
    }
    const float inv = 1.f / l;
    if (wvalid) {
      float* op = p.out + (size_t)pix * p.ldo + head * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4) *reinterpret_cast<f32x4*>(op + c) = (f32x4){o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
      if (p.lse) p.lse[(size_t)pix * p.heads + head] = mx + logf(l);
    }
  }
}

// Backward. Inputs: qkv, the forward output `out` (for D = dO.O), lse = log-sum-exp per (token, head).
template <int T, int HD>
__global__ __launch_bounds__(64) void winattn_bwd_kernel(const AttnParams p) {
  constexpr int G = 64 / T, WS = (T == 64) ? 8 : 4, LDK = HD + 1, NB = (2 * WS - 1) * (2 * WS - 1);
  __shared__ float Qs[G][T][LDK], Ks[G][T][LDK], Vs[G][T][LDK], Ds[G][T][LDK];
  __shared__ float Lse[G][T], Dd[G][T], Bias[NB];
  __shared__ float AccS[G][T][T + 1];   // sum over this workgroup's windows of dS (rows owned by one lane each)
  const int lane = threadIdx.x, g = lane / T, t = lane % T, ty = t / WS, tx = t % WS;
  const int head = blockIdx.y;
  const float scale = rsqrtf((float)HD);
  for (int i = lane; i < NB; i += 64) Bias[i] = p.relbias[head * NB + i];
  for (int j = 0; j < T; ++j) AccS[g][t][j] = 0.f;
  const int nwin = p.nwin_y * p.nwin_x;
  for (int gi = 0; gi < p.groups_per_block; ++gi) {
    const int grp = blockIdx.x * p.groups_per_block + gi;
    if (grp >= p.groups_total) break;
    const bool wvalid = grp * G + g < p.windows_total;
    const int widx = wvalid ? grp * G + g : 0;
    const int b = widx / nwin, wr = widx - b * nwin, wy = wr / p.nwin_x, wx = wr - wy * p.nwin_x;
    const int pix = token_pixel(p, b, wy, wx, ty, tx);
    const size_t row = (size_t)pix * p.ldq;
    float q[HD], d_o[HD];
    float dsum = 0.f;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(p.qkv + row + head * HD + c);
      const f32x4 kv = *reinterpret_cast<const f32x4*>(p.qkv + row + p.C + head * HD + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.qkv + row + 2 * p.C + head * HD + c);
      const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dout + (size_t)pix * p.lddo + head * HD + c);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(p.out + (size_t)pix * p.ldo + head * HD + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q[c + e] = qv[e] * scale; d_o[c + e] = dv[e];
        dsum = fmaf(dv[e], ov[e], dsum);
        Qs[g][t][c + e] = qv[e] * scale; Ks[g][t][c + e] = kv[e]; Vs[g][t][c + e] = vv[e]; Ds[g][t][c + e] = dv[e];
      }
    }
    const float lse = p.lse[(size_t)pix * p.heads + head];
    Lse[g][t] = lse; Dd[g][t] = dsum;
    __syncthreads();
    // ---- pass 1: lane = query row -> dQ and this row of dS (for the relative-bias gradient) ----
    float dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] = 0.f;
#pragma unroll 4
    for (int j = 0; j < T; ++j) {
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { a = fmaf(q[c], Ks[g][j][c], a); dp = fmaf(d_o[c], Vs[g][j][c], dp); }
      const int ky = j / WS, kx = j % WS;
      a += Bias[(ty - ky + WS - 1) * (2 * WS - 1) + (tx - kx + WS - 1)];
      const float pj = masked(p, wy, wx, ty, tx, ky, kx) ? 0.f : expf(a - lse);
      const float ds = pj * (dp - dsum);
      AccS[g][t][j] += wvalid ? ds : 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, Ks[g][j][c], dq[c]);
    }
    if (wvalid) {
      float* dqp = p.dqkv + (size_t)pix * p.lddq + head * HD;
#pragma unroll
      for (int c = 0; c < HD; c += 4)
        *reinterpret_cast<f32x4*>(dqp + c) = (f32x4){dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale};
    }
    // ---- pass 2: lane = key column. dV = sum_i P[i][t] dO[i]; dK = sum_i dS[i][t] q_i ----
    float kk[HD], vv2[HD], dk[HD], dvv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) { kk[c] = Ks[g][t][c]; vv2[c] = Vs[g][t][c]; dk[c] = 0.f; dvv[c] = 0.f; }
#pragma unroll 2
    for (int i = 0; i < T; ++i) {
      const int qy = i / WS, qx = i % WS;
      float a = 0.f, dpv = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) { a = fmaf(Qs[g][i][c], kk[c], a); dpv = fmaf(Ds[g][i][c], vv2[c], dpv); }
      a += Bias[(qy - ty + WS - 1) * (2 * WS - 1) + (qx - tx + WS - 1)];
      const float pij = masked(p, wy, wx, qy, qx, ty, tx) ? 0.f : expf(a - Lse[g][i]);
      const float ds = pij * (dpv - Dd[g][i]);
#pragma unroll
      for (int c = 0; c < HD; ++c) { dvv[c] = fmaf(pij, Ds[g][i][c], dvv[c]); dk[c] = fmaf(ds, Qs[g][i][c], dk[c]); }
    }
    float* dkp = p.dqkv + (size_t)pix * p.lddq + p.C + head * HD;
    float* dvp = p.dqkv + (size_t)pix * p.lddq + 2 * p.C + head * HD;
    if (wvalid)
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      // Qs already carries the hd^-1/2 factor, so dk is complete as is
      *reinterpret_cast<f32x4*>(dkp + c) = (f32x4){dk[c], dk[c + 1], dk[c + 2], dk[c + 3]};
      *reinterpret_cast<f32x4*>(dvp + c) = (f32x4){dvv[c], dvv[c + 1], dvv[c + 2], dvv[c + 3]};
    }
  }
  // ---- fold sum_windows dS[p][q] into the (2ws-1)^2 relative-position bins, fixed order ----
  __syncthreads();
  for (int bin = lane; bin < NB; bin += 64) {
    const int dy = bin / (2 * WS - 1) - (WS - 1), dx = bin % (2 * WS - 1) - (WS - 1);
    float sum = 0.f;
    for (int gg = 0; gg < G; ++gg)
      for (int qy = 0; qy < WS; ++qy) {
        const int ky = qy - dy;
        if (ky < 0 || ky >= WS) continue;
        for (int qx = 0; qx < WS; ++qx) {
          const int kx = qx - dx;
          if (kx < 0 || kx >= WS) continue;
          sum += AccS[gg][qy * WS + qx][ky * WS + kx];
        }
      }
    p.dbias_partial[((size_t)blockIdx.x * p.heads + head) * NB + bin] = sum;
  }
}

__global__ void dbias_reduce_kernel(const float* __restrict__ partial, int nblocks, int n, float* out, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * n + i];
  out[i] = s;
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

void fill(AttnParams& p, int B, int H, int W, int C, int heads, int ws, int shift, int target_blocks) {
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.ws = ws; p.shift = shift;
  p.nwin_y = H / ws; p.nwin_x = W / ws;
  const int T = ws * ws, G = 64 / T;
  p.windows_total = B * p.nwin_y * p.nwin_x;
  p.groups_total = (p.windows_total + G - 1) / G;
  int per = (p.groups_total * heads + target_blocks - 1) / target_blocks;
  if (per < 1) per = 1;
  p.groups_per_block = per;
}

#define DISPATCH(KERNEL, T_, hd, grid, p, st)                                                          \
  do {                                                                                                 \
    if (hd == 8) hipLaunchKernelGGL((KERNEL<T_, 8>), grid, dim3(64), 0, st, p);                         \
    else if (hd == 16) hipLaunchKernelGGL((KERNEL<T_, 16>), grid, dim3(64), 0, st, p);                  \
    else hipLaunchKernelGGL((KERNEL<T_, 32>), grid, dim3(64), 0, st, p);                                \
  } while (0)

}  // namespace

extern "C" int clc_winattn_fwd(const float* qkv, int ldq, const float* relbias, float* out, int ldo, float* lse, int B, int H, int W,
                               int C, int heads, int ws, int shift, clc_stream_t stream) {
  CLC_CHECK(qkv && relbias && out, "clc_winattn_fwd: null pointer");
  if (check_geom("clc_winattn_fwd", B, H, W, C, heads, ws, ldo, ldq)) return -1;
  CLC_CHECK(aligned16(qkv) && aligned16(out), "clc_winattn_fwd: unaligned");
  AttnParams p{};
  p.qkv = qkv; p.relbias = relbias; p.out = out; p.lse = lse; p.ldq = ldq; p.ldo = ldo;
  fill(p, B, H, W, C, heads, ws, shift, 8192);
  dim3 grid((p.groups_total + p.groups_per_block - 1) / p.groups_per_block, heads);
  const int hd = C / heads;
  if (ws == 8) DISPATCH(winattn_fwd_kernel, 64, hd, grid, p, (hipStream_t)stream);
  else DISPATCH(winattn_fwd_kernel, 16, hd, grid, p, (hipStream_t)stream);
  CLC_LAUNCH_CHECK();
  return 0;
}

static int bwd_blocks_x(int B, int H, int W, int heads, int ws) {
  AttnParams p{};
  fill(p, B, H, W, heads /*C unused*/, heads, ws, 0, 2048);
  return (p.groups_total + p.groups_per_block - 1) / p.groups_per_block;
}

extern "C" size_t clc_winattn_bwd_workspace_bytes(int B, int H, int W, int heads, int ws) {
  const int nb = (2 * ws - 1) * (2 * ws - 1);
  return (size_t)bwd_blocks_x(B, H, W, heads, ws) * heads * nb * sizeof(float);
}

extern "C" int clc_winattn_bwd(const float* dout, int lddo, const float* qkv, int ldq, const float* relbias, const float* out, int ldo,
                               const float* lse, float* dqkv, int lddq, float* drelbias, int accumulate, int B, int H, int W, int C,
                               int heads, int ws, int shift, void* wsb, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(dout && qkv && relbias && out && lse && dqkv && drelbias, "clc_winattn_bwd: null pointer");
  CLC_CHECK(ldo >= C && ldo % 4 == 0 && aligned16(out), "clc_winattn_bwd: bad out");
  if (check_geom("clc_winattn_bwd", B, H, W, C, heads, ws, lddo, ldq)) return -1;
  CLC_CHECK(lddq >= 3 * C && lddq % 4 == 0, "clc_winattn_bwd: bad lddq");
  CLC_CHECK(aligned16(qkv) && aligned16(dout) && aligned16(dqkv), "clc_winattn_bwd: unaligned");
  CLC_CHECK(wsb && ws_bytes >= clc_winattn_bwd_workspace_bytes(B, H, W, heads, ws), "clc_winattn_bwd: workspace too small");
  AttnParams p{};
  p.qkv = qkv; p.relbias = relbias; p.dout = dout; p.dqkv = dqkv; p.dbias_partial = (float*)wsb;
  p.out = const_cast<float*>(out); p.lse = const_cast<float*>(lse);
  p.ldq = ldq; p.lddo = lddo; p.lddq = lddq; p.ldo = ldo;
  fill(p, B, H, W, C, heads, ws, shift, 2048);
  const int nbx = (p.groups_total + p.groups_per_block - 1) / p.groups_per_block;
  dim3 grid(nbx, heads);
  const int hd = C / heads;
  if (ws == 8) DISPATCH(winattn_bwd_kernel, 64, hd, grid, p, (hipStream_t)stream);
  else DISPATCH(winattn_bwd_kernel, 16, hd, grid, p, (hipStream_t)stream);
  CLC_LAUNCH_CHECK();
  const int n = heads * (2 * ws - 1) * (2 * ws - 1);
  hipLaunchKernelGGL(dbias_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)wsb, nbx, n, drelbias, accumulate);
  CLC_LAUNCH_CHECK();
  return 0;
}
