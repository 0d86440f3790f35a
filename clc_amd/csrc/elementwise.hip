// elementwise.hip — HBM-bound fused elementwise / normalisation / reduction kernels (gfx950).
//
// Everything here is a streaming pass: 16-B loads where the layout allows, grid-stride loops
// capped at 2048 workgroups, wave64 shuffles for row reductions, and two-stage (partials ->
// fixed-order sum) reductions instead of float atomics so results are bitwise reproducible.
//   * LayerNorm fwd/bwd            nn.LayerNorm in Block, /root/reference/models/CLC_run.py:180,183,191-192
//   * activation backward          LeakyReLU / ReLU / GELU of the CompressAI blocks and MLPs
//   * GDN backward pieces          CompressAI GDN
//   * SWAtten gate fwd/bwd         /root/reference/models/CLC_run.py:241-242
//   * column sums, axpby, strided slice copy, fixed-order partial sums, squared-difference
#include "common.h"

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(long n, int per_block) {
  long b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

// ---------------------------------------------------------------- LayerNorm
// C in {64,128,256}: a row is handled by G = C/4 lanes holding one float4 each, so a wave covers 64/G rows per
// instruction (4 rows at C=64) — 4x the bytes in flight of a lane-per-element layout. Row statistics by xor-shuffles
// inside the G-lane group.  General C (<= 512) falls back to one wave per row.
constexpr int LN_MAX_PER_LANE = 8;

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int G>   // G lanes per row, C = 4*G
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                              float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                              const float* __restrict__ gamma2, const float* __restrict__ beta2, long half_rows) {
  // gamma2 / beta2 (paired modules): rows >= half_rows are normalised with the second module's affine parameters
  constexpr int C = 4 * G, RPB = 256 / G;   // rows per block pass
  const int sub = threadIdx.x % G, rl = threadIdx.x / G;
  const f32x4 gm1 = *reinterpret_cast<const f32x4*>(gamma + sub * 4), bt1 = *reinterpret_cast<const f32x4*>(beta + sub * 4);
  const f32x4 gm2 = gamma2 ? *reinterpret_cast<const f32x4*>(gamma2 + sub * 4) : gm1, bt2 = gamma2 ? *reinterpret_cast<const f32x4*>(beta2 + sub * 4) : bt1;
  for (long r = (long)blockIdx.x * RPB + rl; r < rows; r += (long)gridDim.x * RPB) {
    const bool sec = gamma2 != nullptr && r >= half_rows;
    const f32x4 gm = sec ? gm2 : gm1, bt = sec ? bt2 : bt1;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + sub * 4);
    const float mu = group_sum<G>((v[0] + v[1]) + (v[2] + v[3])) / (float)C;
    const f32x4 d = v - mu;
    const float rs = rsqrtf(group_sum<G>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) / (float)C + 1e-5f);
    *reinterpret_cast<f32x4*>(y + r * ldy + sub * 4) = d * rs * gm + bt;
    if (sub == 0 && mean) { mean[r] = mu; rstd[r] = rs; }
  }
}

template <int G>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dx, int lddx,
                                                              const float* __restrict__ dadd, int ldadd, float* __restrict__ ws, long rows,
                                                              const float* __restrict__ gamma2, long half_rows) {
  // gamma2 (paired modules): the first half of the grid works on rows [0, half_rows) with gamma, the second half on
  // [half_rows, rows) with gamma2, so every block's partial dgamma/dbeta row belongs to exactly one module
  constexpr int C = 4 * G, RPB = 256 / G;
  __shared__ f32x4 sm[2][RPB][G];
  const int sub = threadIdx.x % G, rl = threadIdx.x / G;
  const int nbh = gamma2 ? (int)gridDim.x / 2 : (int)gridDim.x;
  const int side = (gamma2 && (int)blockIdx.x >= nbh) ? 1 : 0;
  const long r_begin = side ? half_rows : 0, r_end = gamma2 ? (side ? rows : half_rows) : rows;
  const f32x4 gm = *reinterpret_cast<const f32x4*>((side ? gamma2 : gamma) + sub * 4);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
  for (long r = r_begin + (long)((int)blockIdx.x - side * nbh) * RPB + rl; r < r_end; r += (long)nbh * RPB) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * lddy + sub * 4);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ldx + sub * 4);
    const float mu = mean[r], rs = rstd[r];
    const f32x4 xh = (xv - mu) * rs, g = d * gm;
    dg += d * xh;
    db += d;
    const float s1 = group_sum<G>((g[0] + g[1]) + (g[2] + g[3])) / (float)C;
    const float s2 = group_sum<G>((g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3])) / (float)C;
    f32x4 o = rs * (g - s1 - xh * s2);
    if (dadd) o += *reinterpret_cast<const f32x4*>(dadd + r * ldadd + sub * 4);   // gradient of the residual branch, folded in
    *reinterpret_cast<f32x4*>(dx + r * lddx + sub * 4) = o;
  }
  sm[0][rl][sub] = dg; sm[1][rl][sub] = db;
  __syncthreads();
  if (threadIdx.x < 2 * G) {   // fixed-order sum over the block's row lanes
    const int which = threadIdx.x / G, c = threadIdx.x % G;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < RPB; ++q) t += sm[which][q][c];
    *reinterpret_cast<f32x4*>(ws + ((size_t)blockIdx.x * 2 + which) * C + c * 4) = t;
  }
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                          float* __restrict__ mean, float* __restrict__ rstd, long rows, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (C + 63) / 64;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + i * 64;
      v[i] = (i < per && c < C) ? x[r * ldx + c] : 0.f;
      s += v[i];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + i * 64;
      const float d = (i < per && c < C) ? v[i] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + i * 64;
      if (i < per && c < C) y[r * ldy + c] = (v[i] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0 && mean) { mean[r] = mu; rstd[r] = rs; }
  }
}

// dx per row; per-block partial dgamma/dbeta -> ws[block][2][C]
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx, int lddx,
                                                          const float* __restrict__ dadd, int ldadd, float* __restrict__ ws, long rows, int C) {
  extern __shared__ float sm[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (C + 63) / 64;
  float dg[LN_MAX_PER_LANE], db[LN_MAX_PER_LANE];
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) dg[i] = db[i] = 0.f;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float mu = mean[r], rs = rstd[r];
    float g[LN_MAX_PER_LANE], xh[LN_MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + i * 64;
      const bool ok = (i < per && c < C);
      const float d = ok ? dy[r * lddy + c] : 0.f;
      xh[i] = ok ? (x[r * ldx + c] - mu) * rs : 0.f;
      g[i] = ok ? d * gamma[c] : 0.f;
      dg[i] += d * xh[i];
      db[i] += d;
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + i * 64;
      if (i < per && c < C) dx[r * lddx + c] = rs * (g[i] - s1 - xh[i] * s2) + (dadd ? dadd[r * ldadd + c] : 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + i * 64;
    if (i < per && c < C) { sm[(wave * 2 + 0) * C + c] = dg[i]; sm[(wave * 2 + 1) * C + c] = db[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, cc = c - which * C;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += sm[(w * 2 + which) * C + cc];
    ws[((size_t)blockIdx.x * 2 + which) * C + cc] = s;
  }
}

// out[c] (+)= sum_b ws[b][which][c]; 4 interleaved block groups per column, combined ((g0+g1)+(g2+g3)) (fixed order)
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ ws, int nblocks, int C, float* dgamma, float* dbeta, int accumulate) {
  __shared__ float sm[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < 2 * C) {
    const int which = c / C, cc = c - which * C;
    for (int b = g; b < nblocks; b += 4) s += ws[((size_t)b * 2 + which) * C + cc];
  }
  sm[g][tx] = s;
  __syncthreads();
  if (g == 0 && c < 2 * C) {
    const int which = c / C, cc = c - which * C;
    float* out = which == 0 ? dgamma : dbeta;
    out[cc] = (accumulate ? out[cc] : 0.f) + ((sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]));
  }
}

// ---------------------------------------------------------------- activation backward
__global__ void act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ saved, int lds, int use_pre, int act,
                               float* __restrict__ dz, int lddz, long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float g = dy[r * lddy + c], s = saved[r * lds + c];
    float d;
    switch (act) {
      case CLC_ACT_LRELU: d = s > 0.f ? 1.f : 0.01f; break;
      case CLC_ACT_RELU: d = s > 0.f ? 1.f : 0.f; break;
      case CLC_ACT_GELU: d = gelu_grad_f(s); break;
      case CLC_ACT_HALFTANH: {
        const float t = use_pre ? tanhf(s) : 2.f * s;  // output = 0.5*tanh(v)
        d = 0.5f * (1.f - t * t);
        break;
      }
      case CLC_ACT_SAVED_DERIV: d = s; break;
      default: d = 1.f;
    }
    dz[r * lddz + c] = g * d;
  }
}

// ---------------------------------------------------------------- column sums (two stage)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ld, long rows, int C, float* __restrict__ ws, long rows_per_block) {
  // block handles rows [b*rpb, (b+1)*rpb); thread t handles columns t, t+256, ...
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += x[r * ld + c];
    ws[(size_t)blockIdx.x * C + c] = s;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ ws, int nblocks, int C, float* out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = accumulate ? out[c] : 0.f;
  for (int b = 0; b < nblocks; ++b) s += ws[(size_t)b * C + c];
  out[c] = s;
}

// ---------------------------------------------------------------- GDN backward pieces
__global__ void gdn_bwd_elem_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ v,
                                    float* __restrict__ dxd, float* __restrict__ dv, long n, int inverse) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = dy[i], xv = x[i], vv = v[i];
    if (!inverse) {
      const float rs = rsqrtf(vv);            // y = x * v^-1/2
      dxd[i] = g * rs;
      dv[i] = g * xv * (-0.5f) * rs / vv;     // d/dv v^-1/2 = -1/2 v^-3/2
    } else {
      const float sq = sqrtf(vv);             // y = x * v^1/2
      dxd[i] = g * sq;
      dv[i] = g * xv * 0.5f / sq;
    }
  }
}
__global__ void gdn_bwd_combine_kernel(const float* __restrict__ dxd, const float* __restrict__ x, const float* __restrict__ t,
                                       float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dxd[i] + 2.f * x[i] * t[i];
}

// ---------------------------------------------------------------- GDN parameter re-parametrisation
// CompressAI NonNegativeParametrizer on gamma [C,C] and beta [C] in ONE launch: y = max(x, bound)^2 - pedestal
// (the torch version is ~20 parameter-sized launches per GDN, forward + backward).  gamma_eff is also written transposed:
// the data-gradient 1x1 conv wants it as [Cin][Cout].
__global__ void gdn_reparam_fwd_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C, float gbound, float bbound,
                                       float ped, float* __restrict__ g_eff, float* __restrict__ g_eff_t, float* __restrict__ b_eff) {
  const int n = C * C + C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < C * C) {
      const float t = fmaxf(gamma[i], gbound), y = t * t - ped;
      g_eff[i] = y;
      const int r = i / C, c = i - r * C;
      g_eff_t[c * C + r] = y;
    } else {
      const float t = fmaxf(beta[i - C * C], bbound);
      b_eff[i - C * C] = t * t - ped;
    }
  }
}
// ... for ALL GDN modules of a model in one launch (clc_amd.train refreshes them once per step, like the transposed filter images):
// block b works on 256 consecutive elements of the entry whose block range holds it.
__global__ __launch_bounds__(256) void gdn_reparam_fwd_batched_kernel(const clc_gdn_entry* __restrict__ table, int n_entries) {
  __shared__ int s_e;
  if (threadIdx.x == 0) {
    int e = 0;
    while (e + 1 < n_entries && (int)blockIdx.x >= table[e + 1].first_block) ++e;
    s_e = e;
  }
  __syncthreads();
  const clc_gdn_entry en = table[s_e];
  const int C = en.C, n = C * C + C;
  const int i = ((int)blockIdx.x - en.first_block) * 256 + (int)threadIdx.x;
  if (i >= n) return;
  if (i < C * C) {
    const float t = fmaxf(en.gamma[i], en.gamma_bound), y = t * t - en.pedestal;
    en.gamma_eff[i] = y;
    const int r = i / C, c = i - r * C;
    en.gamma_eff_t[c * C + r] = y;
  } else {
    const float t = fmaxf(en.beta[i - C * C], en.beta_bound);
    en.beta_eff[i - C * C] = t * t - en.pedestal;
  }
}
// backward of the above with the LowerBound gradient rule: dt = 2 max(x,bound) dy; dx = (x >= bound || dt < 0) ? dt : 0
__global__ void gdn_reparam_bwd_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C, float gbound, float bbound,
                                       const float* __restrict__ dg_eff, const float* __restrict__ db_eff, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate) {
  const int n = C * C + C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const bool isg = i < C * C;
    const int j = isg ? i : i - C * C;
    const float x = isg ? gamma[j] : beta[j], bound = isg ? gbound : bbound, dy = isg ? dg_eff[j] : db_eff[j];
    const float dt = 2.f * fmaxf(x, bound) * dy;
    const float dx = (x >= bound || dt < 0.f) ? dt : 0.f;
    float* out = isg ? dgamma : dbeta;
    out[j] = accumulate ? out[j] + dx : dx;
  }
}

// ---------------------------------------------------------------- PixelShuffle(2) backward (+ activation backward)
// dz[n, h, w, 4q + r] = dy[n, 2h + (r>>1), 2w + (r&1), q] * act'(saved[same])   (dy / saved: [N, C/4, 2H, 2W] pixel-major)
// One pass instead of act_bwd + pixel_unshuffle's reshape/permute clone + the channels_last copy.
__global__ void unshuffle_act_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ saved, int lds, int use_pre, int act,
                                         float* __restrict__ dz, int N, int H, int W, int C) {
  const int Q = C / 4;
  const long total = (long)N * H * W * Q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long pix = i / Q;
    const int w = (int)(pix % W);
    const long t = pix / W;
    const int h = (int)(t % H), n = (int)(t / H);
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long sp = ((long)(n * 2 * H + 2 * h + (r >> 1)) * (2 * W) + 2 * w + (r & 1));
      float g = dy[sp * lddy + q];
      if (saved) g *= act_deriv(saved[sp * lds + q], act, use_pre);
      v[r] = g;
    }
    *reinterpret_cast<f32x4*>(dz + pix * C + q * 4) = v;
  }
}

// ---------------------------------------------------------------- gate
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__global__ void gate_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ idn, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = a[i] * sigmoidf_(b[i]) + idn[i];
}
__global__ void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                float* __restrict__ db, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float s = sigmoidf_(b[i]);
    da[i] = g[i] * s;
    db[i] = g[i] * a[i] * s * (1.f - s);
  }
}

__global__ void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b, float beta, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
// dst[r][c] = ((src0 + src1) + src2) + ... — up to 12 pixel-major sources with their own leading dimensions (gradient views into wider
// buffers), fixed order: the gradient of a tensor with several consumers in ONE launch instead of autograd's chain of pairwise adds.
struct SumN { const float* src[12]; int ld[12]; int n; };
__global__ void sum_n_kernel(const SumN s, float* __restrict__ dst, int ldd, long rows, int C4) {
  const long total = rows * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(s.src[0] + r * s.ld[0] + c);
    for (int k = 1; k < s.n; ++k) acc = acc + *reinterpret_cast<const f32x4*>(s.src[k] + r * s.ld[k] + c);
    *reinterpret_cast<f32x4*>(dst + r * ldd + c) = acc;
  }
}
__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    dst[r * ldd + c] = src[r * lds + c];
  }
}

// ---------------------------------------------------------------- fixed-order sums
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ p, int n, float scale, float* out, int accumulate) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += p[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + scale * sm[0];
}
// The scalar tail of the rate-distortion loss (train_CLC.py:43-59) in ONE launch: the three fixed-order sums (sum_partials_kernel's tree: the
// same bits as three clc_sum_partials calls) and the arithmetic the reference writes as tensor expressions, rounding step by rounding step:
//   bpp = (0 + s_y + s_z) / (-num_pixels);  mse = sq / numel;  loss = float(lmbda * 255^2) * mse + bpp
__device__ __forceinline__ float block_sum_256(const float* __restrict__ p, int n, float* sm) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += p[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  const float r = sm[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void rd_combine_kernel(const float* __restrict__ py, int ny, const float* __restrict__ pz, int nz, const float* __restrict__ psq, int nsq,
                                                         float neg_num_pixels, float numel, float c, float* bpp, float* mse, float* loss) {
  __shared__ float sm[256];
  const float sy = block_sum_256(py, ny, sm), sz = block_sum_256(pz, nz, sm), sq = block_sum_256(psq, nsq, sm);
  if (threadIdx.x == 0) {
    const float b = (sy + sz) / neg_num_pixels;
    const float m = sq / numel;
    const float cm = c * m;
    bpp[0] = b;
    mse[0] = m;
    loss[0] = cm + b;
  }
}
// ... and of its backward: the device scalars the three gradient kernels take (g_* may be NULL = no gradient arrived for that output)
__global__ void rd_grad_scalars_kernel(const float* g_bpp, const float* g_mse, const float* g_loss, float neg_num_pixels, float numel, float c, float* g_logsum, float* g_sq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float gl = g_loss ? g_loss[0] : 0.f;
  float gb = gl;                       // d loss / d bpp = 1
  if (g_bpp) gb = gb + g_bpp[0];
  float gm = gl * c;                   // MulBackward
  if (g_mse) gm = gm + g_mse[0];
  g_logsum[0] = gb / neg_num_pixels;   // DivBackward
  g_sq[0] = gm / numel;
}
__global__ __launch_bounds__(256) void sqdiff_partials_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ partials) {
  __shared__ float sm[256];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s += d * d;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void log2_sum_partials_kernel(const float* __restrict__ x, int ld, long rows, int C, float* __restrict__ partials) {
  __shared__ float sm[4];
  const long total = rows * C;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    s += log2f(x[r * ld + (i - r * C)]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void scaled_recip_kernel(const float* __restrict__ x, int ld, long rows, int C, const float* __restrict__ g, float coef,
                                    float* __restrict__ out, int ldo) {
  const long total = rows * C;
  const float k = g[0] * coef;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    out[r * ldo + c] = k / x[r * ld + c];
  }
}
__global__ void scaled_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, const float* __restrict__ g, float coef,
                                   float* __restrict__ out) {
  const float k = g[0] * coef;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = k * (a[i] - b[i]);
}

}  // namespace


// ---- pooling (the reference-retrieval feature extractor, /root/reference/dataloader_ref_cluster.py:41-44, dataloader_CLC.py:250-256) ----
// nn.MaxPool2d(ks, stride, pad) on a pixel-major tensor: one thread per (output pixel, 4 channels); padding = -inf
__global__ void maxpool2d_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int H, int W, int C, int ks, int stride, int pad,
                                 int OH, int OW, long total) {
  const int c4n = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long pix = i / c4n;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), n = (int)(pix / ((long)OW * OH));
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int kh = 0; kh < ks; ++kh) {
      const int iy = oy * stride - pad + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < ks; ++kw) {
        const int ix = ox * stride - pad + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + iy) * W + ix) * ldx + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], v[q]);
      }
    }
    *reinterpret_cast<f32x4*>(y + (size_t)pix * ldy + c) = m;
  }
}
// F.adaptive_avg_pool2d / F.adaptive_max_pool2d to L x L: bin (i, j) covers rows floor(i H / L) .. ceil((i+1) H / L) - 1 (PyTorch's rule);
// out [N][C][L][L] (NCHW order, as the reference flattens it: h.view(N, -1)); the average sums its bin in row-major order
__global__ void adaptive_pool_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int H, int W, int C, int L, int is_max, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int j = (int)(r % L), bi = (int)((r / L) % L), n = (int)(r / ((long)L * L));
    const int y0 = (bi * H) / L, y1 = ((bi + 1) * H + L - 1) / L, x0 = (j * W) / L, x1 = ((j + 1) * W + L - 1) / L;
    float acc = is_max ? -INFINITY : 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float v = x[((size_t)(n * H + yy) * W + xx) * ldx + c];
        acc = is_max ? fmaxf(acc, v) : acc + v;
      }
    if (!is_max) acc = acc / (float)((y1 - y0) * (x1 - x0));
    out[(((size_t)n * C + c) * L + bi) * L + j] = acc;
  }
}

#define ST ((hipStream_t)stream)

extern "C" int clc_log2_sum_partials(const float* x, int ld, long rows, int C, float* partials, int n_partials, clc_stream_t stream) {
  CLC_CHECK(x && partials && rows > 0 && C > 0 && n_partials > 0 && n_partials <= kMaxBlocks, "clc_log2_sum_partials: bad args");
  hipLaunchKernelGGL(log2_sum_partials_kernel, dim3(n_partials), dim3(256), 0, ST, x, ld, rows, C, partials);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_scaled_recip(const float* x, int ld, long rows, int C, const float* g_dev, float coef, float* out, int ldo, clc_stream_t stream) {
  CLC_CHECK(x && g_dev && out && rows > 0 && C > 0, "clc_scaled_recip: bad args");
  hipLaunchKernelGGL(scaled_recip_kernel, dim3(grid_for(rows * C, 1024)), dim3(256), 0, ST, x, ld, rows, C, g_dev, coef, out, ldo);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_scaled_diff(const float* a, const float* b, long n, const float* g_dev, float coef, float* out, clc_stream_t stream) {
  CLC_CHECK(a && b && g_dev && out && n > 0, "clc_scaled_diff: bad args");
  hipLaunchKernelGGL(scaled_diff_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, a, b, n, g_dev, coef, out);
  CLC_LAUNCH_CHECK();
  return 0;
}

static int ln_fwd_impl(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2, long half_rows,
                       float* y, int ldy, float* mean, float* rstd, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(x && gamma && beta && y && rows > 0 && C > 0, "clc_layernorm_fwd: bad args");
  CLC_CHECK(C <= 64 * LN_MAX_PER_LANE, "clc_layernorm_fwd: C=%d too large", C);
  CLC_CHECK((mean == nullptr) == (rstd == nullptr), "clc_layernorm_fwd: mean/rstd must both be given or both NULL");
  const bool vec = (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) &&
                   (gamma2 == nullptr || (aligned16(gamma2) && aligned16(beta2)));
  CLC_CHECK(gamma2 == nullptr || (vec && (C == 64 || C == 128 || C == 256)), "clc_layernorm_fwd_pair: needs C in {64,128,256} and 16-B aligned operands");
  if (vec && C == 64) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<16>, dim3(grid_for(rows, 16)), dim3(256), 0, ST, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, gamma2, beta2, half_rows);
  else if (vec && C == 128) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<32>, dim3(grid_for(rows, 8)), dim3(256), 0, ST, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, gamma2, beta2, half_rows);
  else if (vec && C == 256) hipLaunchKernelGGL(layernorm_fwd_vec_kernel<64>, dim3(grid_for(rows, 4)), dim3(256), 0, ST, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, gamma2, beta2, half_rows);
  else hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, ST, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean,
                                 float* rstd, long rows, int C, clc_stream_t stream) {
  return ln_fwd_impl(x, ldx, gamma, beta, nullptr, nullptr, 0, y, ldy, mean, rstd, rows, C, stream);
}
extern "C" int clc_layernorm_fwd_pair(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                                      long half_rows, float* y, int ldy, float* mean, float* rstd, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(gamma2 && beta2 && half_rows > 0 && half_rows < rows, "clc_layernorm_fwd_pair: bad args");
  return ln_fwd_impl(x, ldx, gamma, beta, gamma2, beta2, half_rows, y, ldy, mean, rstd, rows, C, stream);
}

// (16 rows per block on the 16x16 maps of the slice loop: 2048 rows -> 128 blocks instead of 16 serial ones)
static int ln_bwd_blocks(long rows) { long b = (rows + 15) / 16; return (int)(b < 1 ? 1 : (b > 512 ? 512 : b)); }

extern "C" int clc_layernorm_bwd_blocks(long rows, int paired) {
  const int nb = ln_bwd_blocks(rows);
  return paired ? (nb + 1) / 2 * 2 : nb;
}
extern "C" size_t clc_layernorm_bwd_workspace_bytes(long rows, int C) { return (size_t)(ln_bwd_blocks(rows) + 1) * 2 * C * sizeof(float); }

static int ln_bwd_impl(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* gamma2, long half_rows,
                       const float* mean, const float* rstd, float* dx, int lddx, const float* dx_add, int ld_add, float* dgamma,
                       float* dbeta, float* dgamma2, float* dbeta2, int accumulate, long rows, int C, void* ws, size_t ws_bytes,
                       clc_stream_t stream) {
  CLC_CHECK(dy && x && gamma && mean && rstd && dx && ((dgamma == nullptr) == (dbeta == nullptr)) && rows > 0, "clc_layernorm_bwd: bad args");
  CLC_CHECK(C <= 64 * LN_MAX_PER_LANE, "clc_layernorm_bwd: C=%d too large", C);
  CLC_CHECK(ws && ws_bytes >= clc_layernorm_bwd_workspace_bytes(rows, C), "clc_layernorm_bwd: workspace too small");
  int nb = ln_bwd_blocks(rows);
  if (gamma2) nb = (nb + 1) / 2 * 2;   // an even grid: half of it per module (the workspace has room for one more block)
  const bool vec = (ldx % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(ws) &&
                   (dx_add == nullptr || (ld_add % 4 == 0 && aligned16(dx_add))) && (gamma2 == nullptr || aligned16(gamma2));
  CLC_CHECK(gamma2 == nullptr || (vec && (C == 64 || C == 128 || C == 256)), "clc_layernorm_bwd_pair: needs C in {64,128,256} and 16-B aligned operands");
  if (vec && C == 64) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<16>, dim3(nb), dim3(256), 0, ST, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ld_add, (float*)ws, rows, gamma2, half_rows);
  else if (vec && C == 128) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<32>, dim3(nb), dim3(256), 0, ST, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ld_add, (float*)ws, rows, gamma2, half_rows);
  else if (vec && C == 256) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<64>, dim3(nb), dim3(256), 0, ST, dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, dx_add, ld_add, (float*)ws, rows, gamma2, half_rows);
  else hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), (size_t)8 * C * sizeof(float), ST, dy, lddy, x, ldx, gamma, mean, rstd, dx,
                     lddx, dx_add, ld_add, (float*)ws, rows, C);
  CLC_LAUNCH_CHECK();
  if (dgamma == nullptr) return 0;   // partial rows [blocks][2][C] stay in ws for clc_partial_reduce_batched
  const int nb1 = gamma2 ? nb / 2 : nb;
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, ST, (const float*)ws, nb1, C, dgamma, dbeta, accumulate);
  CLC_LAUNCH_CHECK();
  if (gamma2) {
    CLC_CHECK(dgamma2 && dbeta2, "clc_layernorm_bwd_pair: dgamma2 / dbeta2 missing");
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, ST, (const float*)ws + (size_t)nb1 * 2 * C, nb1, C, dgamma2, dbeta2, accumulate);
    CLC_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int clc_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, int lddx, const float* dx_add, int ld_add, float* dgamma, float* dbeta,
                                 int accumulate, long rows, int C, void* ws, size_t ws_bytes, clc_stream_t stream) {
  return ln_bwd_impl(dy, lddy, x, ldx, gamma, nullptr, 0, mean, rstd, dx, lddx, dx_add, ld_add, dgamma, dbeta, nullptr, nullptr, accumulate, rows, C, ws,
                     ws_bytes, stream);
}
extern "C" int clc_layernorm_bwd_pair(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* gamma2, long half_rows,
                                      const float* mean, const float* rstd, float* dx, int lddx, const float* dx_add, int ld_add,
                                      float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, int accumulate, long rows, int C, void* ws,
                                      size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(gamma2 && half_rows > 0 && half_rows < rows, "clc_layernorm_bwd_pair: bad args");
  return ln_bwd_impl(dy, lddy, x, ldx, gamma, gamma2, half_rows, mean, rstd, dx, lddx, dx_add, ld_add, dgamma, dbeta, dgamma2, dbeta2, accumulate, rows, C,
                     ws, ws_bytes, stream);
}

// Deferred parameter-gradient reductions (LayerNorm gamma/beta, relative-position bias): every backward kernel leaves its
// per-block partial rows in a workspace; ONE launch per step (per 64 entries) sums them all — instead of one 5-16 us
// reduce launch behind each of the ~110 LayerNorm / attention backward kernels of a step.
constexpr int kMaxReduce = 64;
struct ReduceGroup {
  int count;
  int blk_end[kMaxReduce];            // exclusive prefix of workgroups (32 columns each)
  clc_reduce_entry e[kMaxReduce];
};
// out[i] (+)= sum_b partial[b][i]; 32 columns x 8 interleaved block groups per workgroup, combined in a fixed tree
__global__ __launch_bounds__(256) void partial_reduce_batched_kernel(const ReduceGroup g) {
  __shared__ float sm[8][32];
  int idx = 0;
  while (idx + 1 < g.count && (int)blockIdx.x >= g.blk_end[idx]) ++idx;
  const clc_reduce_entry e = g.e[idx];
  const int b0 = idx == 0 ? 0 : g.blk_end[idx - 1];
  const int tx = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - b0) * 32 + tx;
  float s = 0.f;
  if (i < e.n)
    for (int b = grp; b < e.nblocks; b += 8) s += e.partial[(size_t)b * e.n + i];
  sm[grp][tx] = s;
  __syncthreads();
  if (grp == 0 && i < e.n) {
    const float t = ((sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx])) + ((sm[4][tx] + sm[5][tx]) + (sm[6][tx] + sm[7][tx]));
    float* out = i < e.split ? e.out0 + i : e.out1 + (i - e.split);
    *out = (e.accumulate ? *out : 0.f) + t;
  }
}

extern "C" int clc_partial_reduce_batched(const clc_reduce_entry* entries, int count, clc_stream_t stream) {
  CLC_CHECK(entries && count > 0, "clc_partial_reduce_batched: bad args");
  for (int base = 0; base < count; base += kMaxReduce) {
    ReduceGroup g;
    g.count = count - base < kMaxReduce ? count - base : kMaxReduce;
    int blocks = 0;
    for (int k = 0; k < g.count; ++k) {
      const clc_reduce_entry& e = entries[base + k];
      CLC_CHECK(e.partial && e.out0 && e.nblocks > 0 && e.n > 0 && e.split >= 0 && (e.split >= e.n || e.out1), "clc_partial_reduce_batched: bad entry %d", base + k);
      g.e[k] = e;
      blocks += (e.n + 31) / 32;
      g.blk_end[k] = blocks;
    }
    hipLaunchKernelGGL(partial_reduce_batched_kernel, dim3(blocks), dim3(256), 0, ST, g);
    CLC_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int clc_act_bwd(const float* dy, int lddy, const float* saved, int lds, int use_pre, int act, float* dz, int lddz, long rows,
                           int C, clc_stream_t stream) {
  CLC_CHECK(dy && saved && dz && rows > 0 && C > 0, "clc_act_bwd: bad args");
  CLC_CHECK(!(act == CLC_ACT_GELU && !use_pre), "clc_act_bwd: GELU needs the pre-activation");
  CLC_CHECK(act >= CLC_ACT_NONE && act <= CLC_ACT_HALFTANH, "clc_act_bwd: unsupported activation %d", act);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(rows * C, 1024)), dim3(256), 0, ST, dy, lddy, saved, lds, use_pre, act, dz, lddz, rows, C);
  CLC_LAUNCH_CHECK();
  return 0;
}

static int colsum_blocks(long rows) { long b = (rows + 255) / 256; return (int)(b < 1 ? 1 : (b > 512 ? 512 : b)); }
extern "C" size_t clc_colsum_workspace_bytes(long rows, int C) { return (size_t)colsum_blocks(rows) * C * sizeof(float); }
extern "C" int clc_colsum(const float* x, int ld, long rows, int C, float* out, int accumulate, void* ws, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(x && out && rows > 0 && C > 0, "clc_colsum: bad args");
  CLC_CHECK(ws && ws_bytes >= clc_colsum_workspace_bytes(rows, C), "clc_colsum: workspace too small");
  const int nb = colsum_blocks(rows);
  const long rpb = (rows + nb - 1) / nb;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, ST, x, ld, rows, C, (float*)ws, rpb);
  CLC_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, (const float*)ws, nb, C, out, accumulate);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_gdn_bwd_elem(const float* dy, const float* x, const float* v, float* dx_direct, float* dv, long n, int inverse, clc_stream_t stream) {
  CLC_CHECK(dy && x && v && dx_direct && dv && n > 0, "clc_gdn_bwd_elem: bad args");
  hipLaunchKernelGGL(gdn_bwd_elem_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, dy, x, v, dx_direct, dv, n, inverse);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gdn_bwd_combine(const float* dx_direct, const float* x, const float* t, float* dx, long n, clc_stream_t stream) {
  CLC_CHECK(dx_direct && x && t && dx && n > 0, "clc_gdn_bwd_combine: bad args");
  hipLaunchKernelGGL(gdn_bwd_combine_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, dx_direct, x, t, dx, n);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gdn_reparam_fwd(const float* gamma, const float* beta, int C, float gamma_bound, float beta_bound, float pedestal,
                                   float* gamma_eff, float* gamma_eff_t, float* beta_eff, clc_stream_t stream) {
  CLC_CHECK(gamma && beta && gamma_eff && gamma_eff_t && beta_eff && C > 0, "clc_gdn_reparam_fwd: bad args");
  hipLaunchKernelGGL(gdn_reparam_fwd_kernel, dim3(grid_for((long)C * C + C, 256)), dim3(256), 0, ST, gamma, beta, C, gamma_bound, beta_bound, pedestal,
                     gamma_eff, gamma_eff_t, beta_eff);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gdn_reparam_fwd_batched(const clc_gdn_entry* table_dev, int n_entries, int total_blocks, clc_stream_t stream) {
  CLC_CHECK(table_dev && n_entries > 0 && total_blocks > 0, "clc_gdn_reparam_fwd_batched: bad args");
  hipLaunchKernelGGL(gdn_reparam_fwd_batched_kernel, dim3(total_blocks), dim3(256), 0, ST, table_dev, n_entries);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gdn_reparam_bwd(const float* gamma, const float* beta, int C, float gamma_bound, float beta_bound, const float* dgamma_eff,
                                   const float* dbeta_eff, float* dgamma, float* dbeta, int accumulate, clc_stream_t stream) {
  CLC_CHECK(gamma && beta && dgamma_eff && dbeta_eff && dgamma && dbeta && C > 0, "clc_gdn_reparam_bwd: bad args");
  hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(grid_for((long)C * C + C, 256)), dim3(256), 0, ST, gamma, beta, C, gamma_bound, beta_bound,
                     dgamma_eff, dbeta_eff, dgamma, dbeta, accumulate);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_unshuffle_act_bwd(const float* dy, int lddy, const float* saved, int lds, int use_pre, int act, float* dz, int N, int H,
                                     int W, int C, clc_stream_t stream) {
  CLC_CHECK(dy && dz && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "clc_unshuffle_act_bwd: bad args");
  CLC_CHECK(lddy >= C / 4 && (!saved || lds >= C / 4) && aligned16(dz), "clc_unshuffle_act_bwd: bad leading dims / alignment");
  hipLaunchKernelGGL(unshuffle_act_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 4), 256)), dim3(256), 0, ST, dy, lddy, saved, lds, use_pre, act, dz,
                     N, H, W, C);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gate_fwd(const float* a, const float* b, const float* idn, float* out, long n, clc_stream_t stream) {
  CLC_CHECK(a && b && idn && out && n > 0, "clc_gate_fwd: bad args");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, a, b, idn, out, n);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_gate_bwd(const float* dout, const float* a, const float* b, float* da, float* db, long n, clc_stream_t stream) {
  CLC_CHECK(dout && a && b && da && db && n > 0, "clc_gate_bwd: bad args");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, dout, a, b, da, db, n);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_sum_n(const float* const* srcs, const int* lds, int n, float* dst, int ldd, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(srcs && lds && dst && n >= 1 && n <= 12 && rows > 0 && C > 0 && C % 4 == 0 && ldd % 4 == 0 && aligned16(dst), "clc_sum_n: bad args (1..12 sources, C %% 4 == 0)");
  SumN s;
  s.n = n;
  for (int k = 0; k < 12; ++k) { s.src[k] = k < n ? srcs[k] : nullptr; s.ld[k] = k < n ? lds[k] : 0; }
  for (int k = 0; k < n; ++k) CLC_CHECK(srcs[k] && aligned16(srcs[k]) && lds[k] % 4 == 0 && lds[k] >= C, "clc_sum_n: source %d unaligned", k);
  hipLaunchKernelGGL(sum_n_kernel, dim3(grid_for(rows * (C / 4), 1024)), dim3(256), 0, ST, s, dst, ldd, rows, C / 4);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_axpby(const float* a, float alpha, const float* b, float beta, float* out, long n, clc_stream_t stream) {
  CLC_CHECK(a && out && n > 0, "clc_axpby: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, a, alpha, b, beta, out, n);
  CLC_LAUNCH_CHECK();
  return 0;
}
// Patch rows of a few-channel image (the RGB heads, Cin = 3): col[m][(kh*ks+kw)*C + c] = x[n, oh*s-pad+kh, ow*s-pad+kw, c],
// zero outside the image and in the padding columns up to ldc.  One thread per (output pixel, 4-float column group).
__global__ void im2col_small_kernel(const float* __restrict__ x, int ldx, int H, int W, int C, int ks, int stride, int pad,
                                    float* __restrict__ col, int ldc, int OH, int OW, long total) {
  const int groups = ldc / 4, K = ks * ks * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long m = i / groups;
    const int ow = (int)(m % OW);
    const long t = m / OW;
    const int oh = (int)(t % OH), n = (int)(t / OH);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = g * 4 + q;
      float e = 0.f;
      if (k < K) {
        const int tap = k / C, c = k - tap * C;
        const int kh = tap / ks, kw = tap - kh * ks;
        const int iy = oh * stride - pad + kh, ix = ow * stride - pad + kw;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) e = x[((long)(n * H + iy) * W + ix) * ldx + c];
      }
      v[q] = e;
    }
    *reinterpret_cast<f32x4*>(col + m * ldc + g * 4) = v;
  }
}

extern "C" int clc_im2col_small(const float* x, int ldx, int N, int H, int W, int C, int ks, int stride, int pad, float* col, int ldc,
                                int OH, int OW, clc_stream_t stream) {
  CLC_CHECK(x && col && N > 0 && H > 0 && W > 0 && C > 0 && ks > 0 && stride > 0, "clc_im2col_small: bad args");
  CLC_CHECK(ldc % 4 == 0 && ldc >= ks * ks * C && aligned16(col), "clc_im2col_small: ldc must be a multiple of 4 and >= ks*ks*C");
  CLC_CHECK(OH == (H + 2 * pad - ks) / stride + 1 && OW == (W + 2 * pad - ks) / stride + 1, "clc_im2col_small: output dims");
  const long total = (long)N * OH * OW * (ldc / 4);
  hipLaunchKernelGGL(im2col_small_kernel, dim3(grid_for(total, 4096)), dim3(256), 0, ST, x, ldx, H, W, C, ks, stride, pad, col, ldc, OH, OW, total);
  CLC_LAUNCH_CHECK();
  return 0;
}
// ---- RGB-head filters as patch-row matrices -------------------------------------------------------------------------------------
// The stride-2 3x3 head convolution and its 1x1 skip run as 1x1 convolutions over 32-column patch rows (clc_im2col_small).  Their filter
// matrices are the parameters re-laid: w1[o][k] = w3[o][k] for k < 9 cin (the parameter's own [o][kh][kw][c] order), zero beyond;
// ws[o][4 cin + c] = w1x1[o][c] (the window's centre tap), zero elsewhere.  One launch builds both (was 2 x (fill + copy) per step), one
// launch adds both gradients back into the parameters' gradient buffers (was a slice, a copy and an add per tensor).
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w3, const float* __restrict__ w1x1, float* __restrict__ w1,
                                                        float* __restrict__ ws, int co, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= co * 32) return;
  const int o = i >> 5, k = i & 31;
  w1[i] = k < 9 * cin ? w3[o * 9 * cin + k] : 0.f;
  ws[i] = (k >= 4 * cin && k < 5 * cin) ? w1x1[o * cin + k - 4 * cin] : 0.f;
}
__global__ __launch_bounds__(256) void stem_unpack_add_kernel(const float* __restrict__ dw1, const float* __restrict__ dws, float* __restrict__ g3,
                                                              float* __restrict__ g1x1, int co, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= co * 32) return;
  const int o = i >> 5, k = i & 31;
  if (k < 9 * cin) g3[o * 9 * cin + k] += dw1[i];
  if (k >= 4 * cin && k < 5 * cin) g1x1[o * cin + k - 4 * cin] += dws[i];
}
extern "C" int clc_stem_pack(const float* w3, const float* w1x1, float* w1, float* ws, int co, int cin, clc_stream_t stream) {
  CLC_CHECK(w3 && w1x1 && w1 && ws && co > 0 && cin > 0 && 9 * cin <= 32, "clc_stem_pack: bad args (9 * cin must fit 32 columns)");
  hipLaunchKernelGGL(stem_pack_kernel, dim3((co * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w3, w1x1, w1, ws, co, cin);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_stem_unpack_add(const float* dw1, const float* dws, float* g3, float* g1x1, int co, int cin, clc_stream_t stream) {
  CLC_CHECK(dw1 && dws && g3 && g1x1 && co > 0 && cin > 0 && 9 * cin <= 32, "clc_stem_unpack_add: bad args");
  hipLaunchKernelGGL(stem_unpack_add_kernel, dim3((co * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw1, dws, g3, g1x1, co, cin);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_copy2d(const float* src, int lds, float* dst, int ldd, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(src && dst && rows > 0 && C > 0, "clc_copy2d: bad args");
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * C, 1024)), dim3(256), 0, ST, src, lds, dst, ldd, rows, C);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_sum_partials(const float* partials, int n, float scale, float* out, int accumulate, clc_stream_t stream) {
  CLC_CHECK(partials && out && n > 0, "clc_sum_partials: bad args");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, partials, n, scale, out, accumulate);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_rd_combine(const float* py, int ny, const float* pz, int nz, const float* psq, int nsq, float neg_num_pixels, float numel, float c,
                              float* bpp, float* mse, float* loss, clc_stream_t stream) {
  CLC_CHECK(py && pz && psq && ny > 0 && nz > 0 && nsq > 0 && bpp && mse && loss && neg_num_pixels < 0.f && numel > 0.f, "clc_rd_combine: bad args");
  hipLaunchKernelGGL(rd_combine_kernel, dim3(1), dim3(256), 0, ST, py, ny, pz, nz, psq, nsq, neg_num_pixels, numel, c, bpp, mse, loss);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_rd_grad_scalars(const float* g_bpp, const float* g_mse, const float* g_loss, float neg_num_pixels, float numel, float c, float* g_logsum,
                                   float* g_sq, clc_stream_t stream) {
  CLC_CHECK(g_logsum && g_sq && neg_num_pixels < 0.f && numel > 0.f, "clc_rd_grad_scalars: bad args");
  hipLaunchKernelGGL(rd_grad_scalars_kernel, dim3(1), dim3(64), 0, ST, g_bpp, g_mse, g_loss, neg_num_pixels, numel, c, g_logsum, g_sq);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_sqdiff_partials(const float* a, const float* b, long n, float* partials, int n_partials, clc_stream_t stream) {
  CLC_CHECK(a && b && partials && n > 0 && n_partials > 0 && n_partials <= kMaxBlocks, "clc_sqdiff_partials: bad args");
  hipLaunchKernelGGL(sqdiff_partials_kernel, dim3(n_partials), dim3(256), 0, ST, a, b, n, partials);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_maxpool2d(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int ks, int stride, int pad, int OH, int OW, clc_stream_t stream) {
  CLC_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y), "clc_maxpool2d: bad args (C, ld multiples of 4)");
  CLC_CHECK(ks > 0 && stride > 0 && pad >= 0 && 2 * pad <= ks && OH == (H + 2 * pad - ks) / stride + 1 && OW == (W + 2 * pad - ks) / stride + 1, "clc_maxpool2d: output dims");
  const long total = (long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST, x, ldx, y, ldy, H, W, C, ks, stride, pad, OH, OW, total);
  CLC_LAUNCH_CHECK();
  return 0;
}
extern "C" int clc_adaptive_pool2d(const float* x, int ldx, float* out, int N, int H, int W, int C, int L, int is_max, clc_stream_t stream) {
  CLC_CHECK(x && out && N > 0 && H > 0 && W > 0 && C > 0 && L > 0 && L <= H && L <= W && ldx >= C, "clc_adaptive_pool2d: bad args");
  const long total = (long)N * L * L * C;
  hipLaunchKernelGGL(adaptive_pool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ST, x, ldx, out, H, W, C, L, is_max, total);
  CLC_LAUNCH_CHECK();
  return 0;
}
