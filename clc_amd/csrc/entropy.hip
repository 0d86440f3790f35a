// entropy.hip — entropy-model kernels of the CLC path (gfx950): HBM-streaming, wave-reduced.
//
//   * GaussianConditional likelihood + rate, forward and backward
//       (/root/reference/models/CLC_run.py:569-571, :718-736; train_CLC.py:48-51)
//   * quantize("symbols") + build_indexes for the arithmetic coder — INT path, bit-exact
//       (/root/reference/models/CLC_run.py:689-690, 791)
//   * EntropyBottleneck factorised density: likelihood fwd/bwd, aux loss fwd/bwd
//       (/root/reference/models/CLC_run.py:526-530, train_CLC.py:181)
// Algorithmic bytes per latent element (forward): read y, mu, scale (+noise) = 12-16 B,
// write lik, y_hat = 8 B.  The rate sum is a wave shuffle reduction -> one partial per
// workgroup -> fixed-order final sum (no float atomics).
#include "common.h"

namespace {

constexpr float kScaleBound = 0.11f;
constexpr float kLikBound = 1e-9f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

__device__ __forceinline__ float std_cum(float x) { return 0.5f * erfcf(-kInvSqrt2 * x); }

__global__ __launch_bounds__(256) void gauss_lik_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mu, int ldmu,
                                                          const float* __restrict__ scale, int ldsc, const float* __restrict__ noise, int ldn,
                                                          float* __restrict__ lik, int ldl, float* __restrict__ y_hat, int ldh, long rows, int C,
                                                          int mode, float* __restrict__ bits_partial) {
  __shared__ float sm[4];
  const long total = rows * C;
  float bits = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float yv = y[r * ldy + c], m = mu[r * ldmu + c];
    const float rq = rintf(yv - m);  // torch.round = round-half-to-even
    const float yin = (mode == 0) ? yv + noise[r * ldn + c] : rq + m;
    const float sg = fmaxf(scale[r * ldsc + c], kScaleBound);
    const float v = fabsf(yin - m);
    const float l = std_cum((0.5f - v) / sg) - std_cum((-0.5f - v) / sg);
    const float lb = fmaxf(l, kLikBound);
    lik[r * ldl + c] = lb;
    if (y_hat) y_hat[r * ldh + c] = rq + m;
    bits += log2f(lb);
  }
  if (bits_partial) {
    bits = wave_sum(bits);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) bits_partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  }
}

__global__ __launch_bounds__(256) void gauss_lik_bwd_kernel(const float* __restrict__ dlik, int lddl, const float* __restrict__ y, int ldy,
                                                          const float* __restrict__ mu, int ldmu, const float* __restrict__ scale, int ldsc,
                                                          const float* __restrict__ noise, int ldn, float* __restrict__ dy, int lddy,
                                                          float* __restrict__ dmu, int lddmu, float* __restrict__ dscale, int lddsc, long rows,
                                                          int C, int mode, const float* __restrict__ dy_add, int ldadd) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float yv = y[r * ldy + c], m = mu[r * ldmu + c], sc = scale[r * ldsc + c];
    const float yin = (mode == 0) ? yv + noise[r * ldn + c] : rintf(yv - m) + m;
    const float sg = fmaxf(sc, kScaleBound);
    const float d = yin - m, v = fabsf(d);
    const float a = (0.5f - v) / sg, b = (-0.5f - v) / sg;
    const float l = std_cum(a) - std_cum(b);
    float g = dlik[r * lddl + c];
    // LowerBound(lik, 1e-9): pass if lik >= bound or grad < 0
    if (!(l >= kLikBound || g < 0.f)) g = 0.f;
    const float pa = kInvSqrt2Pi * expf(-0.5f * a * a), pb = kInvSqrt2Pi * expf(-0.5f * b * b);
    const float dl_dv = (pb - pa) / sg;
    float dl_dsg = (pb * b - pa * a) / sg;   // d/dsg [Phi(a) - Phi(b)], da/dsg = -a/sg
    float gs = g * dl_dsg;
    // LowerBound(scale, 0.11): pass if scale >= bound or grad < 0
    if (!(sc >= kScaleBound || gs < 0.f)) gs = 0.f;
    dscale[r * lddsc + c] = gs;
    float gy = 0.f;
    if (mode == 0) {
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      gy = g * dl_dv * sgn;
    }
    if (dy) dy[r * lddy + c] = dy_add ? gy + dy_add[r * ldadd + c] : gy;   // (+ the straight-through gradient of y_hat = round(y - mu) + mu)
    if (dmu) dmu[r * lddmu + c] = -gy;
  }
}

__global__ __launch_bounds__(256) void quantize_build_indexes_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mu, int ldmu,
                                                                   const float* __restrict__ scale, int ldsc, const float* __restrict__ table,
                                                                   int n_scales, int32_t* __restrict__ symbols, int32_t* __restrict__ indexes,
                                                                   float* __restrict__ y_hat, int ldh, long rows, int C) {
  __shared__ float tb[256];
  for (int i = threadIdx.x; i < n_scales; i += 256) tb[i] = table[i];
  __syncthreads();
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float m = mu[r * ldmu + c];
    const float q = rintf(y[r * ldy + c] - m);
    const float sg = fmaxf(scale[r * ldsc + c], kScaleBound);
    int idx = n_scales - 1;
    for (int k = 0; k < n_scales - 1; ++k) idx -= (sg <= tb[k]) ? 1 : 0;
    // symbols/indexes are emitted in the coder's order: element (row r, channel c) of an NHWC slice
    // belongs at NCHW position handled by the caller's (rows,C) -> stream permutation
    symbols[i] = (int32_t)q;
    indexes[i] = idx;
    if (y_hat) y_hat[r * ldh + c] = q + m;
  }
}

// ---------------------------------------------------------------- EntropyBottleneck
// Per channel: M0[3x1] b0[3] f0[3] | M1[3x3] b1[3] f1[3] | M2 | M3 | M4[1x3] b4[1]  (58 floats)
struct EBParams {
  const float* m[5]; const float* b[5]; const float* f[4];
};
struct EBGrads {
  float* m[5]; float* b[5]; float* f[4];
};
__host__ __device__ constexpr int eb_f(int k) { return (k == 0 || k == 5) ? 1 : 3; }

__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }

struct EBChan {  // one channel's parameters, transformed, in registers
  float sp[5][9];   // softplus(matrix) [out][in]
  float sg[5][9];   // sigmoid(matrix)  (= d softplus)
  float b[5][3];
  float tf[4][3];   // tanh(factor)
};

__device__ void eb_load(const EBParams& P, int ch, EBChan& c) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int fo = eb_f(k + 1), fi = eb_f(k);
    for (int i = 0; i < fo * fi; ++i) { const float mv = P.m[k][ch * fo * fi + i]; c.sp[k][i] = softplusf_(mv); c.sg[k][i] = sigmoid_(mv); }
    for (int i = 0; i < fo; ++i) c.b[k][i] = P.b[k][ch * fo + i];
    if (k < 4) for (int i = 0; i < fo; ++i) c.tf[k][i] = tanhf(P.f[k][ch * fo + i]);
  }
}

// forward of logits_cumulative keeping per-layer inputs h[k] and tanh(pre_k)
struct EBTape { float h[5][3]; float tp[4][3]; float out; };

__device__ void eb_forward(const EBChan& c, float x, EBTape& t) {
  t.h[0][0] = x;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int fo = eb_f(k + 1), fi = eb_f(k);
    float pre[3];
    for (int i = 0; i < fo; ++i) {
      float a = c.b[k][i];
      for (int j = 0; j < fi; ++j) a = fmaf(c.sp[k][i * fi + j], t.h[k][j], a);
      pre[i] = a;
    }
    if (k < 4) {
      for (int i = 0; i < fo; ++i) { const float tp = tanhf(pre[i]); t.tp[k][i] = tp; t.h[k + 1][i] = pre[i] + c.tf[k][i] * tp; }
    } else {
      t.out = pre[0];
    }
  }
}

// reverse pass: accumulates parameter grads (raw-parameter space) into g*, returns d out / d x * dout
struct EBAcc { float m[5][9]; float b[5][3]; float f[4][3]; };

__device__ float eb_backward(const EBChan& c, const EBTape& t, float dout, EBAcc* acc) {
  float dh[3] = {dout, 0.f, 0.f};  // gradient wrt h[k+1] (or out for k = 4)
#pragma unroll
  for (int k = 4; k >= 0; --k) {
    const int fo = eb_f(k + 1), fi = eb_f(k);
    float dpre[3];
    for (int i = 0; i < fo; ++i) {
      if (k < 4) {
        const float tp = t.tp[k][i];
        dpre[i] = dh[i] * (1.f + c.tf[k][i] * (1.f - tp * tp));
        if (acc) acc->f[k][i] += dh[i] * tp * (1.f - c.tf[k][i] * c.tf[k][i]);
      } else {
        dpre[i] = dh[i];
      }
      if (acc) acc->b[k][i] += dpre[i];
    }
    float dprev[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < fo; ++i)
      for (int j = 0; j < fi; ++j) {
        if (acc) acc->m[k][i * fi + j] += dpre[i] * t.h[k][j] * c.sg[k][i * fi + j];
        dprev[j] = fmaf(c.sp[k][i * fi + j], dpre[i], dprev[j]);
      }
    dh[0] = dprev[0]; dh[1] = dprev[1]; dh[2] = dprev[2];
  }
  return dh[0];
}

// z: [rows, C] NHWC (ld), lik likewise. One workgroup per channel (rows are few: B*H/64*W/64).
// mode 0: v = z + noise; mode 1: v = round(z - med) + med.
__global__ __launch_bounds__(256) void eb_lik_fwd_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ noise, int ldn,
                                                       const float* __restrict__ quantiles, EBParams P, float* __restrict__ lik, int ldl,
                                                       float* __restrict__ z_hat, int ldh, long rows, int C, int mode) {
  const int ch = blockIdx.x;
  EBChan c;
  eb_load(P, ch, c);
  const float med = quantiles[ch * 3 + 1];
  for (long r = threadIdx.x; r < rows; r += 256) {
    const float zv = z[r * ldz + ch];
    const float rq = rintf(zv - med) + med;
    const float v = (mode == 0) ? zv + noise[r * ldn + ch] : rq;
    EBTape tl, tu;
    eb_forward(c, v - 0.5f, tl);
    eb_forward(c, v + 0.5f, tu);
    const float sm = tl.out + tu.out;
    const float s = sm > 0.f ? -1.f : (sm < 0.f ? 1.f : 0.f);
    const float l = fabsf(sigmoid_(s * tu.out) - sigmoid_(s * tl.out));
    lik[r * ldl + ch] = fmaxf(l, kLikBound);
    if (z_hat) z_hat[r * ldh + ch] = rq;
  }
}

__device__ void block_reduce_store(float v, float* dst, float* sm) {
  // deterministic tree over 256 threads
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *dst = sm[0];
  __syncthreads();
}

__global__ __launch_bounds__(256) void eb_lik_bwd_kernel(const float* __restrict__ dlik, int lddl, const float* __restrict__ z, int ldz,
                                                       const float* __restrict__ noise, int ldn, const float* __restrict__ quantiles, EBParams P,
                                                       EBGrads G, float* __restrict__ dz, int lddz, long rows, int C, int mode) {
  const int ch = blockIdx.x;
  EBChan c;
  eb_load(P, ch, c);
  const float med = quantiles[ch * 3 + 1];
  EBAcc acc;
  for (int k = 0; k < 5; ++k) { for (int i = 0; i < 9; ++i) acc.m[k][i] = 0.f; for (int i = 0; i < 3; ++i) acc.b[k][i] = 0.f; }
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 3; ++i) acc.f[k][i] = 0.f;
  for (long r = threadIdx.x; r < rows; r += 256) {
    const float zv = z[r * ldz + ch];
    const float v = (mode == 0) ? zv + noise[r * ldn + ch] : rintf(zv - med) + med;
    EBTape tl, tu;
    eb_forward(c, v - 0.5f, tl);
    eb_forward(c, v + 0.5f, tu);
    const float smm = tl.out + tu.out;
    const float s = smm > 0.f ? -1.f : (smm < 0.f ? 1.f : 0.f);
    const float su = sigmoid_(s * tu.out), sl = sigmoid_(s * tl.out);
    const float diff = su - sl, l = fabsf(diff);
    float g = dlik[r * lddl + ch];
    if (!(l >= kLikBound || g < 0.f)) g = 0.f;
    const float sd = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    const float gu = g * sd * su * (1.f - su) * s;
    const float gl = -g * sd * sl * (1.f - sl) * s;
    const float dxu = eb_backward(c, tu, gu, &acc);
    const float dxl = eb_backward(c, tl, gl, &acc);
    if (dz) dz[r * lddz + ch] = (mode == 0) ? (dxu + dxl) : 0.f;
  }
  // per-channel parameter gradients: deterministic workgroup reduction — the SAME pairwise tree as block_reduce_store (t += t + o,
  // o = 128 ... 1), but for all accumulators at once: 2 x 9 barriers instead of 58 x 10 (this kernel was 100 us for 24 K elements)
  __shared__ float red[36][256];
  __shared__ float tot[72];
  float flat[72];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int i = 0; i < 9; ++i) flat[k * 9 + i] = acc.m[k][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) flat[45 + k * 3 + i] = acc.b[k][i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) flat[60 + k * 3 + i] = acc.f[k][i];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < 36; ++i) red[i][threadIdx.x] = flat[half * 36 + i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      for (int idx = threadIdx.x; idx < 36 * o; idx += 256) {
        const int i = idx / o, t = idx - i * o;
        red[i][t] += red[i][t + o];
      }
      __syncthreads();
    }
    if (threadIdx.x < 36) tot[half * 36 + threadIdx.x] = red[threadIdx.x][0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int fo = eb_f(k + 1), fi = eb_f(k);
      for (int i = 0; i < fo * fi; ++i) G.m[k][ch * fo * fi + i] = tot[k * 9 + i];
      for (int i = 0; i < fo; ++i) G.b[k][ch * fo + i] = tot[45 + k * 3 + i];
      if (k < 4) for (int i = 0; i < fo; ++i) G.f[k][ch * fo + i] = tot[60 + k * 3 + i];
    }
  }
}

// aux loss: sum_c sum_q |logits(quantiles[c][q]) - target[q]| ; grad wrt quantiles only (params detached)
__global__ void eb_aux_kernel(const float* __restrict__ quantiles, EBParams P, const float* __restrict__ target, float* __restrict__ loss_partial,
                              float* __restrict__ dquant, int C) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  EBChan c;
  eb_load(P, ch, c);
  float loss = 0.f;
  for (int q = 0; q < 3; ++q) {
    EBTape t;
    eb_forward(c, quantiles[ch * 3 + q], t);
    const float d = t.out - target[q];
    loss += fabsf(d);
    if (dquant) dquant[ch * 3 + q] = eb_backward(c, t, d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f), nullptr);
  }
  loss_partial[ch] = loss;
}

inline int grid_for(long n) { long b = (n + 1023) / 1024; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int clc_gauss_lik_partials(long rows, int C) { return grid_for(rows * C); }

extern "C" int clc_gauss_lik_fwd(const float* y, int ldy, const float* mu, int ldmu, const float* scale, int ldsc, const float* noise,
                                 int ldn, float* lik, int ldl, float* y_hat, int ldh, long rows, int C, int mode, float* bits_partial,
                                 int n_partials, clc_stream_t stream) {
  CLC_CHECK(y && mu && scale && lik && rows > 0 && C > 0, "clc_gauss_lik_fwd: bad args");
  CLC_CHECK(mode == 1 || noise, "clc_gauss_lik_fwd: training mode needs noise");
  const int nb = grid_for(rows * C);
  CLC_CHECK(!bits_partial || n_partials == nb, "clc_gauss_lik_fwd: n_partials must be clc_gauss_lik_partials() = %d", nb);
  hipLaunchKernelGGL(gauss_lik_fwd_kernel, dim3(nb), dim3(256), 0, ST, y, ldy, mu, ldmu, scale, ldsc, noise, ldn, lik, ldl, y_hat, ldh, rows, C,
                     mode, bits_partial);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_gauss_lik_bwd(const float* dlik, int lddl, const float* y, int ldy, const float* mu, int ldmu, const float* scale,
                                 int ldsc, const float* noise, int ldn, float* dy, int lddy, float* dmu, int lddmu, float* dscale,
                                 int lddsc, long rows, int C, int mode, const float* dy_add, int ldadd, clc_stream_t stream) {
  CLC_CHECK(dlik && y && mu && scale && dscale && rows > 0 && C > 0, "clc_gauss_lik_bwd: bad args");
  CLC_CHECK(mode == 1 || noise, "clc_gauss_lik_bwd: training mode needs noise");
  hipLaunchKernelGGL(gauss_lik_bwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, ST, dlik, lddl, y, ldy, mu, ldmu, scale, ldsc, noise, ldn, dy,
                     lddy, dmu, lddmu, dscale, lddsc, rows, C, mode, dy_add, ldadd);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_quantize_build_indexes(const float* y, int ldy, const float* mu, int ldmu, const float* scale, int ldsc,
                                          const float* scale_table, int n_scales, int32_t* symbols, int32_t* indexes, float* y_hat,
                                          int ldh, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(y && mu && scale && scale_table && symbols && indexes && rows > 0 && C > 0, "clc_quantize_build_indexes: bad args");
  CLC_CHECK(n_scales > 1 && n_scales <= 256, "clc_quantize_build_indexes: n_scales out of range");
  hipLaunchKernelGGL(quantize_build_indexes_kernel, dim3(grid_for(rows * C)), dim3(256), 0, ST, y, ldy, mu, ldmu, scale, ldsc, scale_table,
                     n_scales, symbols, indexes, y_hat, ldh, rows, C);
  CLC_LAUNCH_CHECK();
  return 0;
}

static EBParams mk_params(const float* const* m, const float* const* b, const float* const* f) {
  EBParams P;
  for (int k = 0; k < 5; ++k) { P.m[k] = m[k]; P.b[k] = b[k]; }
  for (int k = 0; k < 4; ++k) P.f[k] = f[k];
  return P;
}

extern "C" int clc_eb_lik_fwd(const float* z, int ldz, const float* noise, int ldn, const float* quantiles, const float* const* matrices,
                              const float* const* biases, const float* const* factors, float* lik, int ldl, float* z_hat, int ldh,
                              long rows, int C, int mode, clc_stream_t stream) {
  CLC_CHECK(z && quantiles && matrices && biases && factors && lik && rows > 0 && C > 0, "clc_eb_lik_fwd: bad args");
  CLC_CHECK(mode == 1 || noise, "clc_eb_lik_fwd: training mode needs noise");
  hipLaunchKernelGGL(eb_lik_fwd_kernel, dim3(C), dim3(256), 0, ST, z, ldz, noise, ldn, quantiles, mk_params(matrices, biases, factors), lik, ldl,
                     z_hat, ldh, rows, C, mode);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_eb_lik_bwd(const float* dlik, int lddl, const float* z, int ldz, const float* noise, int ldn, const float* quantiles,
                              const float* const* matrices, const float* const* biases, const float* const* factors,
                              float* const* dmatrices, float* const* dbiases, float* const* dfactors, float* dz, int lddz, long rows,
                              int C, int mode, clc_stream_t stream) {
  CLC_CHECK(dlik && z && quantiles && matrices && biases && factors && dmatrices && dbiases && dfactors && rows > 0 && C > 0,
            "clc_eb_lik_bwd: bad args");
  CLC_CHECK(mode == 1 || noise, "clc_eb_lik_bwd: training mode needs noise");
  EBGrads G;
  for (int k = 0; k < 5; ++k) { G.m[k] = dmatrices[k]; G.b[k] = dbiases[k]; }
  for (int k = 0; k < 4; ++k) G.f[k] = dfactors[k];
  hipLaunchKernelGGL(eb_lik_bwd_kernel, dim3(C), dim3(256), 0, ST, dlik, lddl, z, ldz, noise, ldn, quantiles, mk_params(matrices, biases, factors),
                     G, dz, lddz, rows, C, mode);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_eb_aux(const float* quantiles, const float* const* matrices, const float* const* biases, const float* const* factors,
                          const float* target, float* loss_partial /* [C] */, float* dquantiles /* [C*3] or NULL */, int C,
                          clc_stream_t stream) {
  CLC_CHECK(quantiles && matrices && biases && factors && target && loss_partial && C > 0, "clc_eb_aux: bad args");
  hipLaunchKernelGGL(eb_aux_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, quantiles, mk_params(matrices, biases, factors), target, loss_partial,
                     dquantiles, C);
  CLC_LAUNCH_CHECK();
  return 0;
}
