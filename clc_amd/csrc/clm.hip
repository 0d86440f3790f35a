// clm.hip — Conditional Latent Matching ops (standalone module /root/reference/models/CLM.py) on gfx950.
//
// The reference implements these steps as Python loops over pixels (CLM.py:16-20, :39-58: O(B*H*W*9) tensor ops,
// seconds per call).  Here each is one streaming kernel over NHWC latents:
//   clm_sim_colsum   CLM.py:107-109 + :14-20   S = softmax_q(f(y)[p].f(y_r)[q] / tau); the "weighted_x" loop only ever
//                                              uses the COLUMN SUMS  w[q] = sum_p S[p][q]  (weighted_x = w * x), so the
//                                              [HW x HW] matrix is never materialised: one row at a time, softmax in
//                                              LDS, column sums accumulated per workgroup, fixed-order 2-stage reduce.
//   clm_scale_rows   CLM.py:16-22              weighted_x[r][c] = w[r] * x[r][c] written next to x in the concat buffer
//   clm_deform       CLM.py:35-60              9-tap modulated bilinear sampling, zero outside [0,H-1]x[0,W-1]
//   clm_fuse         CLM.py:118-125            softmax over the references of the 1-channel attention logits, weighted
//                                              sum of the aligned features, + y   (SimpleCLM :166-179 variant: features
//                                              are gated by sigmoid(attention) first)
// The 1x1 / 3x3 convolutions around them run on the implicit-GEMM kernel (clc_conv2d, sigmoid epilogue for the modulation).
#include "common.h"

namespace {

constexpr int kMaxHW = 4096;   // similarity row length handled per workgroup (64x64 latent)

// one workgroup = ROWS query rows of one batch item; thread t owns columns t, t+256, ...
template <int ROWS>
__global__ __launch_bounds__(256) void clm_sim_colsum_kernel(const float* __restrict__ yt, int ldy, const float* __restrict__ yr, int ldr,
                                                            int HW, int C, float inv_tau, float* __restrict__ partial) {
  extern __shared__ float sm[];        // [C] query row + [256] reduction scratch
  float* qrow = sm;
  float* red = sm + C;
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  constexpr int MAXC = kMaxHW / 256;   // columns per thread
  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
  const int ncol = (HW + 255) / 256;
  for (int rr = 0; rr < ROWS; ++rr) {
    const int p = chunk * ROWS + rr;
    if (p >= HW) break;                 // block-uniform
    __syncthreads();
    for (int c = tid; c < C; c += 256) qrow[c] = yt[((size_t)b * HW + p) * ldy + c];
    __syncthreads();
    float s[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      const int q = tid + j * 256;
      float d = -INFINITY;
      if (j < ncol && q < HW) {
        const float* r = yr + ((size_t)b * HW + q) * ldr;
        d = 0.f;
        for (int c = 0; c < C; ++c) d = fmaf(qrow[c], r[c], d);
        d *= inv_tau;
      }
      s[j] = d;
      mx = fmaxf(mx, d);
    }
    // block max
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { s[j] = (s[j] == -INFINITY) ? 0.f : expf(s[j] - mx); l += s[j]; }
    red[tid] = l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) acc[j] += s[j] * inv;
  }
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int q = tid + j * 256;
    if (j < ncol && q < HW) partial[((size_t)b * gridDim.x + chunk) * HW + q] = acc[j];
  }
}

__global__ void clm_colsum_reduce_kernel(const float* __restrict__ partial, int nchunks, int HW, float* __restrict__ out) {
  const int b = blockIdx.y, q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= HW) return;
  float s = 0.f;
  for (int k = 0; k < nchunks; ++k) s += partial[((size_t)b * nchunks + k) * HW + q];
  out[(size_t)b * HW + q] = s;
}

__global__ void clm_scale_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, float* __restrict__ out, int ldo,
                                      long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    out[r * ldo + c] = w[r] * x[r * ldx + c];
  }
}

// offset: [B,H,W,18] channel = 2*k + {0: dh, 1: dw}; modulation: [B,H,W,9] (already sigmoid-ed)
__global__ void clm_deform_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ off, int ldo, const float* __restrict__ mod,
                                  int ldm, float* __restrict__ out, int ldy, int B, int H, int W, int C) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float oh = (float)h + off[pix * ldo + 2 * k], ow = (float)w + off[pix * ldo + 2 * k + 1];
      if (oh >= 0.f && oh <= (float)(H - 1) && ow >= 0.f && ow <= (float)(W - 1)) {
        const int h0 = (int)oh, w0 = (int)ow;           // int() truncation; oh >= 0 so this is floor
        const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
        const float lh = oh - (float)h0, lw = ow - (float)w0;
        const float* xb = x + (size_t)b * H * W * ldx + c;
        const float v = (1.f - lh) * (1.f - lw) * xb[(size_t)(h0 * W + w0) * ldx] + lh * (1.f - lw) * xb[(size_t)(h1 * W + w0) * ldx] +
                        (1.f - lh) * lw * xb[(size_t)(h0 * W + w1) * ldx] + lh * lw * xb[(size_t)(h1 * W + w1) * ldx];
        acc += v * mod[pix * ldm + k];
      }
    }
    out[pix * ldy + c] = acc;
  }
}

struct FusePtrs { const float* feat[8]; const float* att[8]; };

// out = sum_m softmax_m(att_m[r]) * g(feat_m[r][c]) + y[r][c];  gate=1 -> feat is first multiplied by sigmoid(att_m[r])
__global__ void clm_fuse_kernel(FusePtrs P, int M, int ldf, int lda, const float* __restrict__ y, int ldy, float* __restrict__ out, int ldo,
                                long rows, int C, int gate) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    float a[8], mx = -INFINITY;
    for (int m = 0; m < M; ++m) { a[m] = P.att[m][r * lda]; mx = fmaxf(mx, a[m]); }
    float l = 0.f, acc = 0.f;
    for (int m = 0; m < M; ++m) {
      const float e = expf(a[m] - mx);
      l += e;
      float f = P.feat[m][r * ldf + c];
      if (gate) f *= 1.f / (1.f + expf(-a[m]));
      acc += e * f;
    }
    out[r * ldo + c] = acc / l + y[r * ldy + c];
  }
}

inline int grid_for(long n) { long b = (n + 1023) / 1024; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" size_t clc_clm_sim_colsum_workspace_bytes(int B, int HW) { return (size_t)B * ((HW + 15) / 16) * HW * sizeof(float); }

extern "C" int clc_clm_sim_colsum(const float* yt, int ldy, const float* yrt, int ldr, int B, int HW, int C, float temperature, float* colsum,
                                  void* ws, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(yt && yrt && colsum && B > 0 && HW > 0 && C > 0 && temperature > 0.f, "clc_clm_sim_colsum: bad args");
  CLC_CHECK(HW <= kMaxHW, "clc_clm_sim_colsum: HW=%d exceeds %d", HW, kMaxHW);
  CLC_CHECK(ws && ws_bytes >= clc_clm_sim_colsum_workspace_bytes(B, HW), "clc_clm_sim_colsum: workspace too small");
  const int nchunks = (HW + 15) / 16;
  hipLaunchKernelGGL(clm_sim_colsum_kernel<16>, dim3(nchunks, B), dim3(256), (size_t)(C + 256) * sizeof(float), ST, yt, ldy, yrt, ldr, HW, C,
                     1.f / temperature, (float*)ws);
  CLC_LAUNCH_CHECK();
  hipLaunchKernelGGL(clm_colsum_reduce_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, ST, (const float*)ws, nchunks, HW, colsum);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_clm_scale_rows(const float* x, int ldx, const float* w, float* out, int ldo, long rows, int C, clc_stream_t stream) {
  CLC_CHECK(x && w && out && rows > 0 && C > 0, "clc_clm_scale_rows: bad args");
  hipLaunchKernelGGL(clm_scale_rows_kernel, dim3(grid_for(rows * C)), dim3(256), 0, ST, x, ldx, w, out, ldo, rows, C);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_clm_deform(const float* x, int ldx, const float* offset, int ldo, const float* modulation, int ldm, float* out, int ldy,
                              int B, int H, int W, int C, clc_stream_t stream) {
  CLC_CHECK(x && offset && modulation && out && B > 0 && H > 0 && W > 0 && C > 0, "clc_clm_deform: bad args");
  CLC_CHECK(ldo >= 18 && ldm >= 9, "clc_clm_deform: offset needs 18 and modulation 9 channels");
  hipLaunchKernelGGL(clm_deform_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, ST, x, ldx, offset, ldo, modulation, ldm, out, ldy, B,
                     H, W, C);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_clm_fuse(const float* const* feats, const float* const* atts, int M, int ldf, int lda, const float* y, int ldy, float* out,
                            int ldo, long rows, int C, int gate, clc_stream_t stream) {
  CLC_CHECK(feats && atts && y && out && M > 0 && M <= 8 && rows > 0 && C > 0, "clc_clm_fuse: bad args (M must be 1..8)");
  FusePtrs P;
  for (int m = 0; m < 8; ++m) { P.feat[m] = m < M ? feats[m] : nullptr; P.att[m] = m < M ? atts[m] : nullptr; }
  hipLaunchKernelGGL(clm_fuse_kernel, dim3(grid_for(rows * C)), dim3(256), 0, ST, P, M, ldf, lda, y, ldy, out, ldo, rows, C, gate);
  CLC_LAUNCH_CHECK();
  return 0;
}
