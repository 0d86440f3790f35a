// msssim.hip — MS-SSIM distortion term, forward and backward (gfx950).
//
// Replaces pytorch_msssim.ms_ssim(x_hat, x, data_range=1) of the MS-SSIM training configs
// (/root/reference/train_CLC.py:33-34,55-57; algorithm: SURVEY.md A.6 — 11-tap Gaussian sigma 1.5, separable, "valid",
// K = (0.01, 0.03), 5 scales, 2x2 average pooling between scales).  NHWC fp32 images (C = 3).
// Per scale, ONE kernel computes the five filtered moments (E[x], E[y], E[x^2], E[y^2], E[xy]) of a 16x16 output tile
// from a 26x26 window staged in LDS (row pass into LDS, column pass in registers), forms the cs / ssim maps and reduces
// them to per-workgroup partial sums; a second tiny kernel sums the partials in a fixed order (no float atomics).
// Backward recomputes the moments, emits the three coefficient maps dL/dE[x], dL/dE[x^2], dL/dE[xy], and an adjoint
// ("full") separable filter kernel folds them into dL/dx = G^T*A1 + 2x G^T*A11 + y G^T*A12 (+ the pooled gradient
// coming from the next coarser scale).
#include "common.h"

namespace {

constexpr int KW = 11, R = 5, TS = 16, WIN = TS + KW - 1;   // 26
__constant__ float kGauss[KW];

struct SsimParams {
  const float* x; const float* y; int ldx, ldy;
  int B, H, W, C, OH, OW;
  float C1, C2;
};

__device__ __forceinline__ void moments_tile(const SsimParams& p, int b, int ch, int oy0, int ox0, float (*xs)[WIN], float (*ys)[WIN],
                                             float (*hx)[WIN][TS], float e[5]) {
  const int tid = threadIdx.x;
  for (int i = tid; i < WIN * WIN; i += 256) {
    const int r = i / WIN, c = i - r * WIN;
    const int yy = oy0 + r, xx = ox0 + c;
    const bool ok = yy < p.H && xx < p.W;
    const size_t pix = ((size_t)b * p.H + yy) * p.W + xx;
    xs[r][c] = ok ? p.x[pix * p.ldx + ch] : 0.f;
    ys[r][c] = ok ? p.y[pix * p.ldy + ch] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < WIN * TS; i += 256) {
    const int r = i / TS, j = i - r * TS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float g = kGauss[k], xv = xs[r][j + k], yv = ys[r][j + k];
      a0 = fmaf(g, xv, a0); a1 = fmaf(g, yv, a1); a2 = fmaf(g, xv * xv, a2); a3 = fmaf(g, yv * yv, a3); a4 = fmaf(g, xv * yv, a4);
    }
    hx[0][r][j] = a0; hx[1][r][j] = a1; hx[2][r][j] = a2; hx[3][r][j] = a3; hx[4][r][j] = a4;
  }
  __syncthreads();
  const int i = tid / TS, j = tid % TS;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) a = fmaf(kGauss[k], hx[q][i + k][j], a);
    e[q] = a;
  }
}

// grid: (tiles_x, tiles_y, B*C); partial[(bc * ntiles + tile) * 2 + {0: cs, 1: ssim}]
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const SsimParams p, float* __restrict__ partial) {
  __shared__ float xs[WIN][WIN], ys[WIN][WIN], hx[5][WIN][TS];
  __shared__ float r0[256], r1[256];
  const int bc = blockIdx.z, b = bc / p.C, ch = bc - b * p.C;
  const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
  float e[5];
  moments_tile(p, b, ch, oy0, ox0, xs, ys, hx, e);
  const int i = threadIdx.x / TS, j = threadIdx.x % TS;
  float cs = 0.f, ss = 0.f;
  if (oy0 + i < p.OH && ox0 + j < p.OW) {
    const float mu1 = e[0], mu2 = e[1];
    const float s11 = e[2] - mu1 * mu1, s22 = e[3] - mu2 * mu2, s12 = e[4] - mu1 * mu2;
    cs = (2.f * s12 + p.C2) / (s11 + s22 + p.C2);
    ss = ((2.f * mu1 * mu2 + p.C1) / (mu1 * mu1 + mu2 * mu2 + p.C1)) * cs;
  }
  r0[threadIdx.x] = cs; r1[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    partial[((size_t)bc * ntiles + tile) * 2 + 0] = r0[0];
    partial[((size_t)bc * ntiles + tile) * 2 + 1] = r1[0];
  }
}

// means[bc*2 + q] = sum_tiles partial / npos
__global__ void ssim_reduce_kernel(const float* __restrict__ partial, int ntiles, float inv_npos, float* __restrict__ means, int nbc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbc * 2) return;
  const int bc = i >> 1, q = i & 1;
  float s = 0.f;
  for (int t = 0; t < ntiles; ++t) s += partial[((size_t)bc * ntiles + t) * 2 + q];
  means[i] = s * inv_npos;
}

// coefficient maps A[q][b][oy][ox][ch], q = 0: dL/dE[x], 1: dL/dE[x^2], 2: dL/dE[xy]; g = upstream grads of the two means
__global__ __launch_bounds__(256) void ssim_bwd_coef_kernel(const SsimParams p, const float* __restrict__ g_means, float inv_npos,
                                                           float* __restrict__ A) {
  __shared__ float xs[WIN][WIN], ys[WIN][WIN], hx[5][WIN][TS];
  const int bc = blockIdx.z, b = bc / p.C, ch = bc - b * p.C;
  const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
  float e[5];
  moments_tile(p, b, ch, oy0, ox0, xs, ys, hx, e);
  const int i = threadIdx.x / TS, j = threadIdx.x % TS;
  const int oy = oy0 + i, ox = ox0 + j;
  if (oy >= p.OH || ox >= p.OW) return;
  const float g_cs = g_means[bc * 2 + 0] * inv_npos, g_ss = g_means[bc * 2 + 1] * inv_npos;
  const float mu1 = e[0], mu2 = e[1];
  const float s11 = e[2] - mu1 * mu1, s22 = e[3] - mu2 * mu2, s12 = e[4] - mu1 * mu2;
  const float Dc = s11 + s22 + p.C2, cs = (2.f * s12 + p.C2) / Dc;
  const float Dl = mu1 * mu1 + mu2 * mu2 + p.C1, l = (2.f * mu1 * mu2 + p.C1) / Dl;
  const float a = g_cs + g_ss * l, bb = g_ss * cs;
  const float dcs_ds11 = -cs / Dc, dcs_ds12 = 2.f / Dc;
  const float dl_dmu1 = (2.f * mu2 - 2.f * l * mu1) / Dl;
  const size_t plane = (size_t)p.B * p.OH * p.OW * p.C;
  const size_t o = (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + ch;
  A[o] = bb * dl_dmu1 + a * (dcs_ds11 * (-2.f * mu1) + dcs_ds12 * (-mu2));
  A[plane + o] = a * dcs_ds11;
  A[2 * plane + o] = a * dcs_ds12;
}

// dx[b,y,x,c] = F1 + 2 x F11 + y F12 (+ 0.25 * dnext[b, y/2, x/2, c]),  F = full separable correlation of the A maps
__global__ __launch_bounds__(256) void ssim_bwd_adjoint_kernel(const SsimParams p, const float* __restrict__ A, const float* __restrict__ dnext,
                                                              float* __restrict__ dx, int lddx) {
  __shared__ float as[3][WIN][WIN], ha[3][WIN][TS];
  const int bc = blockIdx.z, b = bc / p.C, ch = bc - b * p.C;
  const int y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;   // tile of INPUT pixels
  const size_t plane = (size_t)p.B * p.OH * p.OW * p.C;
  const int tid = threadIdx.x;
  // window of the A maps: positions (y0 - 10 + r, x0 - 10 + c), r,c in [0,26)
  for (int i = tid; i < WIN * WIN; i += 256) {
    const int r = i / WIN, c = i - r * WIN;
    const int oy = y0 - (KW - 1) + r, ox = x0 - (KW - 1) + c;
    const bool ok = oy >= 0 && ox >= 0 && oy < p.OH && ox < p.OW;
    const size_t o = ok ? (((size_t)b * p.OH + oy) * p.OW + ox) * p.C + ch : 0;
    as[0][r][c] = ok ? A[o] : 0.f; as[1][r][c] = ok ? A[plane + o] : 0.f; as[2][r][c] = ok ? A[2 * plane + o] : 0.f;
  }
  __syncthreads();
  // dx(y,x) = sum_{ky,kx} g[ky] g[kx] A[y-ky, x-kx]  -> window index (y - y0 + 10 - ky): flipped taps (kernel is symmetric anyway)
  for (int i = tid; i < WIN * TS; i += 256) {
    const int r = i / TS, j = i - r * TS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const float g = kGauss[k];
      a0 = fmaf(g, as[0][r][j + (KW - 1) - k], a0); a1 = fmaf(g, as[1][r][j + (KW - 1) - k], a1); a2 = fmaf(g, as[2][r][j + (KW - 1) - k], a2);
    }
    ha[0][r][j] = a0; ha[1][r][j] = a1; ha[2][r][j] = a2;
  }
  __syncthreads();
  const int i = tid / TS, j = tid % TS;
  const int yy = y0 + i, xx = x0 + j;
  if (yy >= p.H || xx >= p.W) return;
  float f[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) a = fmaf(kGauss[k], ha[q][i + (KW - 1) - k][j], a);
    f[q] = a;
  }
  const size_t pix = ((size_t)b * p.H + yy) * p.W + xx;
  float v = f[0] + 2.f * p.x[pix * p.ldx + ch] * f[1] + p.y[pix * p.ldy + ch] * f[2];
  if (dnext) v += 0.25f * dnext[(((size_t)b * (p.H / 2) + yy / 2) * (p.W / 2) + xx / 2) * p.C + ch];
  dx[pix * lddx + ch] = v;
}

__global__ void pool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int B, int H, int W, int C) {
  const long total = (long)B * (H / 2) * (W / 2) * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long q = i / C;
    const int ox = (int)(q % (W / 2)), oy = (int)((q / (W / 2)) % (H / 2)), b = (int)(q / ((long)(W / 2) * (H / 2)));
    const size_t p00 = ((size_t)b * H + 2 * oy) * W + 2 * ox;
    out[i] = 0.25f * ((x[p00 * ldx + c] + x[(p00 + 1) * ldx + c]) + (x[(p00 + W) * ldx + c] + x[(p00 + W + 1) * ldx + c]));
  }
}

bool g_gauss_ready = false;
int ensure_gauss() {
  if (g_gauss_ready) return 0;
  float g[KW], s = 0.f;
  for (int i = 0; i < KW; ++i) { const float c = (float)(i - R); g[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f)); s += g[i]; }
  for (int i = 0; i < KW; ++i) g[i] /= s;
  if (hipMemcpyToSymbol(HIP_SYMBOL(kGauss), g, sizeof(g)) != hipSuccess) return -1;
  g_gauss_ready = true;
  return 0;
}

SsimParams mk(const float* x, int ldx, const float* y, int ldy, int B, int H, int W, int C, float data_range) {
  SsimParams p;
  p.x = x; p.y = y; p.ldx = ldx; p.ldy = ldy; p.B = B; p.H = H; p.W = W; p.C = C; p.OH = H - KW + 1; p.OW = W - KW + 1;
  p.C1 = (0.01f * data_range) * (0.01f * data_range); p.C2 = (0.03f * data_range) * (0.03f * data_range);
  return p;
}

}  // namespace

#define ST ((hipStream_t)stream)

// Call once outside any graph capture (uploads the 11-tap window).
extern "C" int clc_ssim_init(void) {
  CLC_CHECK(ensure_gauss() == 0, "clc_ssim_init: hipMemcpyToSymbol failed");
  return 0;
}

extern "C" size_t clc_ssim_workspace_bytes(int B, int H, int W, int C) {
  const int OH = H - KW + 1, OW = W - KW + 1;
  const size_t tiles = (size_t)((OW + TS - 1) / TS) * ((OH + TS - 1) / TS);
  const size_t fwd = (size_t)B * C * tiles * 2;
  const size_t bwd = (size_t)3 * B * OH * OW * C;
  return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

// means: [B*C][2] = (mean cs, mean ssim) of one scale
extern "C" int clc_ssim_scale_fwd(const float* x, int ldx, const float* y, int ldy, int B, int H, int W, int C, float data_range, float* means,
                                  void* ws, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(x && y && means && B > 0 && C > 0 && H >= KW && W >= KW, "clc_ssim_scale_fwd: bad args (image side must be >= 11)");
  CLC_CHECK(g_gauss_ready, "clc_ssim_scale_fwd: call clc_ssim_init() first");
  CLC_CHECK(ws && ws_bytes >= clc_ssim_workspace_bytes(B, H, W, C), "clc_ssim_scale_fwd: workspace too small");
  const SsimParams p = mk(x, ldx, y, ldy, B, H, W, C, data_range);
  dim3 grid((p.OW + TS - 1) / TS, (p.OH + TS - 1) / TS, B * C);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, ST, p, (float*)ws);
  CLC_LAUNCH_CHECK();
  hipLaunchKernelGGL(ssim_reduce_kernel, dim3((B * C * 2 + 63) / 64), dim3(64), 0, ST, (const float*)ws, (int)(grid.x * grid.y),
                     1.f / ((float)p.OH * (float)p.OW), means, B * C);
  CLC_LAUNCH_CHECK();
  return 0;
}

// dx (+)= gradient of sum_bc (g_means[bc][0]*mean_cs + g_means[bc][1]*mean_ssim) wrt x, plus 0.25*upsampled dnext
extern "C" int clc_ssim_scale_bwd(const float* x, int ldx, const float* y, int ldy, int B, int H, int W, int C, float data_range,
                                  const float* g_means, const float* dnext, float* dx, int lddx, void* ws, size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(x && y && g_means && dx && B > 0 && C > 0 && H >= KW && W >= KW, "clc_ssim_scale_bwd: bad args");
  CLC_CHECK(!dnext || (H % 2 == 0 && W % 2 == 0), "clc_ssim_scale_bwd: odd image side with a coarser scale");
  CLC_CHECK(g_gauss_ready, "clc_ssim_scale_bwd: call clc_ssim_init() first");
  CLC_CHECK(ws && ws_bytes >= clc_ssim_workspace_bytes(B, H, W, C), "clc_ssim_scale_bwd: workspace too small");
  const SsimParams p = mk(x, ldx, y, ldy, B, H, W, C, data_range);
  dim3 g1((p.OW + TS - 1) / TS, (p.OH + TS - 1) / TS, B * C);
  hipLaunchKernelGGL(ssim_bwd_coef_kernel, g1, dim3(256), 0, ST, p, g_means, 1.f / ((float)p.OH * (float)p.OW), (float*)ws);
  CLC_LAUNCH_CHECK();
  dim3 g2((W + TS - 1) / TS, (H + TS - 1) / TS, B * C);
  hipLaunchKernelGGL(ssim_bwd_adjoint_kernel, g2, dim3(256), 0, ST, p, (const float*)ws, dnext, dx, lddx);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_avgpool2(const float* x, int ldx, float* out, int B, int H, int W, int C, clc_stream_t stream) {
  CLC_CHECK(x && out && B > 0 && H % 2 == 0 && W % 2 == 0 && C > 0, "clc_avgpool2: bad args (even sizes only)");
  const long total = (long)B * (H / 2) * (W / 2) * C;
  long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(pool2_kernel, dim3((unsigned)nb), dim3(256), 0, ST, x, ldx, out, B, H, W, C);
  CLC_LAUNCH_CHECK();
  return 0;
}
