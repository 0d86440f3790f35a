// patch_match.hip — patch-matching side-information kernels (numeric core of /root/reference/models/Patch_Matching.py).
//
//   pm_prep            reduce_mean_and_std_normalize_images + rgb_transform (:913-934), fused, planar [N,3,H,W]
//   pm_patch_stats     per query patch: sum x, sum x^2                      (L2_or_pearson_corr :873-884)
//   pm_box_sums        per image position: sum y, sum y^2 over the C x ph x pw window (the two box-filter conv2d's, :868-871,887-889)
//   pm_pearson         Pearson correlation of every query patch with every image position (:854-910) =
//                      conv2d(y, patches) as an implicit GEMM on v_mfma_f32_32x32x2_f32 (M = patches, N = positions,
//                      K = C*ph*pw) with the normalisation, the Gaussian prior mask (:779-807) and the product fused in
//                      the epilogue.  The image window of a workgroup (C x ph x (64+pw-1) floats) is staged once in LDS
//                      and re-used by all K-tiles; patch fragments stream straight from global memory (K-contiguous).
//   pm_topk            top-k positions per patch, ties -> lowest index (torch.argmax / topk, :105,225)
//   pm_gather          softmax(value*temperature)-weighted gather of the k best reference patches, re-tiled into an
//                      image (SI_Wraper :218-240, SI_Finder_at_Image_Domain :106-112)
// Layout here is the reference's planar NCHW (3-channel images; channels_last would leave 12-byte pixels).
#include "common.h"

namespace {

__global__ void pm_prep_kernel(const float* __restrict__ x, float* __restrict__ out, long n_img, long hw, float in_scale) {
  // KITTI statistics hard-coded by the reference (:915-916)
  const float mean[3] = {93.70454143384742f, 98.28243432206516f, 94.84678088809876f};
  const float var[3] = {73.56493292844912f, 75.88547006820752f, 76.74838442810665f};
  const long total = n_img * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    const float* src = x + n * 3 * hw + p;
    const float R = (src[0] * in_scale - mean[0]) / var[0];
    const float G = (src[hw] * in_scale - mean[1]) / var[1];
    const float B = (src[2 * hw] * in_scale - mean[2]) / var[2];
    float* dst = out + n * 3 * hw + p;
    dst[0] = R + G;
    dst[hw] = R - G;
    dst[2 * hw] = 0.5f * (R + B);
  }
}

// one workgroup per patch: centre it (x' = x - mean(x), written to qc) and leave sum x' (~0, kept for exactness) and sum x'^2.
// The correlation's numerator sum x y - mean(y) sum x is then accumulated as sum x' y directly — two O(K) terms that cancel to
// O(sqrt K) no longer meet in fp32 (the uncentred form lost the 3rd decimal of the map: tests allowed 2e-3) — and
// den_x = sum x'^2 needs no subtraction at all.  Fixed-order trees -> reproducible.
__global__ __launch_bounds__(256) void pm_patch_stats_kernel(const float* __restrict__ q, int K, float* __restrict__ qc, float* __restrict__ x_sum,
                                                         float* __restrict__ x_sq) {
  __shared__ float s1[256], s2[256];
  const float* row = q + (size_t)blockIdx.x * K;
  float* crow = qc + (size_t)blockIdx.x * K;
  float a = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) a += row[k];
  s1[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s1[threadIdx.x] += s1[threadIdx.x + o];
    __syncthreads();
  }
  const float mean = s1[0] / (float)K;
  __syncthreads();
  float c1 = 0.f, c2 = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) { const float v = row[k] - mean; crow[k] = v; c1 += v; c2 = fmaf(v, v, c2); }
  s1[threadIdx.x] = c1; s2[threadIdx.x] = c2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { x_sum[blockIdx.x] = s1[0]; x_sq[blockIdx.x] = s2[0]; }
}

// box sums over the C x ph x pw window at every valid position; one thread per position, rows summed via a
// horizontal running window held in registers is overkill here: the image is small and L2-resident.
// S1 = window mean, S2 = sum (y - mean)^2 over the window = sum y^2 - K mean^2, evaluated in double (768-term sums whose
// difference is the variance: in fp32 the subtraction costs 3-4 digits; this kernel is a few microseconds either way).
__global__ void pm_box_sums_kernel(const float* __restrict__ y, int C, int H, int W, int ph, int pw, float* __restrict__ S1, float* __restrict__ S2) {
  const int cw = W - pw + 1, chh = H - ph + 1;
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= cw * chh) return;
  const int oy = pos / cw, ox = pos - oy * cw;
  double a = 0.0, b = 0.0;
  for (int c = 0; c < C; ++c)
    for (int dy = 0; dy < ph; ++dy) {
      const float* r = y + ((size_t)c * H + oy + dy) * W + ox;
      for (int dx = 0; dx < pw; ++dx) { const double v = (double)r[dx]; a += v; b += v * v; }
    }
  const double K = (double)C * ph * pw, mean = a / K;
  S1[pos] = (float)mean; S2[pos] = (float)(b - mean * mean * K);
}

struct PearsonParams {
  const float* q; const float* y; const float* x_sum; const float* x_sq; const float* S1; const float* S2; const float* mask; float* out;
  int P, C, H, W, ph, pw, K, cw, chh;
};

// tile: 64 patches (M) x 64 consecutive positions of one output row (N); 4 waves as 2(M) x 2(N), 32x32 each
__global__ __launch_bounds__(256, 2) void pm_pearson_kernel(const PearsonParams p) {
  extern __shared__ float win[];                       // [C*ph][WW], WW = 64 + pw - 1
  const int WW = 64 + p.pw - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;
  const int segs = (p.cw + 63) / 64;
  const int oy = blockIdx.x / segs, ox0 = (blockIdx.x - oy * segs) * 64;
  const int n0 = blockIdx.y * 64;
  // stage the image window: rows (c, dy), columns ox0 .. ox0+WW-1 (zero past the image edge)
  const int rows = p.C * p.ph;
  for (int e = tid; e < rows * WW; e += 256) {
    const int r = e / WW, cx = e - r * WW;
    const int c = r / p.ph, dy = r - c * p.ph;
    const int xx = ox0 + cx;
    win[e] = (xx < p.W) ? p.y[((size_t)c * p.H + oy + dy) * p.W + xx] : 0.f;
  }
  __syncthreads();
  const unsigned qbytes = (unsigned)((size_t)p.P * p.K * 4);
  const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.q), 0, qbytes, 0x00020000);
  const int patch = n0 + wm * 32 + li;
  const unsigned a_row = (unsigned)patch * (unsigned)p.K;
  const bool a_ok = patch < p.P;
  const int jpos = wn * 32 + li;                          // this lane's position column inside the window
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ktiles = (p.K + 31) / 32;
  for (int kt = 0; kt < ktiles; ++kt) {
    f32x4 af[4], bf[4];
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8) {
      const int k0 = kt * 32 + t8 * 8 + 4 * h;             // 4 consecutive k = 4 consecutive dx of one (c, dy) row (pw % 4 == 0)
      const bool k_ok = k0 < p.K;
      af[t8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, (a_ok && k_ok) ? (a_row + (unsigned)k0) * 4u : 0x80000000u, 0, 0));
      const int r = k0 / p.pw, dx = k0 - r * p.pw;
      const float* wp = win + r * WW + jpos + dx;
      bf[t8] = k_ok ? (f32x4){wp[0], wp[1], wp[2], wp[3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t8 = 0; t8 < 4; ++t8)
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t8][ss], bf[t8][ss], acc, 0, 0, 0);
  }
  // epilogue: D[i = patch][j = position]; col = lane&31 -> position (coalesced stores), row = (r&3)+8*(r>>2)+4h -> patch
  const int ox = ox0 + wn * 32 + li;
  if (ox >= p.cw) return;
  const int pos = oy * p.cw + ox;
  const float psz = (float)p.K;
  const float y_mean = p.S1[pos];          // window mean
  const float den_y = p.S2[pos];           // centred sum of squares of the window
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (n >= p.P) continue;
    const float xs = p.x_sum[n];             // sum of the CENTRED patch (~1e-5: rounding of the mean)
    const float den_x = p.x_sq[n] - (xs / psz) * xs;
    float v = (acc[r] - y_mean * xs) / sqrtf(den_y * den_x);
    if (p.mask) v *= p.mask[(size_t)n * p.cw * p.chh + pos];
    p.out[(size_t)n * p.cw * p.chh + pos] = v;
  }
}

// Gaussian prior masks (create_gaussian_masks :779-807), evaluated in double like the numpy reference
__global__ void pm_gauss_mask_kernel(float* __restrict__ out, int img_h, int img_w, int ph, int pw) {
  const int cw = img_w - pw + 1, chh = img_h - ph + 1;
  const int P = (img_h * img_w) / (ph * pw);
  const long total = (long)P * cw * chh;
  const double patch_img_w = (double)img_w / (double)pw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / ((long)cw * chh));
    const int r = (int)(i - (long)n * cw * chh);
    const int yy = r / cw + ((ph + 1) / 2 - 1), xx = r % cw + ((pw + 1) / 2 - 1);     // crop offsets of the reference
    const double wv = (double)(xx + 1) - (double)(pw % 2) / 2.0, hv = (double)(yy + 1) - (double)(ph % 2) / 2.0;
    const double center_h = (floor((double)n / patch_img_w) + 0.5) * ph;
    const double center_w = (fmod((double)n, patch_img_w) + 0.5) * pw;
    const double sh = 0.5 * img_h, sw = 0.5 * img_w;
    const double g = exp(-4.0 * log(2.0) * ((hv - center_h) * (hv - center_h) / (sh * sh) + (wv - center_w) * (wv - center_w) / (sw * sw)));
    out[i] = (float)g;
  }
}

// top-k (k <= 8) per patch over npos positions; ties -> lowest index. One workgroup per patch.
__global__ __launch_bounds__(256) void pm_topk_kernel(const float* __restrict__ corr, int npos, int k, float* __restrict__ val, int* __restrict__ idx) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* row = corr + (size_t)blockIdx.x * npos;
  float taken_v[8];
  int taken_i[8];
  for (int kk = 0; kk < k; ++kk) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < npos; i += 256) {
      const float v = row[i];
      bool skip = false;
      for (int t = 0; t < kk; ++t) skip = skip || (taken_i[t] == i);
      if (skip) continue;
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        const float v2 = sv[threadIdx.x + o]; const int i2 = si[threadIdx.x + o];
        if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
      }
      __syncthreads();
    }
    taken_v[kk] = sv[0]; taken_i[kk] = si[0];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int kk = 0; kk < k; ++kk) { val[blockIdx.x * k + kk] = taken_v[kk]; idx[blockIdx.x * k + kk] = taken_i[kk]; }
}

// out[c][by*ph+dy][bx*pw+dx] = sum_j w_j * y[c][iy_j+dy][ix_j+dx], w = softmax(val*temperature) over the k candidates
// (temperature < 0 -> k must be 1 and the weight is 1: SI_Finder_at_Image_Domain)
__global__ void pm_gather_kernel(const float* __restrict__ y, int C, int H, int W, int ph, int pw, int cw, const float* __restrict__ val,
                                 const int* __restrict__ idx, int k, float temperature, float* __restrict__ out) {
  const int bw = W / pw;
  const long total = (long)C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), yy = (int)((i / W) % H), c = (int)(i / ((long)W * H));
    const int by = yy / ph, dy = yy - by * ph, bx = x / pw, dx = x - bx * pw;
    const int n = by * bw + bx;
    float acc = 0.f;
    if (temperature < 0.f) {
      const int id = idx[n * k];
      acc = y[((size_t)c * H + id / cw + dy) * W + id % cw + dx];
    } else {
      float mx = -INFINITY;
      for (int j = 0; j < k; ++j) mx = fmaxf(mx, val[n * k + j] * temperature);
      float l = 0.f;
      for (int j = 0; j < k; ++j) l += expf(val[n * k + j] * temperature - mx);
      for (int j = 0; j < k; ++j) {
        const int id = idx[n * k + j];
        const float wgt = expf(val[n * k + j] * temperature - mx) / l;
        acc += wgt * y[((size_t)c * H + id / cw + dy) * W + id % cw + dx];
      }
    }
    out[i] = acc;
  }
}

inline int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int clc_pm_prep(const float* x, float* out, int n_img, int H, int W, float in_scale, clc_stream_t stream) {
  CLC_CHECK(x && out && n_img > 0 && H > 0 && W > 0, "clc_pm_prep: bad args");
  hipLaunchKernelGGL(pm_prep_kernel, dim3(grid_for((long)n_img * H * W)), dim3(256), 0, ST, x, out, (long)n_img, (long)H * W, in_scale);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_pm_gauss_mask(float* out, int img_h, int img_w, int ph, int pw, clc_stream_t stream) {
  CLC_CHECK(out && img_h >= ph && img_w >= pw && ph > 0 && pw > 0 && img_w % pw == 0 && img_h % ph == 0, "clc_pm_gauss_mask: bad args");
  const long total = (long)((img_h * img_w) / (ph * pw)) * (img_w - pw + 1) * (img_h - ph + 1);
  hipLaunchKernelGGL(pm_gauss_mask_kernel, dim3(grid_for(total)), dim3(256), 0, ST, out, img_h, img_w, ph, pw);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t clc_pm_pearson_workspace_bytes(int P, int C, int H, int W, int ph, int pw) {   // statistics + the centred copy of the patches
  return ((size_t)2 * P + (size_t)2 * (H - ph + 1) * (W - pw + 1) + 64 + (size_t)P * C * ph * pw) * sizeof(float);
}

extern "C" int clc_pm_pearson(const float* q, int P, const float* y, int C, int H, int W, int ph, int pw, const float* mask, float* out, void* ws,
                              size_t ws_bytes, clc_stream_t stream) {
  CLC_CHECK(q && y && out && P > 0 && C > 0 && H >= ph && W >= pw && ph > 0 && pw > 0, "clc_pm_pearson: bad args");
  CLC_CHECK(pw % 4 == 0 && aligned16(q), "clc_pm_pearson: patch width must be a multiple of 4 and q 16-byte aligned");
  CLC_CHECK(ws && ws_bytes >= clc_pm_pearson_workspace_bytes(P, C, H, W, ph, pw) && aligned16(ws), "clc_pm_pearson: workspace too small / unaligned");
  const size_t lds = (size_t)C * ph * (64 + pw - 1) * sizeof(float);
  CLC_CHECK(lds <= 64 * 1024, "clc_pm_pearson: window of %zu bytes exceeds the 64 KiB LDS budget", lds);
  CLC_CHECK((size_t)P * C * ph * pw * 4 < (1ull << 31), "clc_pm_pearson: patch tensor too large");
  PearsonParams p;
  p.y = y; p.mask = mask; p.out = out;
  p.P = P; p.C = C; p.H = H; p.W = W; p.ph = ph; p.pw = pw; p.K = C * ph * pw; p.cw = W - pw + 1; p.chh = H - ph + 1;
  float* w = (float*)ws;
  float* x_sum = w; float* x_sq = w + P; float* S1 = w + 2 * P; float* S2 = S1 + (size_t)p.cw * p.chh;
  float* qc = w + (((size_t)2 * P + (size_t)2 * p.cw * p.chh + 63) / 64) * 64;   // centred patches, 256-byte aligned
  p.q = qc; p.x_sum = x_sum; p.x_sq = x_sq; p.S1 = S1; p.S2 = S2;
  hipLaunchKernelGGL(pm_patch_stats_kernel, dim3(P), dim3(256), 0, ST, q, p.K, qc, x_sum, x_sq);
  CLC_LAUNCH_CHECK();
  hipLaunchKernelGGL(pm_box_sums_kernel, dim3((p.cw * p.chh + 255) / 256), dim3(256), 0, ST, y, C, H, W, ph, pw, S1, S2);
  CLC_LAUNCH_CHECK();
  dim3 grid(p.chh * ((p.cw + 63) / 64), (P + 63) / 64);
  hipLaunchKernelGGL(pm_pearson_kernel, grid, dim3(256), lds, ST, p);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_pm_topk(const float* corr, int P, int npos, int k, float* val, int32_t* idx, clc_stream_t stream) {
  CLC_CHECK(corr && val && idx && P > 0 && npos > 0 && k >= 1 && k <= 8 && k <= npos, "clc_pm_topk: bad args (1 <= k <= 8)");
  hipLaunchKernelGGL(pm_topk_kernel, dim3(P), dim3(256), 0, ST, corr, npos, k, val, idx);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_pm_gather(const float* y, int C, int H, int W, int ph, int pw, const float* val, const int32_t* idx, int k, float temperature,
                             float* out, clc_stream_t stream) {
  CLC_CHECK(y && idx && out && C > 0 && H % ph == 0 && W % pw == 0 && k >= 1 && k <= 8, "clc_pm_gather: bad args");
  CLC_CHECK(temperature < 0.f ? k == 1 : val != nullptr, "clc_pm_gather: weighted mode needs the top-k values");
  hipLaunchKernelGGL(pm_gather_kernel, dim3(grid_for((long)C * H * W)), dim3(256), 0, ST, y, C, H, W, ph, pw, W - pw + 1, val, idx, k, temperature, out);
  CLC_LAUNCH_CHECK();
  return 0;
}
