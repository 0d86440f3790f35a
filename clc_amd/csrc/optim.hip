// optim.hip — fused multi-tensor optimizer step for the training hot loop (gfx950, HBM-bound).
//
// Replaces, with three launches over ALL parameters, the per-tensor ATen kernels behind
//   torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)      /root/reference/train_CLC.py:164-167
//   p.grad.nan_to_num_() for every parameter                      /root/reference/train_CLC.py:176-178
//   optimizer.step()  (optim.AdamW, lr 1e-4, default betas/eps/wd) /root/reference/train_CLC.py:108-116,179
// Parameters are described by a device-resident table of (param, grad, m, v, numel) entries
// plus a chunk list (entry, offset) so every workgroup streams one 4096-element chunk with
// 16-B accesses where alignment allows.  Algorithmic bytes: 16 B read + 12 B written per
// parameter element.  The squared gradient norm uses per-chunk partials and a fixed-order sum.
#include "common.h"

namespace {
constexpr int kChunk = 4096;

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const clc_param_entry* __restrict__ table, const int2* __restrict__ chunks,
                                                        float* __restrict__ partials) {
  __shared__ float sm[4];
  const int2 ch = chunks[blockIdx.x];
  const clc_param_entry e = table[ch.x];
  const long beg = (long)ch.y, end = min(e.n, beg + kChunk);
  float s = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) { const float g = e.g[i]; s = fmaf(g, g, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(const clc_param_entry* __restrict__ table, const int2* __restrict__ chunks,
                                                  const float* __restrict__ total_sqnorm, float max_norm, float lr, float beta1, float beta2,
                                                  float eps, float wd, const float* __restrict__ step) {
  const int2 ch = chunks[blockIdx.x];
  const clc_param_entry e = table[ch.x];
  const long beg = (long)ch.y, end = min(e.n, beg + kChunk);
  float clip = 1.f;
  if (max_norm > 0.f && total_sqnorm) {
    const float c = max_norm / (sqrtf(total_sqnorm[0]) + 1e-6f);
    clip = c < 1.f ? c : (c >= 1.f ? 1.f : c);  // NaN propagates like torch.clamp(max=1)
  }
  const float t = step[0];
  const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    float g = e.g[i] * clip;
    if (isnan(g)) g = 0.f; else if (isinf(g)) g = g > 0.f ? 3.402823466e+38f : -3.402823466e+38f;  // nan_to_num_
    float p = e.p[i] * (1.f - lr * wd);
    const float m = beta1 * e.m[i] + (1.f - beta1) * g;
    const float v = beta2 * e.v[i] + (1.f - beta2) * g * g;
    p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    e.p[i] = p; e.m[i] = m; e.v[i] = v; e.g[i] = g;
  }
}

__global__ void scalar_add_kernel(float* x, float v) { x[0] += v; }
}  // namespace

extern "C" int clc_optim_chunk_elems(void) { return kChunk; }

extern "C" int clc_grad_sqnorm_partials(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, float* partials,
                                        clc_stream_t stream) {
  CLC_CHECK(table_dev && chunks_dev && partials && n_chunks > 0, "clc_grad_sqnorm_partials: bad args");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, (const int2*)chunks_dev, partials);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_adamw_step(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, const float* total_sqnorm_dev,
                              float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, const float* step_dev,
                              clc_stream_t stream) {
  CLC_CHECK(table_dev && chunks_dev && step_dev && n_chunks > 0, "clc_adamw_step: bad args");
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, (const int2*)chunks_dev, total_sqnorm_dev,
                     max_norm, lr, beta1, beta2, eps, weight_decay, step_dev);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_scalar_add(float* x, float v, clc_stream_t stream) {
  CLC_CHECK(x, "clc_scalar_add: null");
  hipLaunchKernelGGL(scalar_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, v);
  CLC_LAUNCH_CHECK();
  return 0;
}
