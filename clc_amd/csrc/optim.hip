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
                                                        float* __restrict__ partials, float grad_scale) {
  __shared__ float sm[4];
  const int2 ch = chunks[blockIdx.x];
  const clc_param_entry e = table[ch.x];
  const long beg = (long)ch.y, end = min(e.n, beg + kChunk);
  float s = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) { const float g = e.g[i] * grad_scale; s = fmaf(g, g, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(const clc_param_entry* __restrict__ table, const int2* __restrict__ chunks,
                                                  const float* __restrict__ total_sqnorm, float max_norm, const float* __restrict__ lr_dev, float omb1,
                                                  float beta2, float omb2, float eps, float wd, const float* __restrict__ step, float grad_scale) {
  const int2 ch = chunks[blockIdx.x];
  const clc_param_entry e = table[ch.x];
  const long beg = (long)ch.y, end = min(e.n, beg + kChunk);
  float clip = 1.f;
  if (max_norm > 0.f && total_sqnorm) {
    const float c = max_norm / (sqrtf(total_sqnorm[0]) + 1e-6f);
    clip = c < 1.f ? c : (c >= 1.f ? 1.f : c);  // NaN propagates like torch.clamp(max=1)
  }
  const float lr = lr_dev[0];   // device-resident: a captured hipGraph follows the learning-rate schedule (MultiStepLR, train_CLC.py:453,497)
  // step = {t, 1 - beta1^t, sqrt(1 - beta2^t)}: the bias corrections are evaluated once per step in double (adam_tick_kernel),
  // as torch.optim.AdamW's host code does, not 49 M times in float
  const float step_size = lr / step[1], bc2_sqrt = step[2];
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    float g = (e.g[i] * grad_scale) * clip;   // grad_scale = 1 / world: the rank mean of a summed all-reduce, folded in here
    if (isnan(g)) g = 0.f; else if (isinf(g)) g = g > 0.f ? 3.402823466e+38f : -3.402823466e+38f;  // nan_to_num_
    float p = e.p[i] * (1.f - lr * wd);
    const float m0 = e.m[i];
    const float m = m0 + omb1 * (g - m0);            // exp_avg.lerp_(grad, 1 - beta1)
    const float v = beta2 * e.v[i] + omb2 * g * g;   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));   // addcdiv_(m, sqrt(v)/sqrt(bc2) + eps, value=-step_size)
    e.p[i] = p; e.m[i] = m; e.v[i] = v; e.g[i] = g;
  }
}

__global__ void scalar_add_kernel(float* x, float v) { x[0] += v; }

__global__ void adam_tick_kernel(float* state, double beta1, double beta2) {
  const float t = state[0] + 1.f;
  state[0] = t;
  state[1] = (float)(1.0 - pow(beta1, (double)t));
  state[2] = (float)sqrt(1.0 - pow(beta2, (double)t));
}
}  // namespace

extern "C" int clc_optim_chunk_elems(void) { return kChunk; }

extern "C" int clc_grad_sqnorm_partials(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, float* partials,
                                        float grad_scale, clc_stream_t stream) {
  CLC_CHECK(table_dev && chunks_dev && partials && n_chunks > 0, "clc_grad_sqnorm_partials: bad args");
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, (const int2*)chunks_dev, partials, grad_scale);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_adamw_step(const clc_param_entry* table_dev, const int32_t* chunks_dev, int n_chunks, const float* total_sqnorm_dev,
                              float max_norm, const float* lr_dev, double beta1, double beta2, float eps, float weight_decay,
                              const float* step_dev, float grad_scale, clc_stream_t stream) {
  CLC_CHECK(table_dev && chunks_dev && step_dev && lr_dev && n_chunks > 0, "clc_adamw_step: bad args");
  hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, (const int2*)chunks_dev, total_sqnorm_dev,
                     max_norm, lr_dev, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, step_dev, grad_scale);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_adam_tick(float* state_dev, double beta1, double beta2, clc_stream_t stream) {
  CLC_CHECK(state_dev, "clc_adam_tick: null");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, beta1, beta2);
  CLC_LAUNCH_CHECK();
  return 0;
}

extern "C" int clc_scalar_add(float* x, float v, clc_stream_t stream) {
  CLC_CHECK(x, "clc_scalar_add: null");
  hipLaunchKernelGGL(scalar_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, v);
  CLC_LAUNCH_CHECK();
  return 0;
}
