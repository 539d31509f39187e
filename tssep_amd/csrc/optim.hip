// Optimizer step on the flat parameter / gradient buffers: global-norm gradient clipping + Adam in
// two launches, no host synchronisation (the clip factor never leaves the GPU).
//
// Replaces what padertorch's Trainer does per optimizer step with the reference's settings
// (tssep/train/experiment.py:147-150: Adam, gradient_clipping = 10; torch.nn.utils.clip_grad_norm_
// followed by torch.optim.Adam.step over 42 tensors) -- SURVEY.md 8f item 1.
#include <math.h>
#include "common.h"

namespace {

constexpr int SQ_BLOCKS = 512;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n;
       i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// p, m, v, g: flat fp32.  norm_out[0] = ||g||_2 (before clipping).
__global__ __launch_bounds__(256) void adam_step_kernel(
    float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
    const float* __restrict__ g, int64_t n, const float* __restrict__ part, int nparts,
    float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
    float bc1, float bc2, float* __restrict__ norm_out, const int* __restrict__ guard) {
  __shared__ float s_clip;
  // guard[0] != 0: a W-stationary recurrence launch of this step gave up on a peer (err[0] of include/tssep_hip.h) --
  // its outputs and therefore this gradient are garbage: the update is NOT applied (uniform over the grid; the host
  // raises at its next check of the flag, with the parameters and moments of the last good step intact)
  if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  if (threadIdx.x < 64) {        // fixed-order reduction of the partial sums (deterministic)
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      const float norm = sqrtf(s);
      float c = 1.f;
      if (max_norm > 0.f) {
        c = max_norm / (norm + 1e-6f);     // torch.nn.utils.clip_grad_norm_
        c = c < 1.f ? c : 1.f;
      }
      s_clip = c;
      if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
    }
  }
  __syncthreads();
  const float clip = s_clip;
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * clip;
    const float pi = p[i];
    if (weight_decay != 0.f) gi += weight_decay * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    // torch.optim.Adam: denom = sqrt(v)/sqrt(bias_correction2) + eps ; p -= lr/bias_correction1 * m/denom
    p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

}  // namespace

extern "C" int64_t tssep_adam_workspace_bytes(void) { return SQ_BLOCKS * (int64_t)sizeof(float); }

extern "C" int tssep_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* grad,
                               int64_t n, int64_t step, float max_norm, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float* norm_out,
                               void* ws, void* stream) {
  return tssep_adam_step_guarded(param, exp_avg, exp_avg_sq, grad, n, step, max_norm, lr, beta1, beta2, eps, weight_decay,
                                 norm_out, ws, nullptr, stream);
}

extern "C" int tssep_adam_step_guarded(float* param, float* exp_avg, float* exp_avg_sq, const float* grad,
                                       int64_t n, int64_t step, float max_norm, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, float* norm_out,
                                       void* ws, const int* err, void* stream) {
  if (!param || !exp_avg || !exp_avg_sq || !grad || !ws) return TSSEP_E_NULL;
  if (n <= 0 || step <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(grad)) return TSSEP_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SQ_BLOCKS), dim3(256), 0, s, grad, n, (float*)ws);
  const float bc1 = 1.0f - (float)pow((double)beta1, (double)step);
  const float bc2 = 1.0f - (float)pow((double)beta2, (double)step);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, param, exp_avg,
                     exp_avg_sq, grad, n, (const float*)ws, SQ_BLOCKS, max_norm, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2, norm_out, err);
  return tssep_launch_status();
}
