// Mask head: mask = sigmoid(logit); est = Obs * mask, and its backward.
// Replaces torch.nn.Sigmoid (tssep/train/net.py:669,983) + Masking.__call__
// (tssep/train/enhancer.py:98-100).  Pure HBM streaming: per (b,k,t,f) element the
// forward reads 4 B (logit) and writes 4 B (mask) + 8 B (estimate); the 8-B
// observation bin is shared by the K speakers and served from L2 after the first.
//
// Layout: the [B,K,T,F] tensors are walked as a flat array (F = 513 makes rows
// 4-B aligned only), 4 consecutive elements per thread -> 16-B loads/stores on the
// real tensors and 32-B stores on the complex one.
#include "common.h"

namespace {

// Position of flat element e of a [B,K,T,F] tensor relative to the [B,T,F] observation:
// oi = b*TF + (e mod TF).  One 64-bit division per thread, then incremental updates.
struct Pos {
  int64_t r;    // e - b*KTF  in [0, KTF)
  int64_t tf;   // r mod TF
  int64_t ob;   // b*TF
  __device__ __forceinline__ void init(int64_t e, int64_t KTF, int64_t TF) {
    const int64_t b = e / KTF;
    r = e - b * KTF;
    tf = r % TF;
    ob = b * TF;
  }
  __device__ __forceinline__ void advance(int64_t d, int64_t dtf, int64_t KTF, int64_t TF) {
    r += d;
    while (r >= KTF) { r -= KTF; ob += TF; }
    tf += dtf;
    if (tf >= TF) tf -= TF;
  }
};

__global__ __launch_bounds__(256) void maskhead_fwd_kernel(
    const float* __restrict__ logit, const float2* __restrict__ obs, float* __restrict__ mask,
    float2* __restrict__ est, int64_t total, int64_t KTF, int64_t TF) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  const int64_t stride_tf = stride % TF;
  int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e0 >= total) return;
  Pos p;
  p.init(e0, KTF, TF);
  for (; e0 < total; e0 += stride, p.advance(stride, stride_tf, KTF, TF)) {
    const int n = total - e0 >= 4 ? 4 : (int)(total - e0);
    float lg[4], m[4];
    float2 o[4];
    if (n == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(logit + e0);
      lg[0] = v[0]; lg[1] = v[1]; lg[2] = v[2]; lg[3] = v[3];
    } else {
      for (int i = 0; i < 4; ++i) lg[i] = i < n ? logit[e0 + i] : 0.f;
    }
    Pos q = p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[i] = sigmoidf_acc(lg[i]);
      const float2 x = i < n ? obs[q.ob + q.tf] : make_float2(0.f, 0.f);
      o[i] = make_float2(x.x * m[i], x.y * m[i]);
      q.advance(1, 1, KTF, TF);
    }
    if (n == 4) {
      *reinterpret_cast<f32x4*>(mask + e0) = f32x4{m[0], m[1], m[2], m[3]};
      f32x4* ep = reinterpret_cast<f32x4*>(est + e0);
      ep[0] = f32x4{o[0].x, o[0].y, o[1].x, o[1].y};
      ep[1] = f32x4{o[2].x, o[2].y, o[3].x, o[3].y};
    } else {
      for (int i = 0; i < n; ++i) { mask[e0 + i] = m[i]; est[e0 + i] = o[i]; }
    }
  }
}

__global__ __launch_bounds__(256) void maskhead_bwd_kernel(
    const float2* __restrict__ dest, const float* __restrict__ dmask,
    const float* __restrict__ mask, const float2* __restrict__ obs, float* __restrict__ dlogit,
    int64_t total, int64_t KTF, int64_t TF) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  const int64_t stride_tf = stride % TF;
  int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e0 >= total) return;
  Pos p;
  p.init(e0, KTF, TF);
  for (; e0 < total; e0 += stride, p.advance(stride, stride_tf, KTF, TF)) {
    const int n = total - e0 >= 4 ? 4 : (int)(total - e0);
    float m[4], dm[4] = {0.f, 0.f, 0.f, 0.f}, dre[4], dim[4], out[4];
    if (n == 4) {
      const f32x4 mv = *reinterpret_cast<const f32x4*>(mask + e0);
      const f32x4* dp = reinterpret_cast<const f32x4*>(dest + e0);
      const f32x4 d0 = dp[0], d1 = dp[1];
      m[0] = mv[0]; m[1] = mv[1]; m[2] = mv[2]; m[3] = mv[3];
      dre[0] = d0[0]; dim[0] = d0[1]; dre[1] = d0[2]; dim[1] = d0[3];
      dre[2] = d1[0]; dim[2] = d1[1]; dre[3] = d1[2]; dim[3] = d1[3];
      if (dmask) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dmask + e0);
        dm[0] = g[0]; dm[1] = g[1]; dm[2] = g[2]; dm[3] = g[3];
      }
    } else {
      for (int i = 0; i < 4; ++i) {
        const bool ok = i < n;
        m[i] = ok ? mask[e0 + i] : 0.f;
        const float2 d = ok ? dest[e0 + i] : make_float2(0.f, 0.f);
        dre[i] = d.x; dim[i] = d.y;
        dm[i] = (ok && dmask) ? dmask[e0 + i] : 0.f;
      }
    }
    Pos q = p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 x = i < n ? obs[q.ob + q.tf] : make_float2(0.f, 0.f);
      out[i] = (x.x * dre[i] + x.y * dim[i] + dm[i]) * m[i] * (1.0f - m[i]);
      q.advance(1, 1, KTF, TF);
    }
    if (n == 4) *reinterpret_cast<f32x4*>(dlogit + e0) = f32x4{out[0], out[1], out[2], out[3]};
    else for (int i = 0; i < n; ++i) dlogit[e0 + i] = out[i];
  }
}

// Standalone Masking (mask given, not logits): est = Obs * mask and dmask = Re(conj(Obs) dest).
__global__ __launch_bounds__(256) void mask_mul_kernel(
    const float* __restrict__ mask, const float2* __restrict__ dest, const float2* __restrict__ obs,
    float2* __restrict__ est, float* __restrict__ dmask, int64_t total, int64_t KTF, int64_t TF) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t stride_tf = stride % TF;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  Pos p;
  p.init(e, KTF, TF);
  for (; e < total; e += stride, p.advance(stride, stride_tf, KTF, TF)) {
    const float2 x = obs[p.ob + p.tf];
    if (est) {
      const float m = mask[e];
      est[e] = make_float2(x.x * m, x.y * m);
    } else {
      const float2 d = dest[e];
      dmask[e] = x.x * d.x + x.y * d.y;
    }
  }
}

inline unsigned stream_grid(int64_t total_vec4) {
  int64_t blocks = (total_vec4 + 255) / 256;
  const int64_t cap = 256 * 8;  // 8 workgroups per CU, grid-stride beyond
  return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

}  // namespace

extern "C" int tssep_maskhead_fwd(const float* logit, const float* obs, float* mask, float* est,
                                  int64_t B, int64_t K, int64_t T, int F, void* stream) {
  if (!logit || !obs || !mask || !est) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(logit) || !aligned16(mask) || !aligned16(est) || (((uintptr_t)obs) & 7u))
    return TSSEP_E_ALIGN;
  const int64_t TF = T * F, KTF = K * TF, total = B * KTF;
  hipLaunchKernelGGL(maskhead_fwd_kernel, dim3(stream_grid((total + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, logit, (const float2*)obs, mask, (float2*)est, total, KTF,
                     TF);
  return tssep_launch_status();
}

extern "C" int tssep_maskhead_bwd(const float* dest, const float* dmask, const float* mask,
                                  const float* obs, float* dlogit, int64_t B, int64_t K, int64_t T,
                                  int F, void* stream) {
  if (!dest || !mask || !obs || !dlogit) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(dest) || !aligned16(mask) || !aligned16(dlogit) || (dmask && !aligned16(dmask)) ||
      (((uintptr_t)obs) & 7u))
    return TSSEP_E_ALIGN;
  const int64_t TF = T * F, KTF = K * TF, total = B * KTF;
  hipLaunchKernelGGL(maskhead_bwd_kernel, dim3(stream_grid((total + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)dest, dmask, mask, (const float2*)obs,
                     dlogit, total, KTF, TF);
  return tssep_launch_status();
}

extern "C" int tssep_mask_mul_fwd(const float* mask, const float* obs, float* est, int64_t B,
                                  int64_t K, int64_t T, int F, void* stream) {
  if (!mask || !obs || !est) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  const int64_t TF = T * F, KTF = K * TF, total = B * KTF;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     mask, (const float2*)nullptr, (const float2*)obs, (float2*)est,
                     (float*)nullptr, total, KTF, TF);
  return tssep_launch_status();
}
extern "C" int tssep_mask_mul_bwd(const float* dest, const float* obs, float* dmask, int64_t B,
                                  int64_t K, int64_t T, int F, void* stream) {
  if (!dest || !obs || !dmask) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  const int64_t TF = T * F, KTF = K * TF, total = B * KTF;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)nullptr, (const float2*)dest, (const float2*)obs,
                     (float2*)nullptr, dmask, total, KTF, TF);
  return tssep_launch_status();
}
