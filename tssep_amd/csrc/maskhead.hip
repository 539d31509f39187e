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

// Work split: a wave walks 256 consecutive elements per iteration; lane l owns the element pairs
// {2l, 2l+1} and {128+2l, 129+2l}.  Every store instruction of the dominant complex64 stream then
// covers one contiguous 1 KB (16 B per lane, lanes adjacent) instead of 16-byte pieces at a
// 32-byte stride; the fp32 streams move 8 B per lane.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (Round 5: the backward's restructuring -- all loads of an iteration hoisted in front of the arithmetic, two or four halves --
// measured SLOWER for the forward: 4.97 -> 4.47 / 4.39 TB/s; the forward keeps its load -> sigmoid -> store order per half.)
__global__ __launch_bounds__(256) void maskhead_fwd_kernel(
    const float* __restrict__ logit, const float2* __restrict__ obs, float* __restrict__ mask,
    float2* __restrict__ est, int64_t total, int64_t KTF, int64_t TF) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t stride = nwaves * 256;
  const int64_t stride_tf = stride % TF;
  int64_t base = wave * 256;
  if (base >= total) return;
  Pos p0, p1;
  p0.init(base + 2 * lane < total ? base + 2 * lane : 0, KTF, TF);
  p1.init(base + 128 + 2 * lane < total ? base + 128 + 2 * lane : 0, KTF, TF);
  for (; base < total; base += stride, p0.advance(stride, stride_tf, KTF, TF),
                       p1.advance(stride, stride_tf, KTF, TF)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t e = base + 128 * h + 2 * lane;
      Pos q = h ? p1 : p0;
      if (e + 2 <= total) {
        const f32x2 lg = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(logit + e));
        const float m0 = sigmoidf_mask(lg[0]), m1 = sigmoidf_mask(lg[1]);
        const float2 x0 = obs[q.ob + q.tf];
        q.advance(1, 1, KTF, TF);
        const float2 x1 = obs[q.ob + q.tf];
        __builtin_nontemporal_store(f32x2{m0, m1}, reinterpret_cast<f32x2*>(mask + e));
        __builtin_nontemporal_store(f32x4{x0.x * m0, x0.y * m0, x1.x * m1, x1.y * m1},
                                    reinterpret_cast<f32x4*>(est + e));
      } else if (e < total) {
        const float m0 = sigmoidf_mask(logit[e]);
        const float2 x0 = obs[q.ob + q.tf];
        mask[e] = m0;
        est[e] = make_float2(x0.x * m0, x0.y * m0);
      }
    }
  }
}

// (Round 6, BASELINE configs[4] -- 8 speakers, 30-s chunks, stand-alone forward 0.58-0.59 of 8 TB/s against 0.61 at 4 s: a
// variant with the speakers INNERMOST -- a wave keeps the observation bins of 256 (t, f) positions in registers and walks
// the K speakers over them, so that the 7.7-MB observation of an utterance, which no longer fits an XCD's L2, is read once
// per bin -- measured SLOWER, 4.55-4.60 against 4.69 TB/s in three alternating pairs: the re-reads are served on the die
// (Infinity Cache); the forward is bound by its 12 written bytes per element, where a torch copy of the same bytes reaches
// 4.35-4.88 TB/s on these boxes.  profiles/r6_maskhead_kinner_rejected.jsonl)
// Backward: FOUR 128-element halves per wave and iteration with every load of the iteration issued before the first use
// (8 streaming loads + 8 L2-served observation bins per lane in flight), non-temporal on all four streams: 4.71 -> 4.95 TB/s
// at batch 768 (0.59 -> 0.62 of 8 TB/s; a torch copy of the same bytes 4.6-5.1), alternating builds on one box, round 5
// (two halves + nt 4.85, four halves without nt 4.82, 16 workgroups per CU 4.65: profiles/r5_maskhead_bwd_variants.jsonl)
#ifndef MH_HALVES
#define MH_HALVES 4
#endif
#ifndef MH_NT_ALL
#define MH_NT_ALL 1
#endif
__global__ __launch_bounds__(256) void maskhead_bwd_kernel(
    const float2* __restrict__ dest, const float* __restrict__ dmask,
    const float* __restrict__ mask, const float2* __restrict__ obs, float* __restrict__ dlogit,
    int64_t total, int64_t KTF, int64_t TF) {
  constexpr int HV = MH_HALVES, WE = 128 * HV;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t stride = nwaves * WE;
  const int64_t stride_tf = stride % TF;
  int64_t base = wave * WE;
  if (base >= total) return;
  Pos p[HV];
#pragma unroll
  for (int h = 0; h < HV; ++h) p[h].init(base + 128 * h + 2 * lane < total ? base + 128 * h + 2 * lane : 0, KTF, TF);
  for (; base < total; base += stride) {
    // all loads of the iteration first (2 HV streaming loads + 2 HV L2-served observation bins per lane in flight), then
    // the arithmetic and the stores
    f32x4 d[HV];
    f32x2 m[HV], g[HV];
    float2 x0[HV], x1[HV];
    bool full[HV];
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      const int64_t e = base + 128 * h + 2 * lane;
      full[h] = e + 2 <= total;
      Pos q = p[h];
      g[h] = f32x2{0.f, 0.f};
      if (full[h]) {
        d[h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dest + e));
        m[h] = MH_NT_ALL ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(mask + e))
                         : *reinterpret_cast<const f32x2*>(mask + e);
        if (dmask) g[h] = *reinterpret_cast<const f32x2*>(dmask + e);
        x0[h] = obs[q.ob + q.tf];
        q.advance(1, 1, KTF, TF);
        x1[h] = obs[q.ob + q.tf];
      }
    }
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      const int64_t e = base + 128 * h + 2 * lane;
      if (full[h]) {
        const float o0 = (x0[h].x * d[h][0] + x0[h].y * d[h][1] + g[h][0]) * m[h][0] * (1.0f - m[h][0]);
        const float o1 = (x1[h].x * d[h][2] + x1[h].y * d[h][3] + g[h][1]) * m[h][1] * (1.0f - m[h][1]);
        if (MH_NT_ALL) __builtin_nontemporal_store(f32x2{o0, o1}, reinterpret_cast<f32x2*>(dlogit + e));
        else *reinterpret_cast<f32x2*>(dlogit + e) = f32x2{o0, o1};
      } else if (e < total) {
        const Pos q = p[h];
        const float2 xa = obs[q.ob + q.tf], da = dest[e];
        const float ma = mask[e];
        dlogit[e] = (xa.x * da.x + xa.y * da.y + (dmask ? dmask[e] : 0.f)) * ma * (1.0f - ma);
      }
      p[h].advance(stride, stride_tf, KTF, TF);
    }
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
#ifndef MH_WGS_PER_CU
#define MH_WGS_PER_CU 8
#endif
  const int64_t cap = 256 * MH_WGS_PER_CU;  // workgroups per CU, grid-stride beyond
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
