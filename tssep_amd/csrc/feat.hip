// Feature extraction: ConcaternatedSTFTFeatures(TorchMFCC, Log1pMaxNormAbsSTFT)
// (tssep/train/feature_extractor.py:352-360, feature_extractor_torchaudio.py:93-106,
//  feature_extractor.py:233-248).  Forward only: no parameter precedes the features.
//
// Two streaming passes over X[B,T,F] (complex64), one wave per frame:
//   pass 1: power spectrum -> sparse mel filterbank (each triangular filter only spans a few
//           dozen bins; the [lo,hi) support of every filter is found once from the dense matrix)
//           -> dB, global max (AmplitudeToDB's top_db floor is over the whole batch) and the
//           per-utterance max |X| (atomicMax on the ordered bit pattern: exact, order free)
//   pass 2: dB floor, 40x40 DCT, log1p(|X| (e-1)/max) -> out[B,T, n_mfcc + F]
#include <math.h>
#include "common.h"

namespace {

__device__ __forceinline__ unsigned ord(float f) {      // monotone float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct FeatWs {            // layout of the caller's workspace
  unsigned* gmax;          // [1]   ordered bits of max dB
  unsigned* umax;          // [B]   ordered bits of max |X| per utterance
  int* range;              // [2*n_mels]
  float* db;               // [B*T*n_mels]
};
__host__ __device__ inline FeatWs feat_ws(void* ws, int64_t B, int n_mels) {
  FeatWs w;
  char* p = (char*)ws;
  w.gmax = (unsigned*)p;
  w.umax = (unsigned*)(p + 16);
  const int64_t off1 = 16 + ((B * 4 + 15) / 16) * 16;
  w.range = (int*)(p + off1);
  const int64_t off2 = off1 + (((int64_t)2 * n_mels * 4 + 15) / 16) * 16;
  w.db = (float*)(p + off2);
  return w;
}

__global__ void feat_init_kernel(void* ws, int64_t B, const float* __restrict__ fb, int F,
                                 int n_mels) {
  FeatWs w = feat_ws(ws, B, n_mels);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) *w.gmax = 0u;
  for (int64_t b = tid; b < B; b += (int64_t)gridDim.x * blockDim.x) w.umax[b] = 0u;
  if (tid < n_mels) {
    int lo = F, hi = 0;
    for (int f = 0; f < F; ++f)
      if (fb[(int64_t)f * n_mels + tid] != 0.f) { if (f < lo) lo = f; hi = f + 1; }
    if (hi == 0) lo = 0;
    w.range[2 * tid] = lo;
    w.range[2 * tid + 1] = hi;
  }
}

constexpr int MAXF = 1032;   // LDS line per wave (F <= 1025)

__global__ __launch_bounds__(256) void feat_pass1_kernel(const float2* __restrict__ X, int64_t B,
                                                         int64_t T, int F,
                                                         const float* __restrict__ fb, int n_mels,
                                                         int n_mfcc, void* ws) {
  __shared__ float pw[4][MAXF];
  FeatWs w = feat_ws(ws, B, n_mels);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t frame = (int64_t)blockIdx.x * 4 + wave;
  const bool valid = frame < B * T;
  float amax = 0.f;
  if (valid) {
    const float2* x = X + frame * F;
    for (int f = lane; f < F; f += 64) {
      const float2 v = x[f];
      const float p = v.x * v.x + v.y * v.y;
      pw[wave][f] = p;
      amax = fmaxf(amax, sqrtf(p));
    }
  }
  __syncthreads();
  if (!valid) return;
  amax = wave_max(amax);
  if (lane == 0) atomicMax(w.umax + frame / T, ord(amax));
  if (n_mfcc > 0) {
    float dbmax = -INFINITY;
    for (int m = lane; m < n_mels; m += 64) {
      const int lo = w.range[2 * m], hi = w.range[2 * m + 1];
      float s = 0.f;
      for (int f = lo; f < hi; ++f) s = fmaf(pw[wave][f], fb[(int64_t)f * n_mels + m], s);
      const float db = 10.0f * log10f(fmaxf(s, 1e-10f));
      w.db[frame * n_mels + m] = db;
      dbmax = fmaxf(dbmax, db);
    }
    dbmax = wave_max(dbmax);
    if (lane == 0) atomicMax(w.gmax, ord(dbmax));
  }
}

__global__ __launch_bounds__(256) void feat_pass2_kernel(const float2* __restrict__ X, int64_t B,
                                                         int64_t T, int F,
                                                         const float* __restrict__ dct, int n_mels,
                                                         int n_mfcc, float top_db,
                                                         float* __restrict__ out, int64_t ld_out,
                                                         const void* ws) {
  __shared__ float dbl[4][64];
  FeatWs w = feat_ws(const_cast<void*>(ws), B, n_mels);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t frame = (int64_t)blockIdx.x * 4 + wave;
  const bool valid = frame < B * T;
  float* o = out + (valid ? frame : 0) * ld_out;
  if (n_mfcc > 0) {
    const float floor_db = unord(*w.gmax) - top_db;
    for (int m0 = 0; m0 < n_mels; m0 += 64) {   // n_mels <= 64 in every config; loop kept general
      if (valid && m0 + lane < n_mels)
        dbl[wave][lane] = fmaxf(w.db[frame * n_mels + m0 + lane], floor_db);
      __syncthreads();
      if (valid) {
        const int mm = n_mels - m0 < 64 ? n_mels - m0 : 64;
        for (int c = lane; c < n_mfcc; c += 64) {
          float s = m0 ? o[c] : 0.f;
          for (int m = 0; m < mm; ++m) s = fmaf(dbl[wave][m], dct[(int64_t)(m0 + m) * n_mfcc + c], s);
          o[c] = s;
        }
      }
      __syncthreads();
    }
  }
  if (!valid) return;
  const float norm = unord(w.umax[frame / T]);
  const float scale = (float)(M_E - 1.0) / norm;
  const float2* x = X + frame * F;
  for (int f = lane; f < F; f += 64) {
    const float2 v = x[f];
    o[n_mfcc + f] = log1pf(sqrtf(v.x * v.x + v.y * v.y) * scale);
  }
}

}  // namespace

extern "C" int64_t tssep_feat_workspace_bytes(int64_t B, int64_t T, int n_mels) {
  FeatWs w = feat_ws(nullptr, B, n_mels);
  return (int64_t)((char*)w.db - (char*)nullptr) + B * T * (int64_t)n_mels * 4 + 16;
}

extern "C" int tssep_feat_fwd(const float* X, int64_t B, int64_t T, int F, const float* fb,
                              const float* dct, int n_mels, int n_mfcc, float top_db, float* out,
                              int64_t ld_out, void* ws, void* stream) {
  if (!X || !out || !ws) return TSSEP_E_NULL;
  if (n_mfcc > 0 && (!fb || !dct)) return TSSEP_E_NULL;
  if (B <= 0 || T <= 0 || F <= 0 || F > MAXF || ld_out < n_mfcc + F) return TSSEP_E_SHAPE;
  if (n_mfcc > 0 && (n_mels <= 0 || n_mels > 1024)) return TSSEP_E_SHAPE;
  if (n_mfcc == 0) n_mels = 0;
  if ((((uintptr_t)X) & 7u) || (((uintptr_t)ws) & 15u)) return TSSEP_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int nm = n_mels > 0 ? n_mels : 1;
  hipLaunchKernelGGL(feat_init_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s, ws, B,
                     fb, F, n_mels);
  const unsigned blocks = (unsigned)((B * T + 3) / 4);
  hipLaunchKernelGGL(feat_pass1_kernel, dim3(blocks), dim3(256), 0, s, (const float2*)X, B, T, F,
                     fb, n_mels, n_mfcc, ws);
  hipLaunchKernelGGL(feat_pass2_kernel, dim3(blocks), dim3(256), 0, s, (const float2*)X, B, T, F,
                     dct, n_mels, n_mfcc, top_db, out, ld_out, ws);
  return tssep_launch_status();
}
