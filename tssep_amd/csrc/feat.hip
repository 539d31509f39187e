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
// The maximum of Log1pMaxNormAbsSTFT follows its `statistics_axis` (feature_extractor.py:239-242): 'tf' = one per
// utterance (every shipped config), 't' = one per utterance and frequency (over the frames), 'f' = one per frame.
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

constexpr int MAXF = 1032;   // LDS line per wave / filter row pitch (F <= 1025)
constexpr int GSLOTS = 64;   // the batch-global dB maximum is collected in 64 slots, merged by pass 2

struct FeatWs {            // layout of the caller's workspace
  unsigned* gmax;          // [GSLOTS] ordered bits of max dB (a single address serialised 97 k atomics: 1 ms)
  unsigned* umax;          // [nmax] ordered bits of max |X|: per utterance (nmax = B), per (utterance, bin) (B F), unused (per frame)
  int* range;              // [2*n_mels]
  float* fbT;              // [n_mels][MAXF]  filterbank transposed (a filter's support contiguous)
  float* db;               // [B*T*n_mels]
};
constexpr int STAT_TF = 0, STAT_T = 1, STAT_F = 2;
__host__ __device__ inline int64_t feat_nmax(int64_t B, int F, int stat) { return stat == STAT_T ? B * F : B; }
__host__ __device__ inline FeatWs feat_ws(void* ws, int64_t B /* = feat_nmax(...) */, int n_mels) {
  FeatWs w;
  char* p = (char*)ws;
  w.gmax = (unsigned*)p;
  w.umax = (unsigned*)(p + GSLOTS * 4);
  const int64_t off1 = GSLOTS * 4 + ((B * 4 + 15) / 16) * 16;
  w.range = (int*)(p + off1);
  const int64_t off2 = off1 + (((int64_t)2 * n_mels * 4 + 15) / 16) * 16;
  w.fbT = (float*)(p + off2);
  w.db = (float*)(p + off2 + (int64_t)n_mels * MAXF * 4);
  return w;
}

// one wave per mel filter: support [lo,hi) of the filter and its transposed row; block 0 also
// resets the maxima (a thread-per-filter serial scan of the dense matrix took 190 us)
__global__ __launch_bounds__(64) void feat_init_kernel(void* ws, int64_t B, const float* __restrict__ fb,
                                                       int F, int n_mels) {
  FeatWs w = feat_ws(ws, B, n_mels);
  const int lane = threadIdx.x, m = blockIdx.x;
  if (m == 0) w.gmax[lane] = 0u;  // GSLOTS == 64 == wave size
  for (int64_t b = (int64_t)m * 64 + lane; b < B; b += (int64_t)gridDim.x * 64) w.umax[b] = 0u;
  if (m >= n_mels) return;
  float lo = (float)F, hi = 0.f;
  for (int f = lane; f < F; f += 64) {
    const float v = fb[(int64_t)f * n_mels + m];
    w.fbT[(int64_t)m * MAXF + f] = v;
    if (v != 0.f) { lo = fminf(lo, (float)f); hi = fmaxf(hi, (float)(f + 1)); }
  }
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  if (lane == 0) {
    w.range[2 * m] = hi == 0.f ? 0 : (int)lo;
    w.range[2 * m + 1] = (int)hi;
  }
}

// pass 1.  A block prepares the mel filterbank ONCE in LDS -- the filters' supports [lo, hi) and their
// weights packed back to back (triangular htk filters overlap pairwise: < 2 F weights in total) -- and
// then walks FPW frames per wave.  (Round 1 read range and weights of each of the 40 filters from global
// memory inside every frame's loop: 80 dependent load round trips per frame, 0.7 TB/s.)  The sums keep
// their order (lane-strided partial sums, butterfly): results are bit-identical to round 1.
constexpr int FPW = 4;            // frames per wave and block
constexpr int CWMAX = 2 * MAXF;   // packed filter weights
constexpr int LMELS = 128;        // filters whose table fits in LDS (40 in every shipped config)

__global__ __launch_bounds__(256) void feat_pass1_kernel(const float2* __restrict__ X, int64_t B,
                                                         int64_t T, int F,
                                                         const float* __restrict__ fb, int n_mels,
                                                         int n_mfcc, void* ws, int stat, int log_mels) {
  __shared__ float pw[4][MAXF];
  __shared__ float cw[CWMAX];
  __shared__ int clo[LMELS], chi[LMELS], coff[LMELS + 1];
  __shared__ int s_packed;
  FeatWs w = feat_ws(ws, feat_nmax(B, F, stat), n_mels);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- filterbank -> LDS
  if (n_mfcc > 0) {
    if (tid < n_mels && tid < LMELS) {
      clo[tid] = w.range[2 * tid];
      chi[tid] = w.range[2 * tid + 1];
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      const int nm = n_mels < LMELS ? n_mels : LMELS;
      for (int m = 0; m < nm; ++m) {
        coff[m] = tot;
        const int wd = chi[m] - clo[m];
        tot += wd > 0 ? wd : 0;
      }
      coff[nm] = tot;
      s_packed = (n_mels <= LMELS && tot <= CWMAX) ? 1 : 0;
    }
    __syncthreads();
    if (s_packed)
      for (int m = wave; m < n_mels; m += 4)
        for (int f = clo[m] + lane; f < chi[m]; f += 64) cw[coff[m] + f - clo[m]] = w.fbT[(int64_t)m * MAXF + f];
    __syncthreads();
  }
  const bool packed = n_mfcc > 0 && s_packed;
  float dbmax = -INFINITY;
  for (int it = 0; it < FPW; ++it) {
    const int64_t frame = ((int64_t)blockIdx.x * FPW + it) * 4 + wave;
    if (frame >= B * T) break;                          // (wave-uniform; pw rows are private to a wave)
    const float2* x = X + frame * F;
    float amax = 0.f;
    // all loads of the frame first (F <= 1025: at most 17 per lane), then the arithmetic
    float2 xv[17];
#pragma unroll
    for (int r = 0; r < 17; ++r) {
      const int f = lane + 64 * r;
      xv[r] = f < F ? x[f] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 17; ++r) {
      const int f = lane + 64 * r;
      if (f < F) {
        const float p = xv[r].x * xv[r].x + xv[r].y * xv[r].y;
        pw[wave][f] = p;
        amax = fmaxf(amax, sqrtf(p));
        if (stat == STAT_T) {      // per (utterance, bin) maximum over the frames: most frames do not raise it
          unsigned* slot = w.umax + (frame / T) * F + f;
          const unsigned o_ = ord(sqrtf(p));
          if (o_ > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, o_);
        }
      }
    }
    // pw[wave] is private to the wave: LDS operations of one wave execute in order, a wave-level fence
    // keeps the compiler from reordering the row's writes and the band reads below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    amax = wave_max(amax);
    if (lane == 0 && stat == STAT_TF) atomicMax(w.umax + frame / T, ord(amax));
    if (n_mfcc > 0) {
      // one lane per filter walks its own band front to back (<= 64 bins for the widest filters): ~3 instructions per
      // bin on 40 lanes.  (Rounds 1-2: all 64 lanes shared each band and a butterfly summed them -- ~25 instructions
      // per FILTER and frame, 1000 per frame: the kernel was VALU-bound at a quarter of the HBM rate.)
      for (int m0 = 0; m0 < n_mels; m0 += 64) {
        float mine = 0.f;
        const int mm = n_mels - m0 < 64 ? n_mels - m0 : 64;
        if (lane < mm) {
          const int m = m0 + lane;
          // four interleaved partial sums (bins f, f+1, f+2, f+3 mod 4 from the band's start): eight LDS reads in
          // flight per trip instead of a dependent read -> fma chain per bin
          const int lo = packed ? clo[m] : w.range[2 * m], hi = packed ? chi[m] : w.range[2 * m + 1];
          const float* frow = packed ? cw + coff[m] - lo : w.fbT + (int64_t)m * MAXF;
          const float* prow = pw[wave];
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
          int f = lo;
          for (; f + 3 < hi; f += 4) {
            s0 = fmaf(prow[f], frow[f], s0);
            s1 = fmaf(prow[f + 1], frow[f + 1], s1);
            s2 = fmaf(prow[f + 2], frow[f + 2], s2);
            s3 = fmaf(prow[f + 3], frow[f + 3], s3);
          }
          if (f < hi) s0 = fmaf(prow[f], frow[f], s0);
          if (f + 1 < hi) s1 = fmaf(prow[f + 1], frow[f + 1], s1);
          if (f + 2 < hi) s2 = fmaf(prow[f + 2], frow[f + 2], s2);
          mine = (s0 + s1) + (s2 + s3);
        }
        if (lane < mm) {
          // AmplitudeToDB('power') -- or, `log_mels` (feature_extractor_torchaudio.py:98-100): log(mel + 1e-6), no floor
          const float db = log_mels ? logf(mine + 1e-6f) : 10.0f * log10f(fmaxf(mine, 1e-10f));
          w.db[frame * n_mels + m0 + lane] = db;
          dbmax = fmaxf(dbmax, db);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // band reads done before the next frame's row
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (n_mfcc > 0) {
    dbmax = wave_max(dbmax);
    if (lane == 0 && dbmax > -INFINITY) atomicMax(w.gmax + (blockIdx.x & (GSLOTS - 1)), ord(dbmax));
  }
}

__global__ __launch_bounds__(256) void feat_pass2_kernel(const float2* __restrict__ X, int64_t B,
                                                         int64_t T, int F,
                                                         const float* __restrict__ dct, int n_mels,
                                                         int n_mfcc, float top_db,
                                                         float* __restrict__ out, int64_t ld_out,
                                                         const void* ws, int stat) {
  __shared__ float dbl[4][64];
  FeatWs w = feat_ws(const_cast<void*>(ws), feat_nmax(B, F, stat), n_mels);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t frame = (int64_t)blockIdx.x * 4 + wave;
  const bool valid = frame < B * T;
  float* o = out + (valid ? frame : 0) * ld_out;
  if (n_mfcc > 0) {
    // (top_db < 0 = log_mels: no floor)
    const float floor_db = top_db < 0.f ? -INFINITY : wave_max(unord(w.gmax[lane])) - top_db;      // merge the 64 slots
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
  const float2* x = X + frame * F;
  float2 xv[17];                      // all loads of the frame first (F <= 1025), then the arithmetic
#pragma unroll
  for (int r = 0; r < 17; ++r) {
    const int f = lane + 64 * r;
    xv[r] = f < F ? x[f] : make_float2(0.f, 0.f);
  }
  if (stat == STAT_TF) {
    const float scale = (float)(M_E - 1.0) / unord(w.umax[frame / T]);
#pragma unroll
    for (int r = 0; r < 17; ++r) {
      const int f = lane + 64 * r;
      if (f < F) o[n_mfcc + f] = log1pf(sqrtf(xv[r].x * xv[r].x + xv[r].y * xv[r].y) * scale);
    }
    return;
  }
  // statistics_axis 'f': the frame's own maximum; 't': the (utterance, bin) maximum collected by pass 1
  float fmx = 0.f;
  if (stat == STAT_F) {
#pragma unroll
    for (int r = 0; r < 17; ++r)
      if (lane + 64 * r < F) fmx = fmaxf(fmx, sqrtf(xv[r].x * xv[r].x + xv[r].y * xv[r].y));
    fmx = wave_max(fmx);
  }
#pragma unroll
  for (int r = 0; r < 17; ++r) {
    const int f = lane + 64 * r;
    if (f < F) {
      const float norm = stat == STAT_F ? fmx : unord(w.umax[(frame / T) * F + f]);
      o[n_mfcc + f] = log1pf(sqrtf(xv[r].x * xv[r].x + xv[r].y * xv[r].y) * ((float)(M_E - 1.0) / norm));
    }
  }
}

}  // namespace

extern "C" int64_t tssep_feat_workspace_bytes(int64_t B, int64_t T, int n_mels, int F, int stat_axis) {
  if (stat_axis < 0 || stat_axis > 2 || F <= 0) return 0;
  FeatWs w = feat_ws(nullptr, feat_nmax(B, F, stat_axis), n_mels);
  return (int64_t)((char*)w.db - (char*)nullptr) + B * T * (int64_t)n_mels * 4 + 16;       // incl. fbT
}

extern "C" int tssep_feat_fwd(const float* X, int64_t B, int64_t T, int F, const float* fb,
                              const float* dct, int n_mels, int n_mfcc, float top_db, int stat_axis,
                              float* out, int64_t ld_out, void* ws, void* stream) {
  if (!X || !out || !ws) return TSSEP_E_NULL;
  if (n_mfcc > 0 && (!fb || !dct)) return TSSEP_E_NULL;
  if (B <= 0 || T <= 0 || F <= 0 || F > MAXF || ld_out < n_mfcc + F) return TSSEP_E_SHAPE;
  if (n_mfcc > 0 && (n_mels <= 0 || n_mels > 1024)) return TSSEP_E_SHAPE;
  if (n_mfcc == 0) n_mels = 0;
  if (stat_axis < 0 || stat_axis > 2) return TSSEP_E_SHAPE;
  if ((((uintptr_t)X) & 7u) || (((uintptr_t)ws) & 15u)) return TSSEP_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int nm = n_mels > 0 ? n_mels : 1;
  const int64_t nmax = feat_nmax(B, F, stat_axis);
  const int64_t nb0 = (nmax + 64 * 16 - 1) / (64 * 16);                  // (the maxima are zeroed by the same launch)
  hipLaunchKernelGGL(feat_init_kernel, dim3((unsigned)(nb0 > nm ? (nb0 > 4096 ? 4096 : nb0) : nm)), dim3(64), 0, s, ws, nmax, fb, F, n_mels);
  const unsigned blocks = (unsigned)((B * T + 3) / 4);
  const unsigned blocks1 = (unsigned)((B * T + 4 * FPW - 1) / (4 * FPW));
  hipLaunchKernelGGL(feat_pass1_kernel, dim3(blocks1), dim3(256), 0, s, (const float2*)X, B, T, F,
                     fb, n_mels, n_mfcc, ws, stat_axis, top_db < 0.f ? 1 : 0);
  hipLaunchKernelGGL(feat_pass2_kernel, dim3(blocks), dim3(256), 0, s, (const float2*)X, B, T, F,
                     dct, n_mels, n_mfcc, top_db, out, ld_out, ws, stat_axis);
  return tssep_launch_status();
}
