// Streaming (HBM-bound) helper kernels of the TS-SEP hot path: speaker conditioning,
// tanh backward, layout changes, column sums, split reductions and the two losses.
// Reference call sites are cited per entry point in include/tssep_hip.h.
#include <math.h>
#include "common.h"

namespace {

inline unsigned grid_for(int64_t n, int per_block = 256, int64_t cap = 256 * 16) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (unsigned)(b < cap ? b : cap);
}
#define GRID_STRIDE(i, n)                                                           \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);         \
       i += (int64_t)gridDim.x * blockDim.x)

// ---- speaker conditioning -----------------------------------------------------------------
// rows of xs are (b,k,t); pre rows are (b,t); aux rows are (b,k)
// rows of xs are (b, trial, k, t); trial tr puts speaker (k + tr) % K at position k (net.py:913-924)
__device__ __forceinline__ void cond_row(int64_t row, int64_t K, int64_t T, int trials, int64_t& b,
                                         int64_t& t, int64_t& spk) {
  t = row % T;
  const int64_t q = row / T, k = q % K, bt = q / K;
  const int64_t tr = bt % trials;
  b = bt / trials;
  spk = k + tr;
  if (spk >= K) spk -= K;
}
// one wave per output row: the row decomposition (3 divisions) once per row instead of per element
__global__ __launch_bounds__(256) void cond_mul_fwd_kernel(
    const float* __restrict__ pre, int64_t ld_pre, const float* __restrict__ aux, int64_t ld_aux,
    float* __restrict__ xs, int64_t ld_xs, int64_t B, int64_t K, int64_t T, int F, int trials) {
  const int lane = threadIdx.x & 63;
  const int64_t rows = B * trials * K * T;
  const bool vec = F <= 1024 && ((ld_pre | ld_aux | ld_xs) & 3) == 0 && ld_pre >= ((F + 3) & ~3) &&
                   ld_aux >= ((F + 3) & ~3) && ld_xs >= ((F + 3) & ~3) &&
                   ((((uintptr_t)pre) | ((uintptr_t)aux) | ((uintptr_t)xs)) & 15) == 0;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    int64_t b, t, spk;
    cond_row(row, K, T, trials, b, t, spk);
    const float* p = pre + (b * T + t) * ld_pre;
    const float* a = aux + (b * K + spk) * ld_aux;
    float* o = xs + row * ld_xs;
    if (vec) {
      // 16-byte accesses, all loads of the row first (F <= 1024: at most 4 quads per lane); the pad
      // columns of a row (ld rounded to 4) are written too: they are zero in both inputs
      const int nq = (F + 3) >> 2;
      f32x4 pv[4], av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = lane + 64 * r;
        if (q < nq) {
          pv[r] = *reinterpret_cast<const f32x4*>(p + 4 * q);
          av[r] = *reinterpret_cast<const f32x4*>(a + 4 * q);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = lane + 64 * r;
        if (q < nq) __builtin_nontemporal_store(pv[r] * av[r], reinterpret_cast<f32x4*>(o + 4 * q));
      }
    } else {
      for (int f = lane; f < F; f += 64) o[f] = p[f] * a[f];
    }
  }
}
// 16-byte variant: a thread owns 4 consecutive features of one (b, t) row and sums over (trial, speaker);
// the K * trials loads are independent of each other (no per-element index arithmetic, 4x fewer requests)
__global__ void cond_mul_bwd_v4_kernel(const float* __restrict__ dxs, int64_t ld_dxs,
                                       const float* __restrict__ aux, int64_t ld_aux,
                                       float* __restrict__ dpre, int64_t ld_dpre, int64_t B, int64_t K,
                                       int64_t T, int F, int trials) {
  const int nq = (F + 3) >> 2;
  const int64_t total = B * T * nq;
  GRID_STRIDE(e, total) {
    const int64_t row = e / nq;
    const int q = (int)(e - row * nq);
    const int64_t t = row % T, b = row / T;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int tr = 0; tr < trials; ++tr) {
#pragma unroll 4
      for (int64_t k = 0; k < K; ++k) {
        int64_t spk = k + tr;
        if (spk >= K) spk -= K;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dxs + (((b * trials + tr) * K + k) * T + t) * ld_dxs + 4 * q);
        const f32x4 a = *reinterpret_cast<const f32x4*>(aux + (b * K + spk) * ld_aux + 4 * q);
        s[0] += d[0] * a[0]; s[1] += d[1] * a[1]; s[2] += d[2] * a[2]; s[3] += d[3] * a[3];
      }
    }
    *reinterpret_cast<f32x4*>(dpre + row * ld_dpre + 4 * q) = s;
  }
}
__global__ void cond_mul_bwd_kernel(const float* __restrict__ dxs, int64_t ld_dxs,
                                    const float* __restrict__ aux, int64_t ld_aux,
                                    float* __restrict__ dpre, int64_t ld_dpre, int64_t B,
                                    int64_t K, int64_t T, int F, int trials) {
  const int64_t total = B * T * F;
  GRID_STRIDE(e, total) {
    const int64_t row = e / F;
    const int f = (int)(e - row * F);
    const int64_t t = row % T, b = row / T;
    float s = 0.f;
    for (int tr = 0; tr < trials; ++tr)
      for (int64_t k = 0; k < K; ++k) {
        int64_t spk = k + tr;
        if (spk >= K) spk -= K;
        s += dxs[(((b * trials + tr) * K + k) * T + t) * ld_dxs + f] * aux[(b * K + spk) * ld_aux + f];
      }
    dpre[row * ld_dpre + f] = s;
  }
}
__global__ void cond_cat_fwd_kernel(const float* __restrict__ pre, int64_t ld_pre,
                                    const float* __restrict__ aux, int64_t ld_aux,
                                    float* __restrict__ xs, int64_t ld_xs, int64_t B, int64_t K,
                                    int64_t T, int F, int E, int trials) {
  const int W = F + E;
  const int64_t total = B * trials * K * T * W;
  GRID_STRIDE(e, total) {
    const int64_t row = e / W;
    const int c = (int)(e - row * W);
    int64_t b, t, spk;
    cond_row(row, K, T, trials, b, t, spk);
    xs[row * ld_xs + c] =
        c < F ? pre[(b * T + t) * ld_pre + c] : aux[(b * K + spk) * ld_aux + (c - F)];
  }
}
__global__ void cond_cat_bwd_kernel(const float* __restrict__ dxs, int64_t ld_dxs,
                                    float* __restrict__ dpre, int64_t ld_dpre, int64_t B,
                                    int64_t K, int64_t T, int F, int trials) {
  const int64_t total = B * T * F;
  GRID_STRIDE(e, total) {
    const int64_t row = e / F;
    const int f = (int)(e - row * F);
    const int64_t t = row % T, b = row / T;
    float s = 0.f;
    for (int64_t q = 0; q < (int64_t)trials * K; ++q) s += dxs[((b * trials * K + q) * T + t) * ld_dxs + f];
    dpre[row * ld_dpre + f] = s;
  }
}

// ---- tanh backward (+ optional layout change) ---------------------------------------------
// dz rows are always (b,k,t) x P.  combined != 0: dy and y live in [B,T,K*P].
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dz, int64_t rows, int64_t P, int64_t K,
                                int64_t T, int combined) {
  const int64_t total = rows * P;
  GRID_STRIDE(e, total) {
    int64_t src = e;
    if (combined) {
      const int64_t row = e / P, p = e - row * P;
      const int64_t t = row % T, bk = row / T, k = bk % K, b = bk / K;
      src = ((b * T + t) * K + k) * P + p;
    }
    const float v = y[src];
    dz[e] = dy[src] * (1.0f - v * v);
  }
}
// P % 4 == 0 and 16-byte aligned buffers: one float4 per thread and iteration
__global__ void tanh_bwd_v4_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ y,
                                   f32x4* __restrict__ dz, int64_t rows, int P4, int K, int T,
                                   int combined) {
  const int64_t total = rows * P4;
  GRID_STRIDE(e, total) {
    int64_t src = e;
    if (combined) {
      const int64_t row = e / P4;
      const int p = (int)(e - row * P4);
      const int64_t bk = row / T;
      const int t = (int)(row - bk * T);
      const int64_t b = bk / K;
      const int k = (int)(bk - b * K);
      src = ((b * T + t) * K + k) * P4 + p;
    }
    const f32x4 v = y[src], d = __builtin_nontemporal_load(dy + src);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = d[i] * (1.0f - v[i] * v[i]);
    dz[e] = o;
  }
}

// ---- column sums (bias gradients), deterministic two-pass ---------------------------------
constexpr int CS_SLABS = 128;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A,
                                                             int64_t M, int64_t N, int64_t lda,
                                                             float* __restrict__ ws) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  const int64_t per = (M + CS_SLABS - 1) / CS_SLABS;
  const int64_t m0 = (int64_t)blockIdx.y * per;
  const int64_t m1 = m0 + per < M ? m0 + per : M;
  float s = 0.f;
  if (n < N) {
    // 4 independent loads in flight per thread (one per loop trip left the slab latency-bound)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t m = m0 + ty;
    for (; m + 12 < m1; m += 16) {
      const float a0 = A[m * lda + n], a1 = A[(m + 4) * lda + n];
      const float a2 = A[(m + 8) * lda + n], a3 = A[(m + 12) * lda + n];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; m < m1; m += 4) s0 += A[m * lda + n];
    s = (s0 + s1) + (s2 + s3);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N)
    ws[(int64_t)blockIdx.y * N + n] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// dst[i] (+)= sum_s src[s*stride + i]
__global__ void reduce_splits_kernel(const float* __restrict__ src, int nsplit, int64_t stride,
                                     int64_t count, float* __restrict__ dst, int accumulate) {
  GRID_STRIDE(i, count) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += src[k * stride + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}

// split-K partials [nsplit][M][ldp] of a weight-gradient GEMM whose B operand carried a virtual ones column:
// columns [0, N) -> dW [M, ld_w] (dense), column N -> the bias gradient db [M] (the column sums of dY).
__global__ void reduce_splits_bias_kernel(const float* __restrict__ src, int nsplit, int64_t stride,
                                          int64_t M, int64_t N, int64_t ldp, float* __restrict__ dw,
                                          int64_t ld_w, float* __restrict__ db, int accumulate) {
  const int64_t total = M * (N + 1);
  GRID_STRIDE(i, total) {
    const int64_t m = i / (N + 1), n = i - m * (N + 1);
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += src[k * stride + m * ldp + n];
    float* d = n < N ? dw + m * ld_w + n : db + m;
    *d = accumulate ? *d + s : s;
  }
}

// ---- LogMAE -------------------------------------------------------------------------------
constexpr int LM_CHUNK = 4096;
__global__ __launch_bounds__(256) void absdiff_partial_kernel(const float* __restrict__ est,
                                                              const float* __restrict__ tgt,
                                                              int64_t N, float* __restrict__ part,
                                                              int nchunks) {
  __shared__ float red[4];
  const int64_t row = blockIdx.y;
  const int64_t n0 = (int64_t)blockIdx.x * LM_CHUNK;
  float s = 0.f;
  for (int i = threadIdx.x; i < LM_CHUNK; i += 256) {
    const int64_t n = n0 + i;
    if (n < N) s += fabsf(est[row * N + n] - tgt[row * N + n]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[row * nchunks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// loss[b] = log10( sum_k (sum_c part[b*K+k][c]) / N ),  sums[b] = the argument of the log
__global__ void logmae_finalize_kernel(const float* __restrict__ part, int64_t B, int64_t K,
                                       int nchunks, int64_t N, float* __restrict__ loss,
                                       float* __restrict__ sums) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float tot = 0.f;
  for (int64_t k = 0; k < K; ++k) {
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[(b * K + k) * nchunks + c];
    tot += s / (float)N;
  }
  sums[b] = tot;
  loss[b] = log10f(tot);
}
__global__ void logmae_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                  const float* __restrict__ sums, const float* __restrict__ gout,
                                  int64_t KN, int64_t N, int64_t total, float* __restrict__ dest) {
  const float ln10 = 2.30258509299404568402f;
  GRID_STRIDE(e, total) {
    const int64_t b = e / KN;
    // sums == NULL: plain MAE (tssep/train/loss.py:214-216), d/d est = gout sign(est - tgt) / N
    const float coef = sums ? gout[b] / ((float)N * ln10 * sums[b]) : gout[b] / (float)N;
    const float d = est[e] - tgt[e];
    dest[e] = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
  }
}

// ---- VAD BCE ------------------------------------------------------------------------------
// one wave per (b,k,t) row: x = mean_f logit ; l = max(x,0) - x*y + log1p(exp(-|x|))
__global__ __launch_bounds__(256) void vadbce_rows_kernel(const float* __restrict__ logit,
                                                          const float* __restrict__ vad,
                                                          int64_t rows, int F,
                                                          float* __restrict__ xmean,
                                                          float* __restrict__ lrow) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s += logit[row * F + f];
  s = wave_sum(s);
  if (lane == 0) {
    const float x = s / (float)F, y = vad[row];
    xmean[row] = x;
    lrow[row] = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  }
}
__global__ void vadbce_finalize_kernel(const float* __restrict__ lrow, int64_t B, int64_t KT,
                                       float* __restrict__ loss) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < KT; i += 256) s += lrow[b * KT + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)KT;
}
__global__ void vadbce_bwd_kernel(const float* __restrict__ xmean, const float* __restrict__ vad,
                                  const float* __restrict__ gout, int64_t KT, int F,
                                  int64_t total, float* __restrict__ dlogit) {
  GRID_STRIDE(e, total) {
    const int64_t row = e / F;
    const int64_t b = row / KT;
    const float x = xmean[row];
    dlogit[e] = gout[b] * (sigmoidf_acc(x) - vad[row]) / ((float)KT * (float)F);
  }
}

// ---- logit layout map: raw GEMM output -> [B, K, T, F] (+ trial mean, + 't' broadcast) ------
// Covers the tail of MaskEstimator_v2.forward: the final einops rearrange / reduce-repeat
// (net.py:631-659), the mean over permutation trials (net.py:928-951) and the speaker
// un-permutation (net.py:957-967).
//   raw index, speakers in columns (ts_vad):  ((b*trials + tr)*T + t) * (K*Fr) + k*Fr + fr
//   raw index, speakers in rows (ts_vad off): ((b*K + k)*T + t) * Fr + fr           (trials == 1)
//   Fr = F ('tf') or 1 ('t', value repeated over frequency)
// Trial tr holds speaker (k + tr) % K at position k; speaker s lands at output index perm[b,s].
struct MapArgs {
  int64_t B, K, T; int F, Fr, trials, spk_rows;
  const int32_t* perm; const int32_t* iperm;
};
__device__ __forceinline__ int64_t raw_index(const MapArgs& a, int64_t b, int tr, int64_t t, int k,
                                             int fr) {
  if (a.spk_rows) return ((b * a.K + k) * a.T + t) * a.Fr + fr;
  return ((b * a.trials + tr) * a.T + t) * (a.K * a.Fr) + (int64_t)k * a.Fr + fr;
}
__global__ void logit_map_fwd_kernel(const float* __restrict__ raw, MapArgs a,
                                     float* __restrict__ out) {
  const int64_t total = a.B * a.K * a.T * a.F;
  const float inv = 1.0f / (float)a.trials;
  GRID_STRIDE(e, total) {
    const int64_t row = e / a.F;
    const int f = (int)(e - row * a.F);
    const int64_t t = row % a.T, bj = row / a.T, j = bj % a.K, b = bj / a.K;
    const int s = a.iperm ? a.iperm[b * a.K + j] : (int)j;
    const int fr = a.Fr == 1 ? 0 : f;
    float acc = 0.f;
    for (int tr = 0; tr < a.trials; ++tr) {
      int k = s - tr;
      if (k < 0) k += (int)a.K;
      acc += raw[raw_index(a, b, tr, t, k, fr)];
    }
    out[e] = a.trials == 1 ? acc : acc * inv;
  }
}
// draw (raw layout) <- dout [B,K,T,F]: one wave per run of F contiguous raw elements (one
// (b, trial, t, k)); the index decomposition once per run instead of five divisions per element
__global__ __launch_bounds__(256) void logit_map_bwd_tf_kernel(const float* __restrict__ dout, MapArgs a,
                                                               float* __restrict__ draw) {
  const int lane = threadIdx.x & 63;
  const int64_t runs = a.B * a.trials * a.T * a.K;
  const float inv = 1.0f / (float)a.trials;
  for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < runs; u += (int64_t)gridDim.x * 4) {
    int64_t b, t; int tr, k;
    if (a.spk_rows) {
      t = u % a.T; const int64_t bk = u / a.T; k = (int)(bk % a.K); b = bk / a.K; tr = 0;
    } else {
      k = (int)(u % a.K); const int64_t row = u / a.K;
      t = row % a.T; const int64_t bt = row / a.T; tr = (int)(bt % a.trials); b = bt / a.trials;
    }
    int s = k + tr;
    if (s >= a.K) s -= (int)a.K;
    const int j = a.perm ? a.perm[b * a.K + s] : s;
    const float* src = dout + ((b * a.K + j) * a.T + t) * a.F;
    float* dst = draw + u * a.F;
    for (int f = lane; f < a.F; f += 64) dst[f] = a.trials == 1 ? src[f] : src[f] * inv;
  }
}
__global__ __launch_bounds__(256) void logit_map_bwd_t_kernel(const float* __restrict__ dout,
                                                              MapArgs a, float* __restrict__ draw) {
  const int lane = threadIdx.x & 63;
  const int64_t total = a.B * a.trials * a.T * a.K;     // raw elements (Fr == 1)
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= total) return;
  int64_t b, t; int tr, k;
  if (a.spk_rows) {
    t = e % a.T; const int64_t bk = e / a.T; k = (int)(bk % a.K); b = bk / a.K; tr = 0;
  } else {
    const int64_t row = e / a.K; k = (int)(e - row * a.K);
    t = row % a.T; const int64_t bt = row / a.T; tr = (int)(bt % a.trials); b = bt / a.trials;
  }
  int s = k + tr;
  if (s >= a.K) s -= (int)a.K;
  const int j = a.perm ? a.perm[b * a.K + s] : s;
  const float* p = dout + ((b * a.K + j) * a.T + t) * a.F;
  float acc = 0.f;
  for (int f = lane; f < a.F; f += 64) acc += p[f];
  acc = wave_sum(acc);
  if (lane == 0) draw[e] = acc / (float)a.trials;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int tssep_cond_mul_fwd(const float* pre, int64_t ld_pre, const float* aux,
                                  int64_t ld_aux, float* xs, int64_t ld_xs, int64_t B, int64_t K,
                                  int64_t T, int F, int trials, void* stream) {
  if (!pre || !aux || !xs) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0 || trials <= 0 || trials > K) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(cond_mul_fwd_kernel, dim3(grid_for(B * trials * K * T * 64)), dim3(256), 0, S_,
                     pre, ld_pre, aux, ld_aux, xs, ld_xs, B, K, T, F, trials);
  return tssep_launch_status();
}
extern "C" int tssep_cond_mul_bwd(const float* dxs, int64_t ld_dxs, const float* aux,
                                  int64_t ld_aux, float* dpre, int64_t ld_dpre, int64_t B,
                                  int64_t K, int64_t T, int F, int trials, void* stream) {
  if (!dxs || !aux || !dpre) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0 || trials <= 0 || trials > K) return TSSEP_E_SHAPE;
  const int64_t fp = (F + 3) & ~3;
  if (((ld_dxs | ld_aux | ld_dpre) & 3) == 0 && ld_dxs >= fp && ld_aux >= fp && ld_dpre >= fp &&
      ((((uintptr_t)dxs) | ((uintptr_t)aux) | ((uintptr_t)dpre)) & 15) == 0) {
    // (the pad columns of a row are summed too: products of the inputs' pad columns, zero in the step)
    hipLaunchKernelGGL(cond_mul_bwd_v4_kernel, dim3(grid_for(B * T * (fp / 4))), dim3(256), 0, S_, dxs, ld_dxs,
                       aux, ld_aux, dpre, ld_dpre, B, K, T, F, trials);
    return tssep_launch_status();
  }
  hipLaunchKernelGGL(cond_mul_bwd_kernel, dim3(grid_for(B * T * F)), dim3(256), 0, S_, dxs, ld_dxs,
                     aux, ld_aux, dpre, ld_dpre, B, K, T, F, trials);
  return tssep_launch_status();
}
extern "C" int tssep_cond_cat_fwd(const float* pre, int64_t ld_pre, const float* aux,
                                  int64_t ld_aux, float* xs, int64_t ld_xs, int64_t B, int64_t K,
                                  int64_t T, int F, int E, int trials, void* stream) {
  if (!pre || !aux || !xs) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0 || E <= 0 || trials <= 0 || trials > K)
    return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(cond_cat_fwd_kernel, dim3(grid_for(B * trials * K * T * (F + E))), dim3(256),
                     0, S_, pre, ld_pre, aux, ld_aux, xs, ld_xs, B, K, T, F, E, trials);
  return tssep_launch_status();
}
extern "C" int tssep_cond_cat_bwd(const float* dxs, int64_t ld_dxs, float* dpre, int64_t ld_dpre,
                                  int64_t B, int64_t K, int64_t T, int F, int trials, void* stream) {
  if (!dxs || !dpre) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0 || trials <= 0 || trials > K) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(cond_cat_bwd_kernel, dim3(grid_for(B * T * F)), dim3(256), 0, S_, dxs, ld_dxs,
                     dpre, ld_dpre, B, K, T, F, trials);
  return tssep_launch_status();
}
extern "C" int tssep_tanh_bwd(const float* dy, const float* y, float* dz, int64_t rows, int64_t P,
                              int64_t K, int64_t T, int combined_in, void* stream) {
  if (!dy || !y || !dz) return TSSEP_E_NULL;
  if (rows <= 0 || P <= 0 || K <= 0 || T <= 0 || rows % (K * T)) return TSSEP_E_SHAPE;
  if (P % 4 == 0 && aligned16(dy) && aligned16(y) && aligned16(dz))
    hipLaunchKernelGGL(tanh_bwd_v4_kernel, dim3(grid_for(rows * P / 4)), dim3(256), 0, S_,
                       (const f32x4*)dy, (const f32x4*)y, (f32x4*)dz, rows, (int)(P / 4), (int)K, (int)T,
                       combined_in);
  else
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(rows * P)), dim3(256), 0, S_, dy, y, dz, rows,
                       P, K, T, combined_in);
  return tssep_launch_status();
}
extern "C" int64_t tssep_colsum_workspace_bytes(int64_t M, int64_t N) {
  (void)M;
  return (int64_t)CS_SLABS * N * (int64_t)sizeof(float);
}
extern "C" int tssep_colsum_f32(const float* A, int64_t M, int64_t N, int64_t lda, float* out,
                                int accumulate, void* ws, void* stream) {
  if (!A || !out || !ws) return TSSEP_E_NULL;
  if (M <= 0 || N <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 63) / 64), CS_SLABS), dim3(256), 0,
                     S_, A, M, N, lda, (float*)ws);
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(grid_for(N)), dim3(256), 0, S_, (const float*)ws,
                     CS_SLABS, N, N, out, accumulate);
  return tssep_launch_status();
}
extern "C" int tssep_reduce_splits(const float* src, int nsplit, int64_t stride, int64_t count,
                                   float* dst, int accumulate, void* stream) {
  if (!src || !dst) return TSSEP_E_NULL;
  if (nsplit <= 0 || count <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(grid_for(count)), dim3(256), 0, S_, src, nsplit,
                     stride, count, dst, accumulate);
  return tssep_launch_status();
}

extern "C" int tssep_reduce_splits_bias(const float* src, int nsplit, int64_t stride, int64_t M, int64_t N,
                                        int64_t ldp, float* dw, int64_t ld_w, float* db, int accumulate,
                                        void* stream) {
  if (!src || !dw || !db) return TSSEP_E_NULL;
  if (nsplit <= 0 || M <= 0 || N <= 0 || ldp < N + 1 || ld_w < N) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(reduce_splits_bias_kernel, dim3(grid_for(M * (N + 1))), dim3(256), 0, S_, src, nsplit,
                     stride, M, N, ldp, dw, ld_w, db, accumulate);
  return tssep_launch_status();
}

extern "C" int64_t tssep_logmae_chunks(int64_t N) { return (N + LM_CHUNK - 1) / LM_CHUNK; }
extern "C" int64_t tssep_logmae_workspace_bytes(int64_t B, int64_t K, int64_t N) {
  return B * K * tssep_logmae_chunks(N) * (int64_t)sizeof(float);
}
extern "C" int tssep_logmae_finalize(const float* partial, int64_t B, int64_t K, int64_t nchunks,
                                     int64_t N, float* loss, float* sums, void* stream) {
  if (!partial || !loss || !sums) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || nchunks <= 0 || N <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(logmae_finalize_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, S_,
                     partial, B, K, (int)nchunks, N, loss, sums);
  return tssep_launch_status();
}
extern "C" int tssep_logmae_fwd(const float* est, const float* tgt, int64_t B, int64_t K,
                                int64_t N, float* loss, float* sums, void* ws, void* stream) {
  if (!est || !tgt || !loss || !sums || !ws) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || N <= 0 || B * K > 65535) return TSSEP_E_SHAPE;
  const int nchunks = (int)tssep_logmae_chunks(N);
  hipLaunchKernelGGL(absdiff_partial_kernel, dim3((unsigned)nchunks, (unsigned)(B * K)), dim3(256),
                     0, S_, est, tgt, N, (float*)ws, nchunks);
  return tssep_logmae_finalize((const float*)ws, B, K, nchunks, N, loss, sums, stream);
}
extern "C" int tssep_logmae_bwd(const float* est, const float* tgt, const float* sums,
                                const float* gout, int64_t B, int64_t K, int64_t N, float* dest,
                                void* stream) {
  if (!est || !tgt || !gout || !dest) return TSSEP_E_NULL;      // sums may be NULL (MAE)
  if (B <= 0 || K <= 0 || N <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(logmae_bwd_kernel, dim3(grid_for(B * K * N)), dim3(256), 0, S_, est, tgt, sums,
                     gout, K * N, N, B * K * N, dest);
  return tssep_launch_status();
}
extern "C" int64_t tssep_vadbce_workspace_bytes(int64_t B, int64_t K, int64_t T) {
  return B * K * T * (int64_t)sizeof(float);
}
extern "C" int tssep_vadbce_fwd(const float* logit, const float* vad, int64_t B, int64_t K,
                                int64_t T, int F, float* loss, float* xmean, void* ws,
                                void* stream) {
  if (!logit || !vad || !loss || !xmean || !ws) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  const int64_t rows = B * K * T;
  hipLaunchKernelGGL(vadbce_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_, logit,
                     vad, rows, F, xmean, (float*)ws);
  hipLaunchKernelGGL(vadbce_finalize_kernel, dim3((unsigned)B), dim3(256), 0, S_, (const float*)ws,
                     B, K * T, loss);
  return tssep_launch_status();
}
extern "C" int tssep_vadbce_bwd(const float* xmean, const float* vad, const float* gout, int64_t B,
                                int64_t K, int64_t T, int F, float* dlogit, void* stream) {
  if (!xmean || !vad || !gout || !dlogit) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(vadbce_bwd_kernel, dim3(grid_for(B * K * T * F)), dim3(256), 0, S_, xmean, vad,
                     gout, K * T, F, B * K * T * F, dlogit);
  return tssep_launch_status();
}
static int map_args(MapArgs& a, const int32_t* perm, const int32_t* iperm, int64_t B, int trials,
                    int64_t K, int64_t T, int F, int Fr, int spk_rows) {
  if (B <= 0 || K <= 0 || T <= 0 || F <= 0 || trials <= 0 || (Fr != F && Fr != 1))
    return TSSEP_E_SHAPE;
  if (spk_rows && trials != 1) return TSSEP_E_UNSUPPORTED;
  if ((perm == nullptr) != (iperm == nullptr)) return TSSEP_E_NULL;
  a.B = B; a.K = K; a.T = T; a.F = F; a.Fr = Fr; a.trials = trials; a.spk_rows = spk_rows;
  a.perm = perm; a.iperm = iperm;
  return TSSEP_OK;
}
extern "C" int tssep_logit_map_fwd(const float* raw, const int32_t* perm, const int32_t* iperm,
                                   int64_t B, int trials, int64_t K, int64_t T, int F, int Fr,
                                   int spk_rows, float* out, void* stream) {
  if (!raw || !out) return TSSEP_E_NULL;
  MapArgs a;
  if (int e = map_args(a, perm, iperm, B, trials, K, T, F, Fr, spk_rows)) return e;
  hipLaunchKernelGGL(logit_map_fwd_kernel, dim3(grid_for(B * K * T * F)), dim3(256), 0, S_, raw, a,
                     out);
  return tssep_launch_status();
}
extern "C" int tssep_logit_map_bwd(const float* dout, const int32_t* perm, const int32_t* iperm,
                                   int64_t B, int trials, int64_t K, int64_t T, int F, int Fr,
                                   int spk_rows, float* draw, void* stream) {
  if (!dout || !draw) return TSSEP_E_NULL;
  MapArgs a;
  if (int e = map_args(a, perm, iperm, B, trials, K, T, F, Fr, spk_rows)) return e;
  if (Fr == 1 && F != 1) {
    const int64_t total = B * trials * T * K;
    hipLaunchKernelGGL(logit_map_bwd_t_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, S_,
                       dout, a, draw);
  } else {
    hipLaunchKernelGGL(logit_map_bwd_tf_kernel, dim3(grid_for(B * trials * T * K * 64)), dim3(256),
                       0, S_, dout, a, draw);
  }
  return tssep_launch_status();
}
