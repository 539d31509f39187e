// Mask-based MVDR beamformer (Souden) in complex128 -- the eval-time enhancer of the reference,
// TorchBF.__call__ (tssep/train/enhancer.py:215-265):
//   psd[m,k,f]  = sum_t w_m[k,t,f] Y[:,t,f] Y[:,t,f]^H        (torch.einsum, :226-250)
//   phi         = solve(psd[interference], psd[target])        (torch.linalg.solve, :253)
//   bf[k,f,:]   = phi[:, ref] / max(Re trace(phi), eps)        (:254-258)
//   enh[k,t,f]  = sum_d conj(bf[k,f,d]) Y[d,t,f]   (* max(mask, masking_eps))   (:259-264)
//
// All of it is HBM-bound double-precision vector work (no matrix cores: D <= 8 channels), laid out
// with lane = frequency bin so that every load and store of a wave is 512 B - 1 KB contiguous:
//   1. mvdr_psd_kernel<D>     one wave per (64 bins, time chunk, speaker): 2 D*D doubles of
//                             Hermitian accumulators (target, interference) per lane; 4 speakers
//                             per workgroup share the Y frames through LDS; the chunk partials go
//                             to the workspace (fixed summation order -> run-to-run identical).
//                             The workgroups that share one Y tile are mapped to ONE XCD so the
//                             tile is fetched from HBM once and served from that XCD's L2 after.
//   2. mvdr_reduce_kernel     adds the chunk partials (into chunk 0), one lane per element;
//      mvdr_solve_kernel<D>   one lane per (speaker, bin): LU with partial pivoting (LAPACK zgesv's
//                             pivot rule, |re|+|im|), trace, scaling -- in registers for D <= 6,
//      mvdr_weights_kernel    with the D x D systems in LDS ([element][lane]) for D = 7, 8.
//   3. mvdr_apply_kernel<D>   one wave per (64 bins, time chunk, up to 4 speakers): Y read once
//                             for the speakers of a group.
#include "common.h"

namespace {

constexpr int MAXD = 8;
constexpr int KG = 4;                    // speakers per apply workgroup

struct Plan {
  int nf;          // 64-bin tiles
  int chunks;      // time chunks of the PSD pass
  int64_t tchunk;  // frames per chunk
};
// Time chunks of the statistics pass: 512 workgroups are resident at once (2 per CU); pick the chunk
// count that minimises (rounds of resident workgroups) x (frames per chunk) -- one workgroup more
// than a round costs a whole extra round -- with a small charge per chunk for the partials' traffic.
__host__ Plan make_plan(int64_t B, int K, int64_t T, int F) {
  Plan p;
  p.nf = (F + 63) / 64;
  const int64_t per_chunk = B * p.nf * ((K + 3) / 4);        // workgroups of 4 speakers
  int64_t cmax = (T + 15) / 16;
  if (cmax > 256) cmax = 256;
  int64_t best_c = 1;
  double best = 1e300;
  for (int64_t c = 1; c <= cmax; ++c) {
    const int64_t frames = (T + c - 1) / c;
    const int64_t rounds = (per_chunk * c + 511) / 512;
    const double cost = (double)rounds * (double)(frames + 8) + 0.5 * (double)c;
    if (cost < best) { best = cost; best_c = c; }
  }
  p.tchunk = (T + best_c - 1) / best_c;
  p.chunks = (int)((T + p.tchunk - 1) / p.tchunk);
  return p;
}

__device__ __forceinline__ double load_mask(const void* masks, int f64, int64_t i) {
  return f64 ? static_cast<const double*>(masks)[i] : (double)static_cast<const float*>(masks)[i];
}

// XCD-aware block -> (tile, member): consecutive block ids go round-robin over the 8 XCDs, so the
// `members` workgroups that read the same tile get ids  8*(tile/8*members + member) + tile%8
__device__ __forceinline__ bool xcd_tile(int64_t bid, int members, int64_t tiles, int64_t& tile,
                                         int& member) {
  const int xcd = (int)(bid & 7);
  const int64_t within = bid >> 3;
  tile = (within / members) * 8 + xcd;
  member = (int)(within % members);
  return tile < tiles;
}

// One workgroup = 4 waves = 4 speakers (target AND interference statistics each) of one
// (64 bins, time chunk) tile.  Every wave fetches one frame of Y per step and shares it with the
// other three through LDS (double buffered, one barrier per step), so Y crosses the L2 once per 4
// speakers instead of once per (speaker, mask): a first version with one wave per (speaker, mask)
// reading Y straight from L2 was L2-bandwidth bound (16 x 92 MB per 30-s utterance, 1.7 TB/s of
// algorithmic traffic).
constexpr int PW = 4;      // waves (= speakers) per workgroup
template <typename MT> struct MaskRegs { MT w0[PW], w1[PW]; };

template <int D, typename MT>
__global__ __launch_bounds__(64 * PW, D <= 6 ? 2 : 1) void mvdr_psd_kernel(
    const double2* __restrict__ obs, const MT* __restrict__ masks, double* __restrict__ part,
    int64_t B, int K, int M, int64_t T, int F, Plan plan) {
  // frames arrive by asynchronous global -> LDS copies (1 KB per wave instruction, lane-linear:
  // exactly [channel][lane] of double2), no staging registers
  __shared__ double2 ybuf[2][PW][D][64];
  const int groups = (K + PW - 1) / PW;
  int64_t tile;
  int grp;
  if (!xcd_tile(blockIdx.x, groups, B * plan.chunks * plan.nf, tile, grp)) return;
  const int ft = (int)(tile % plan.nf);
  const int c = (int)((tile / plan.nf) % plan.chunks);
  const int64_t b = tile / ((int64_t)plan.nf * plan.chunks);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: uniform branches
  const int k = grp * PW + wave;
  const bool kvalid = k < K;
  const int fraw = ft * 64 + lane;
  const int f = fraw < F ? fraw : F - 1;
  const int64_t t0 = c * plan.tchunk;
  const int64_t t1 = t0 + plan.tchunk < T ? t0 + plan.tchunk : T;
  const int steps = (int)((t1 - t0 + PW - 1) / PW);
  // target: mask 0.  interference: mask 1 if given, else 1 - mask 0 (enhancer.py:236-250)
  const MT* mk0 = masks + ((b * K + (kvalid ? k : 0)) * M) * T * F + f;
  const MT* mk1 = mk0 + (M == 2 ? T * (int64_t)F : 0);
  const double2* y0 = obs + b * D * T * F + f;

  double diag[2][D];
  double2 off[2][D * (D - 1) / 2 + 1];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int i = 0; i < D; ++i) diag[m][i] = 0.0;
#pragma unroll
    for (int i = 0; i < D * (D - 1) / 2; ++i) off[m][i] = double2{0.0, 0.0};
  }
  // wave w brings frame t0 + PW s + w of step s; frames past the chunk are zero-filled
  auto fetch_y = [&](int s_) {
    const int64_t t = t0 + (int64_t)s_ * PW + wave;
    if (t < t1) {
#pragma unroll
      for (int d = 0; d < D; ++d)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(y0 + (d * T + t) * F),
            (__attribute__((address_space(3))) void*)&ybuf[s_ & 1][wave][d][0], 16, 0, 0);
    } else {
#pragma unroll
      for (int d = 0; d < D; ++d) ybuf[s_ & 1][wave][d][lane] = double2{0.0, 0.0};
    }
  };
  auto fetch_m = [&](int s_, MaskRegs<MT>& r) {
    // unconditional loads (a branch per load would serialise them); frames past the chunk read
    // the last frame's mask and meet zero-filled Y
#pragma unroll
    for (int fr = 0; fr < PW; ++fr) {
      int64_t tt = t0 + (int64_t)s_ * PW + fr;
      tt = tt < t1 ? tt : t1 - 1;
      r.w0[fr] = mk0[tt * F];
      r.w1[fr] = mk1[tt * F];
    }
  };
  auto compute = [&](int s_, const MaskRegs<MT>& r) {
#pragma unroll
    for (int fr = 0; fr < PW; ++fr) {
      double2 y[D];
#pragma unroll
      for (int d = 0; d < D; ++d) y[d] = ybuf[s_ & 1][fr][d][lane];
      const double w0 = (double)r.w0[fr];
      const double w1 = M == 2 ? (double)r.w1[fr] : 1.0 - w0;
      // y_i conj(y_j) once, weighted into both accumulators
      int p = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double pd = y[i].x * y[i].x + y[i].y * y[i].y;
        diag[0][i] += w0 * pd;
        diag[1][i] += w1 * pd;
#pragma unroll
        for (int j = i + 1; j < D; ++j, ++p) {
          const double pr = y[i].x * y[j].x + y[i].y * y[j].y;
          const double pi = y[i].y * y[j].x - y[i].x * y[j].y;
          off[0][p].x += w0 * pr;
          off[0][p].y += w0 * pi;
          off[1][p].x += w1 * pr;
          off[1][p].y += w1 * pi;
        }
      }
    }
  };
  // one barrier per step: it drains the copies of step s (every wave waits for its own before
  // arriving) and separates compute(s-1) from the copies of step s+1 into the same buffer
  MaskRegs<MT> ra, rb;
  fetch_y(0);
  fetch_m(0, ra);
  for (int s = 0; s < steps; s += 2) {
    __syncthreads();
    if (s + 1 < steps) { fetch_y(s + 1); fetch_m(s + 1, rb); }
    if (kvalid) compute(s, ra);
    if (s + 1 >= steps) break;
    __syncthreads();
    if (s + 2 < steps) { fetch_y(s + 2); fetch_m(s + 2, ra); }
    if (kvalid) compute(s + 1, rb);
  }
  if (!kvalid || fraw >= F) return;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    double* out = part + ((((b * plan.chunks + c) * K + k) * 2 + m) * (int64_t)(D * D)) * F + f;
#pragma unroll
    for (int i = 0; i < D; ++i) out[(int64_t)i * F] = diag[m][i];
#pragma unroll
    for (int p = 0; p < D * (D - 1) / 2; ++p) {
      out[(int64_t)(D + 2 * p) * F] = off[m][p].x;
      out[(int64_t)(D + 2 * p + 1) * F] = off[m][p].y;
    }
  }
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// component-wise select (a ?: on the struct type selects between ADDRESSES and pins both arrays
// in scratch memory)
__device__ __forceinline__ double2 sel(bool c, double2 a, double2 b) {
  return double2{c ? a.x : b.x, c ? a.y : b.y};
}
// 1 / a without forming |a|^2 (no spurious overflow / underflow)
__device__ __forceinline__ double2 crecip(double2 a) {
  if (fabs(a.x) >= fabs(a.y)) {
    const double r = a.y / a.x, den = a.x + a.y * r;
    return double2{1.0 / den, -r / den};
  }
  const double r = a.x / a.y, den = a.x * r + a.y;
  return double2{r / den, -1.0 / den};
}

// chunk partials -> chunk 0, in a fixed order (one lane per (b, k, m, element, bin))
__global__ __launch_bounds__(256) void mvdr_reduce_kernel(double* __restrict__ part, int64_t rows,
                                                          int F, int chunks, int64_t per_b) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over B * per_b
  if (i >= rows * F) return;
  const int64_t b = i / per_b, r = i - b * per_b;
  double* p = part + b * chunks * per_b + r;
  double s = p[0];
  for (int c = 1; c < chunks; ++c) s += p[c * per_b];
  p[0] = s;
}

// Register-resident solve for D <= 6 (2 D^2 complex doubles per lane; every index is static after
// unrolling, row exchanges are selects).  Same arithmetic as mvdr_weights_kernel below.
template <int D>
__global__ __launch_bounds__(64) void mvdr_solve_kernel(
    const double* __restrict__ part, double2* __restrict__ wconj, int* __restrict__ info,
    int64_t B, int K, int F, int chunks, int ref, double eps) {
  const int nf = (F + 63) / 64;
  const int ft = blockIdx.x % nf;
  const int k = (blockIdx.x / nf) % K;
  const int64_t b = blockIdx.x / ((int64_t)nf * K);
  const int f = ft * 64 + threadIdx.x;
  if (f >= F) return;
  constexpr int DD = D * D;
  double2 A[D][D], X[D][D];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const double* q = part + (((b * chunks) * K + k) * 2 + m) * (int64_t)DD * F + f;
    int p = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const double v = q[(int64_t)i * F];
      if (m) A[i][i] = double2{v, 0.0}; else X[i][i] = double2{v, 0.0};
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i + 1; j < D; ++j, ++p) {
        const double re = q[(int64_t)(D + 2 * p) * F], im = q[(int64_t)(D + 2 * p + 1) * F];
        if (m) { A[i][j] = double2{re, im}; A[j][i] = double2{re, -im}; }
        else   { X[i][j] = double2{re, im}; X[j][i] = double2{re, -im}; }
      }
  }
  bool singular = false;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    int piv = p;
    double best = fabs(A[p][p].x) + fabs(A[p][p].y);
#pragma unroll
    for (int i = p + 1; i < D; ++i) {
      const double v = fabs(A[i][p].x) + fabs(A[i][p].y);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) singular = true;
#pragma unroll
    for (int i = p + 1; i < D; ++i) {
      const bool sw = piv == i;
#pragma unroll
      for (int j = p; j < D; ++j) {
        const double2 t = A[p][j], u = A[i][j];
        A[p][j] = sel(sw, u, t);
        A[i][j] = sel(sw, t, u);
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double2 t = X[p][j], u = X[i][j];
        X[p][j] = sel(sw, u, t);
        X[i][j] = sel(sw, t, u);
      }
    }
    const double2 r = crecip(A[p][p]);
#pragma unroll
    for (int i = p + 1; i < D; ++i) {
      const double2 l = cmul(A[i][p], r);
#pragma unroll
      for (int j = p + 1; j < D; ++j) {
        const double2 a = A[p][j];
        A[i][j] = double2{A[i][j].x - (l.x * a.x - l.y * a.y), A[i][j].y - (l.x * a.y + l.y * a.x)};
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double2 a = X[p][j];
        X[i][j] = double2{X[i][j].x - (l.x * a.x - l.y * a.y), X[i][j].y - (l.x * a.y + l.y * a.x)};
      }
    }
  }
#pragma unroll
  for (int i = D - 1; i >= 0; --i) {
    const double2 r = crecip(A[i][i]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double2 s = X[i][j];
#pragma unroll
      for (int q = i + 1; q < D; ++q) {
        const double2 a = A[i][q], x = X[q][j];
        s.x -= a.x * x.x - a.y * x.y;
        s.y -= a.x * x.y + a.y * x.x;
      }
      X[i][j] = cmul(s, r);
    }
  }
  if (singular) atomicAdd(info, 1);
  double lam = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) lam += X[i][i].x;
  if (lam < eps) lam = eps;
  const double scl = 1.0 / lam;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    double2 v = X[d][0];
#pragma unroll
    for (int c = 1; c < D; ++c) v = sel(ref == c, X[d][c], v);
    wconj[((b * K + k) * D + d) * (int64_t)F + f] = double2{v.x * scl, -(v.y * scl)};
  }
}

// General (D <= 8) solve with the systems in LDS; chunk partials already reduced (chunk 0).
__global__ __launch_bounds__(64) void mvdr_weights_kernel(
    const double* __restrict__ part, double2* __restrict__ wconj, int* __restrict__ info,
    int64_t B, int K, int D, int F, int chunks, int ref, double eps) {
  extern __shared__ double2 sm[];
  const int lane = threadIdx.x;
  const int nf = (F + 63) / 64;
  const int ft = blockIdx.x % nf;
  const int k = (blockIdx.x / nf) % K;
  const int64_t b = blockIdx.x / ((int64_t)nf * K);
  const int f = ft * 64 + lane;
  if (f >= F) return;
  const int DD = D * D;
  double2* A = sm + lane;                    // interference psd, element e at A[e * 64]
  double2* Bm = sm + DD * 64 + lane;         // target psd -> phi
#define A_(i, j) A[((i) * D + (j)) * 64]
#define B_(i, j) Bm[((i) * D + (j)) * 64]
  for (int m = 0; m < 2; ++m) {
    int p = 0;
    for (int i = 0; i < D; ++i) {
      const double s = part[((((b * chunks) * K + k) * 2 + m) * (int64_t)DD + i) * F + f];
      if (m) A_(i, i) = double2{s, 0.0}; else B_(i, i) = double2{s, 0.0};
    }
    for (int i = 0; i < D; ++i)
      for (int j = i + 1; j < D; ++j, ++p) {
        const double* q = part + ((((b * chunks) * K + k) * 2 + m) * (int64_t)DD + D + 2 * p) * F + f;
        const double re = q[0], im = q[F];
        if (m) { A_(i, j) = double2{re, im}; A_(j, i) = double2{re, -im}; }
        else   { B_(i, j) = double2{re, im}; B_(j, i) = double2{re, -im}; }
      }
  }
  // ---- LU with partial pivoting on [A | B]  (zgetrf's pivot: first max of |re| + |im|)
  bool singular = false;
  for (int p = 0; p < D; ++p) {
    int piv = p;
    double best = fabs(A_(p, p).x) + fabs(A_(p, p).y);
    for (int i = p + 1; i < D; ++i) {
      const double v = fabs(A_(i, p).x) + fabs(A_(i, p).y);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) singular = true;
    if (piv != p)
      for (int j = 0; j < D; ++j) {
        const double2 ta = A_(p, j); A_(p, j) = A_(piv, j); A_(piv, j) = ta;
        const double2 tb = B_(p, j); B_(p, j) = B_(piv, j); B_(piv, j) = tb;
      }
    const double2 r = crecip(A_(p, p));
    for (int i = p + 1; i < D; ++i) {
      const double2 l = cmul(A_(i, p), r);
      for (int j = p + 1; j < D; ++j) {
        const double2 a = A_(p, j), x = A_(i, j);
        A_(i, j) = double2{x.x - (l.x * a.x - l.y * a.y), x.y - (l.x * a.y + l.y * a.x)};
      }
      for (int j = 0; j < D; ++j) {
        const double2 a = B_(p, j), x = B_(i, j);
        B_(i, j) = double2{x.x - (l.x * a.x - l.y * a.y), x.y - (l.x * a.y + l.y * a.x)};
      }
    }
  }
  // ---- back substitution: phi overwrites B
  for (int i = D - 1; i >= 0; --i) {
    const double2 r = crecip(A_(i, i));
    for (int j = 0; j < D; ++j) {
      double2 s = B_(i, j);
      for (int q = i + 1; q < D; ++q) {
        const double2 a = A_(i, q), x = B_(q, j);
        s.x -= a.x * x.x - a.y * x.y;
        s.y -= a.x * x.y + a.y * x.x;
      }
      B_(i, j) = cmul(s, r);
    }
  }
  if (singular) atomicAdd(info, 1);
  double lam = 0.0;
  for (int i = 0; i < D; ++i) lam += B_(i, i).x;
  if (lam < eps) lam = eps;                       // clamp(min=eps); NaN stays NaN
  const double scl = 1.0 / lam;                   // complex / (real + 0i) the way torch divides
  for (int d = 0; d < D; ++d) {
    const double2 v = B_(d, ref);
    wconj[((b * K + k) * D + d) * (int64_t)F + f] = double2{v.x * scl, -(v.y * scl)};
  }
#undef A_
#undef B_
}

template <int D>
__global__ __launch_bounds__(64) void mvdr_apply_kernel(
    const double2* __restrict__ obs, const double2* __restrict__ wconj,
    const void* __restrict__ masks, int mask_f64, double2* __restrict__ enh, int64_t B, int K,
    int M, int64_t T, int F, int nf, int achunks, int64_t tchunk, int masking,
    double masking_eps) {
  const int groups = (K + KG - 1) / KG;
  int64_t tile;
  int grp;
  if (!xcd_tile(blockIdx.x, groups, B * achunks * nf, tile, grp)) return;
  const int ft = (int)(tile % nf);
  const int c = (int)((tile / nf) % achunks);
  const int64_t b = tile / ((int64_t)nf * achunks);
  const int f = ft * 64 + threadIdx.x;
  if (f >= F) return;
  const int k0 = grp * KG;
  const int nk = K - k0 < KG ? K - k0 : KG;
  double2 w[KG][D];
#pragma unroll
  for (int kk = 0; kk < KG; ++kk)
#pragma unroll
    for (int d = 0; d < D; ++d)
      w[kk][d] = kk < nk ? wconj[((b * K + k0 + kk) * D + d) * (int64_t)F + f] : double2{0.0, 0.0};
  const int64_t t0 = c * tchunk;
  const int64_t t1 = t0 + tchunk < T ? t0 + tchunk : T;
  const double2* y0 = obs + b * D * T * F + f;
#pragma unroll 2
  for (int64_t t = t0; t < t1; ++t) {
    double2 y[D];
#pragma unroll
    for (int d = 0; d < D; ++d) y[d] = y0[(d * T + t) * F];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
      if (kk >= nk) break;
      double2 e = {0.0, 0.0};
#pragma unroll
      for (int d = 0; d < D; ++d) {
        e.x += w[kk][d].x * y[d].x - w[kk][d].y * y[d].y;
        e.y += w[kk][d].x * y[d].y + w[kk][d].y * y[d].x;
      }
      if (masking) {
        double mk = load_mask(masks, mask_f64, (((b * K + k0 + kk) * M) * T + t) * F + f);
        if (mk < masking_eps) mk = masking_eps;
        e.x *= mk;
        e.y *= mk;
      }
      enh[((b * K + k0 + kk) * T + t) * F + f] = e;
    }
  }
}

template <int D>
int launch_psd(const double* obs, const void* masks, int mask_f64, double* part, int64_t B, int K,
               int M, int64_t T, int F, const Plan& p, hipStream_t s) {
  const int64_t tiles = B * p.chunks * p.nf;
  const int64_t grid = ((tiles + 7) / 8) * 8 * ((K + PW - 1) / PW);
  if (mask_f64)
    hipLaunchKernelGGL((mvdr_psd_kernel<D, double>), dim3((unsigned)grid), dim3(64 * PW), 0, s,
                       reinterpret_cast<const double2*>(obs), static_cast<const double*>(masks), part,
                       B, K, M, T, F, p);
  else
    hipLaunchKernelGGL((mvdr_psd_kernel<D, float>), dim3((unsigned)grid), dim3(64 * PW), 0, s,
                       reinterpret_cast<const double2*>(obs), static_cast<const float*>(masks), part,
                       B, K, M, T, F, p);
  return tssep_launch_status();
}

template <int D>
int launch_apply(const double* obs, const double* wconj, const void* masks, int mask_f64,
                 double* enh, int64_t B, int K, int M, int64_t T, int F, int masking,
                 double masking_eps, hipStream_t s) {
  const int nf = (F + 63) / 64;
  const int groups = (K + KG - 1) / KG;
  int64_t c = (4096 + B * nf * groups - 1) / (B * nf * groups);
  const int64_t cmax = (T + 15) / 16;
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  const int64_t tchunk = (T + c - 1) / c;
  const int achunks = (int)((T + tchunk - 1) / tchunk);
  const int64_t tiles = B * achunks * nf;
  const int64_t grid = ((tiles + 7) / 8) * 8 * groups;
  hipLaunchKernelGGL(mvdr_apply_kernel<D>, dim3((unsigned)grid), dim3(64), 0, s,
                     reinterpret_cast<const double2*>(obs), reinterpret_cast<const double2*>(wconj),
                     masks, mask_f64, reinterpret_cast<double2*>(enh), B, K, M, T, F, nf, achunks,
                     tchunk, masking, masking_eps);
  return tssep_launch_status();
}

#define DISPATCH_D(D, CALL)                    \
  switch (D) {                                 \
    case 1: return CALL(1);                    \
    case 2: return CALL(2);                    \
    case 3: return CALL(3);                    \
    case 4: return CALL(4);                    \
    case 5: return CALL(5);                    \
    case 6: return CALL(6);                    \
    case 7: return CALL(7);                    \
    case 8: return CALL(8);                    \
    default: return TSSEP_E_UNSUPPORTED;       \
  }

bool shape_ok(int64_t B, int K, int M, int D, int64_t T, int F) {
  return B > 0 && K > 0 && T > 0 && F > 0 && D > 0 && (M == 1 || M == 2) &&
         B * K * 2 * ((F + 63) / 64) * ((T + 15) / 16 + 8) < (int64_t)1 << 28;
}

}  // namespace

extern "C" int64_t tssep_mvdr_partial_bytes(int64_t B, int K, int D, int64_t T, int F) {
  if (B <= 0 || K <= 0 || D <= 0 || D > MAXD || T <= 0 || F <= 0) return 0;
  const Plan p = make_plan(B, K, T, F);
  return B * p.chunks * K * 2 * (int64_t)(D * D) * F * (int64_t)sizeof(double);
}

extern "C" int tssep_mvdr_psd(const double* obs, const void* masks, int mask_f64, double* partials,
                              int64_t B, int K, int M, int D, int64_t T, int F, void* stream) {
  if (!obs || !masks || !partials) return TSSEP_E_NULL;
  if (!shape_ok(B, K, M, D, T, F)) return TSSEP_E_SHAPE;
  if (D > MAXD) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(obs)) return TSSEP_E_ALIGN;
  const Plan p = make_plan(B, K, T, F);
#define CALL(DD) launch_psd<DD>(obs, masks, mask_f64, partials, B, K, M, T, F, p, (hipStream_t)stream)
  DISPATCH_D(D, CALL)
#undef CALL
}

template <int D>
int launch_solve(const double* part, double* wconj, int* info, int64_t B, int K, int F, int nf,
                 int chunks, int ref, double eps, hipStream_t s) {
  hipLaunchKernelGGL(mvdr_solve_kernel<D>, dim3((unsigned)(B * K * nf)), dim3(64), 0, s, part,
                     reinterpret_cast<double2*>(wconj), info, B, K, F, chunks, ref, eps);
  return tssep_launch_status();
}

extern "C" int tssep_mvdr_weights(double* partials, double* wconj, int* info, int64_t B, int K,
                                  int D, int64_t T, int F, int reference_channel, double eps,
                                  void* stream) {
  if (!partials || !wconj || !info) return TSSEP_E_NULL;
  if (!shape_ok(B, K, 1, D, T, F) || reference_channel < 0 || reference_channel >= D)
    return TSSEP_E_SHAPE;
  if (D > MAXD) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(wconj)) return TSSEP_E_ALIGN;
  const Plan p = make_plan(B, K, T, F);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return TSSEP_E_LAUNCH;
  if (p.chunks > 1) {
    const int64_t rows = B * K * 2 * D * D;
    hipLaunchKernelGGL(mvdr_reduce_kernel, dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, s,
                       partials, rows, F, p.chunks, (int64_t)K * 2 * D * D * F);
  }
  switch (D) {
    case 1: return launch_solve<1>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    case 2: return launch_solve<2>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    case 3: return launch_solve<3>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    case 4: return launch_solve<4>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    case 5: return launch_solve<5>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    case 6: return launch_solve<6>(partials, wconj, info, B, K, F, p.nf, p.chunks, reference_channel, eps, s);
    default: break;
  }
  const size_t lds = (size_t)2 * D * D * 64 * sizeof(double2);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mvdr_weights_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * MAXD * MAXD * 64 * (int)sizeof(double2)) != hipSuccess)
      return TSSEP_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(mvdr_weights_kernel, dim3((unsigned)(B * K * p.nf)), dim3(64), lds, s, partials,
                     reinterpret_cast<double2*>(wconj), info, B, K, D, F, p.chunks,
                     reference_channel, eps);
  return tssep_launch_status();
}

extern "C" int tssep_mvdr_apply(const double* obs, const double* wconj, const void* masks,
                                int mask_f64, double* enh, int64_t B, int K, int M, int D, int64_t T,
                                int F, int masking, double masking_eps, void* stream) {
  if (!obs || !wconj || !enh || (masking && !masks)) return TSSEP_E_NULL;
  if (!shape_ok(B, K, M, D, T, F)) return TSSEP_E_SHAPE;
  if (D > MAXD) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(obs) || !aligned16(wconj) || !aligned16(enh)) return TSSEP_E_ALIGN;
#define CALL(DD) \
  launch_apply<DD>(obs, wconj, masks, mask_f64, enh, B, K, M, T, F, masking, masking_eps, (hipStream_t)stream)
  DISPATCH_D(D, CALL)
#undef CALL
}

extern "C" int64_t tssep_mvdr_workspace_bytes(int64_t B, int K, int D, int64_t T, int F) {
  const int64_t pb = tssep_mvdr_partial_bytes(B, K, D, T, F);
  if (pb == 0) return 0;
  return pb + B * K * D * (int64_t)F * 16 + 16;
}

extern "C" int tssep_mvdr_souden_fwd(const double* obs, const void* masks, int mask_f64, double* enh,
                                     void* workspace, int* info, int64_t B, int K, int M, int D,
                                     int64_t T, int F, int reference_channel, double eps, int masking,
                                     double masking_eps, void* stream) {
  if (!workspace) return TSSEP_E_NULL;
  if (!aligned16(workspace)) return TSSEP_E_ALIGN;
  const int64_t pb = tssep_mvdr_partial_bytes(B, K, D, T, F);
  if (pb == 0) return D > MAXD ? TSSEP_E_UNSUPPORTED : TSSEP_E_SHAPE;
  double* part = static_cast<double*>(workspace);
  double* wconj = reinterpret_cast<double*>(static_cast<char*>(workspace) + ((pb + 15) / 16) * 16);
  int st = tssep_mvdr_psd(obs, masks, mask_f64, part, B, K, M, D, T, F, stream);
  if (st != TSSEP_OK) return st;
  st = tssep_mvdr_weights(part, wconj, info, B, K, D, T, F, reference_channel, eps, stream);
  if (st != TSSEP_OK) return st;
  return tssep_mvdr_apply(obs, wconj, masks, mask_f64, enh, B, K, M, D, T, F, masking, masking_eps,
                          stream);
}
