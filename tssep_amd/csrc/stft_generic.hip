// STFT / inverse STFT / adjoint of the inverse STFT for FFT plans OTHER than 1024 / 256 (round 5).
//
// The reference's `fe` slot takes any `size` / `shift` (tssep/exp/init_cfg_common.yaml:33-43; TorchMFCC itself defaults
// to 400 / 200, tssep/train/feature_extractor_torchaudio.py:24-25); the shipped configurations use 1024 / 256, which
// stft.hip serves with a plan specialised to that size (radix-8 x 3 in registers, fused mask head, ring of frames).
// This file is the general plan behind the SAME entry points (tssep_stft_fwd, tssep_istft_fwd, tssep_istft_bwd):
//   size even, size / 2 = 2^a 3^b 5^c <= 2048, 1 <= shift <= min(size, 512)
//   -- e.g. 512 / 128, 400 / 200, 256 / 64, 2048 / 512, 960 / 240.
// A real transform of `size` samples is a complex transform of NH = size / 2 points of z[n] = x[2n] + i x[2n+1] plus a
// butterfly pass (as in stft.hip).  The NH-point transform is a Stockham autosort FFT with run-time radices {4, 2, 3, 5}
// (decimation in frequency: r-point DFT, then the twiddle exp(-2 pi i p j / n) on output j), ONE WAVE per frame, two
// LDS lines per wave in ping-pong; the lines are private to the wave, so the hand-offs between lanes need only a
// wave-level fence.  Memory-bound in the roofline sense like the specialised plan, but neither fused nor tuned:
// correctness first (parity against oracle/stft.py in tests/test_gpu_kernels.py::test_stft_generic_plans).
//
// Inverse: one wave per output hop (`shift` samples): it inverse-transforms the <= ceil(size / shift) frames that cover
// the hop, in ascending frame order, and sums their windowed segments in registers -- a FIXED summation order
// (deterministic overlap-add) and no workspace (the C ABI hands over no scratch buffer); every frame is transformed by
// each hop it covers.
#include <math.h>
#include "common.h"

namespace {

constexpr int GEN_MAX_NH = 2048;
constexpr int GEN_MAX_FACTORS = 12;
struct GenPlan {
  int nh;                       // complex length
  int nf;                       // number of stages
  int radix[GEN_MAX_FACTORS];
};

#define GWAVE_SYNC()                                         \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

__device__ __forceinline__ float2 gmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 gadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 gsub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 gmi(float2 a) { return make_float2(a.y, -a.x); }       // a * (-i)

// r-point DFT (forward kernel exp(-2 pi i jk / r)), in place, natural order
__device__ __forceinline__ void dft2(float2* v) {
  const float2 a = v[0], b = v[1];
  v[0] = gadd(a, b); v[1] = gsub(a, b);
}
__device__ __forceinline__ void dft3(float2* v) {
  const float c = -0.5f, s = -0.86602540378443864676f;       // exp(-2 pi i / 3) = c + i s
  const float2 t = gadd(v[1], v[2]), d = gsub(v[1], v[2]);
  const float2 m = make_float2(v[0].x + c * t.x, v[0].y + c * t.y);
  const float2 e = make_float2(-s * d.y, s * d.x);           // i s d
  v[0] = gadd(v[0], t);
  v[1] = gadd(m, e);
  v[2] = gsub(m, e);
}
__device__ __forceinline__ void dft4(float2* v) {
  const float2 a0 = gadd(v[0], v[2]), a1 = gsub(v[0], v[2]), a2 = gadd(v[1], v[3]), a3 = gmi(gsub(v[1], v[3]));
  v[0] = gadd(a0, a2); v[2] = gsub(a0, a2);
  v[1] = gadd(a1, a3); v[3] = gsub(a1, a3);
}
__device__ __forceinline__ void dft5(float2* v) {
  // exp(-2 pi i k / 5): c1 = cos(2 pi / 5), c2 = cos(4 pi / 5), s1 = sin(2 pi / 5), s2 = sin(4 pi / 5)
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 t1 = gadd(v[1], v[4]), t2 = gadd(v[2], v[3]), d1 = gsub(v[1], v[4]), d2 = gsub(v[2], v[3]);
  const float2 m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
  const float2 m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
  // -i (s1 d1 + s2 d2) and -i (s2 d1 - s1 d2)
  const float2 q1 = make_float2(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y);
  const float2 q2 = make_float2(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y);
  const float2 e1 = gmi(q1), e2 = gmi(q2);
  v[0] = gadd(v[0], gadd(t1, t2));
  v[1] = gadd(m1, e1);
  v[4] = gsub(m1, e1);
  v[2] = gadd(m2, e2);
  v[3] = gsub(m2, e2);
}

// one Stockham stage of radix R (compile-time: the butterfly lives in registers)
template <int R>
__device__ __forceinline__ void stockham_stage(const float2* in, float2* out, const float2* twl, int NH, int n, int s, int lane) {
  const int m = n / R;
  const int tstep = NH / n;                        // twiddle exp(-2 pi i p j / n) = twl[(p j tstep) mod NH]
  for (int bf = lane; bf < NH / R; bf += 64) {
    const int p = bf / s, q = bf - p * s;
    float2 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = in[q + s * (p + k * m)];
    if (R == 4) dft4(v); else if (R == 2) dft2(v); else if (R == 3) dft3(v); else dft5(v);
    out[q + s * (R * p)] = v[0];
    int ti = 0;
#pragma unroll
    for (int j = 1; j < R; ++j) {
      ti += p * tstep;
      if (ti >= NH) ti -= NH;                      // (p tstep < NH / R, so one subtraction suffices)
      out[q + s * (R * p + j)] = gmul(v[j], twl[ti]);
    }
  }
}

// NH-point forward FFT of the wave's line `a` (natural order in, natural order out); `b` is the second line.
// twl[k] = exp(-2 pi i k / NH).  Returns the line that holds the result.
__device__ __forceinline__ float2* stockham_wave(const GenPlan& pl, float2* a, float2* b, const float2* twl, int lane) {
  const int NH = pl.nh;
  int n = NH, s = 1;
  float2* in = a;
  float2* out = b;
  for (int st = 0; st < pl.nf; ++st) {
    const int r = pl.radix[st];
    if (r == 4) stockham_stage<4>(in, out, twl, NH, n, s, lane);
    else if (r == 2) stockham_stage<2>(in, out, twl, NH, n, s, lane);
    else if (r == 3) stockham_stage<3>(in, out, twl, NH, n, s, lane);
    else stockham_stage<5>(in, out, twl, NH, n, s, lane);
    GWAVE_SYNC();
    float2* t = in; in = out; out = t;
    n /= r; s *= r;
  }
  return in;
}

// frames -> rfft.  STFT (window = analysis window, s_in = s_edge = 1) and adjoint of the inverse STFT (window = synthesis
// window, interior bins x 2 / size, DC / Nyquist x 1 / size), as in stft.hip.  tw: [NH] exp(-2 pi i k / NH), then
// [NH + 1] exp(-2 pi i k / size).  Dynamic LDS: twl[NH] | per wave 2 lines of NH float2.
__global__ __launch_bounds__(256) void rfft_generic_kernel(const float* __restrict__ x, int64_t rows, int64_t N, int64_t T,
                                                           int shift, int pad_left, const float* __restrict__ window,
                                                           const float2* __restrict__ tw, float2* __restrict__ X,
                                                           float s_in, float s_edge, GenPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float2 gsm[];
  const int NH = pl.nh;
  float2* twl = gsm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2* la = gsm + NH + (2 * wave) * NH;
  float2* lb = la + NH;
  for (int i = tid; i < NH; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int64_t total = rows * T;
  for (int64_t fidx = (int64_t)blockIdx.x * 4 + wave; fidx < total; fidx += (int64_t)gridDim.x * 4) {
    const int64_t row = fidx / T, t = fidx - row * T;
    const int64_t base = t * shift - pad_left;
    const float* xr = x + row * N;
    for (int n = lane; n < NH; n += 64) {
      const int64_t i0 = base + 2 * n, i1 = i0 + 1;
      const float a = (i0 >= 0 && i0 < N) ? xr[i0] * window[2 * n] : 0.f;
      const float b = (i1 >= 0 && i1 < N) ? xr[i1] * window[2 * n + 1] : 0.f;
      la[n] = make_float2(a, b);
    }
    GWAVE_SYNC();
    const float2* Z = stockham_wave(pl, la, lb, twl, lane);
    float2* Xo = X + fidx * (NH + 1);
    const float hs_in = 0.5f * s_in;
    for (int k = lane; k < NH; k += 64) {
      const float2 zk = Z[k];
      float2 zm = Z[k == 0 ? 0 : NH - k];
      zm.y = -zm.y;
      const float2 u = gmul(tw[NH + k], gsub(zk, zm));
      float2 o = make_float2(zk.x + zm.x + u.y, zk.y + zm.y - u.x);
      if (k == 0) {
        o.x *= 0.5f * s_edge; o.y = 0.f;
      } else {
        o.x *= hs_in; o.y *= hs_in;
      }
      Xo[k] = o;
    }
    if (lane == 0) {
      const float2 z0 = Z[0];
      Xo[NH] = make_float2((z0.x - z0.y) * s_edge, 0.f);
    }
    GWAVE_SYNC();          // the lines are rewritten by this wave's next frame
  }
}

// inverse STFT: one wave per output hop h of one row: samples [h shift, (h + 1) shift) of the padded signal, i.e. output
// samples h shift - pad_left + j.  Covering frames t = h - c + 1 .. h (c = ceil(size / shift)), ascending: each is
// inverse-transformed (irfft semantics: the imaginary parts of DC and Nyquist are ignored, scale 1 / size), multiplied
// by the synthesis window and its segment [(h - t) shift, + shift) added.  Lanes hold ceil(shift / 64) accumulators.
constexpr int GEN_MAX_ACC = 8;             // shift <= 512 (accumulators are registers: static indices only)
__global__ __launch_bounds__(256) void istft_generic_kernel(const float2* __restrict__ X, int64_t rows, int64_t T, int size,
                                                            int shift, int pad_left, int64_t N, const float* __restrict__ wsyn,
                                                            const float2* __restrict__ tw, float* __restrict__ y, GenPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float2 gsm[];
  const int NH = pl.nh;
  float2* twl = gsm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2* la = gsm + NH + (2 * wave) * NH;
  float2* lb = la + NH;
  for (int i = tid; i < NH; i += 256) twl[i] = tw[i];
  __syncthreads();
  const int c = (size + shift - 1) / shift;
  const int64_t hops = (pad_left + N + shift - 1) / shift;       // hops that contain output samples
  const int64_t h0 = pad_left / shift;                           // first hop with an output sample
  const int64_t per_row = hops - h0;
  const int64_t total = rows * per_row;
  const float inv_nh = 1.0f / (float)NH;
  const int nacc = (shift + 63) / 64;
  for (int64_t idx = (int64_t)blockIdx.x * 4 + wave; idx < total; idx += (int64_t)gridDim.x * 4) {
    const int64_t row = idx / per_row, h = h0 + (idx - row * per_row);
    float acc[GEN_MAX_ACC];
#pragma unroll
    for (int a = 0; a < GEN_MAX_ACC; ++a) acc[a] = 0.f;
    for (int64_t t = h - c + 1; t <= h; ++t) {
      if (t < 0 || t >= T) continue;                             // (wave-uniform)
      const int off = (int)(h - t) * shift;                      // segment of frame t that falls into hop h
      if (off >= size) continue;
      const float2* Xr = X + (row * T + t) * (NH + 1);
      // Z_k = E_k + i O_k, E = (X_k + conj X_{NH-k}) / 2, O = conj(w_k) (X_k - conj X_{NH-k}) / 2; the inverse transform as
      // conj(FFT(conj Z)) / NH: the line receives conj(Z)
      for (int k = lane; k < NH; k += 64) {
        float2 xk = Xr[k], xm = Xr[NH - k];
        if (k == 0) { xk.y = 0.f; xm.y = 0.f; }                  // irfft ignores the imaginary parts of DC / Nyquist
        xm.y = -xm.y;
        const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y + xm.y));
        const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y - xm.y));
        const float2 w = tw[NH + k];
        const float2 o = gmul(make_float2(w.x, -w.y), d);        // conj(w_k) d
        // Z = e + i o = (e.x - o.y, e.y + o.x); conj(Z) = (e.x - o.y, -(e.y + o.x))
        la[k] = make_float2(e.x - o.y, -(e.y + o.x));
      }
      GWAVE_SYNC();
      const float2* R = stockham_wave(pl, la, lb, twl, lane);    // R = FFT(conj Z); z[n] = conj(R[n]) / NH = x[2n] + i x[2n+1]
#pragma unroll
      for (int a = 0; a < GEN_MAX_ACC; ++a) {
        const int j = lane + 64 * a;
        if (a < nacc && j < shift && off + j < size) {
          const int sidx = off + j;
          const float2 r = R[sidx >> 1];
          const float v = ((sidx & 1) ? -r.y : r.x) * inv_nh;
          acc[a] += v * wsyn[sidx];
        }
      }
      GWAVE_SYNC();
    }
#pragma unroll
    for (int a = 0; a < GEN_MAX_ACC; ++a) {
      const int j = lane + 64 * a;
      const int64_t n_out = h * shift - pad_left + j;
      if (a < nacc && j < shift && n_out >= 0 && n_out < N) y[row * N + n_out] = acc[a];
    }
  }
}

static bool make_plan(int size, GenPlan* pl) {
  if (size < 4 || (size & 1)) return false;
  int nh = size / 2;
  if (nh > GEN_MAX_NH) return false;
  pl->nh = nh;
  pl->nf = 0;
  int rest = nh;
  const int radices[4] = {4, 2, 3, 5};
  for (int ri = 0; ri < 4; ++ri)
    while (rest % radices[ri] == 0 && rest > 1) {
      if (pl->nf >= GEN_MAX_FACTORS) return false;
      pl->radix[pl->nf++] = radices[ri];
      rest /= radices[ri];
    }
  return rest == 1;
}

}  // namespace

// ---- host side (called from stft.hip's entry points for plans other than 1024 / 256) -------------------------------
static size_t generic_lds_bytes(const GenPlan& pl);
// The kernels keep two LDS lines of size / 2 complex values per wave (72 bytes per value and workgroup of four waves: 144 KB
// at size 4096): a plan is advertised only where that fits the device's LDS (160 KB on gfx950; without a device
// -- build / CPU tests -- the gfx950 figure).
static size_t device_lds_limit() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
    return (size_t)v;
  (void)hipGetLastError();
  return (size_t)160 * 1024;
}
int tssep_generic_plan_supported(int size, int shift) {
  GenPlan pl;
  return make_plan(size, &pl) && shift >= 1 && shift <= size && shift <= 64 * GEN_MAX_ACC &&
         generic_lds_bytes(pl) <= device_lds_limit();
}

int tssep_generic_twiddles(int size, float* host_out) {
  GenPlan pl;
  if (!make_plan(size, &pl)) return TSSEP_E_UNSUPPORTED;
  const int nh = pl.nh;
  for (int k = 0; k < nh; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)nh;
    host_out[2 * k] = (float)cos(a);
    host_out[2 * k + 1] = (float)sin(a);
  }
  for (int k = 0; k <= nh; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)size;
    host_out[2 * (nh + k)] = (float)cos(a);
    host_out[2 * (nh + k) + 1] = (float)sin(a);
  }
  return TSSEP_OK;
}

static size_t generic_lds_bytes(const GenPlan& pl) { return (size_t)(pl.nh + 8 * pl.nh) * sizeof(float2); }

int tssep_generic_rfft(const float* x, int64_t rows, int64_t N, int size, int shift, int pad_left, const float* window,
                       const float* tw, float* X, int64_t T, float s_in, float s_edge, void* stream) {
  GenPlan pl;
  if (!make_plan(size, &pl) || shift < 1 || shift > size) return TSSEP_E_UNSUPPORTED;
  const int64_t total = rows * T;
  int64_t blocks = (total + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  const size_t lds = generic_lds_bytes(pl);
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)rfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TSSEP_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(rfft_generic_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, rows, N, T, shift,
                     pad_left, window, (const float2*)tw, (float2*)X, s_in, s_edge, pl);
  return tssep_launch_status();
}

int tssep_generic_istft(const float* X, int64_t rows, int64_t T, int size, int shift, int pad_left, const float* wsyn,
                        const float* tw, float* y, int64_t N, void* stream) {
  GenPlan pl;
  if (!make_plan(size, &pl) || shift < 1 || shift > size || shift > 64 * GEN_MAX_ACC) return TSSEP_E_UNSUPPORTED;
  const int64_t hops = (pad_left + N + shift - 1) / shift - pad_left / shift;
  const int64_t total = rows * hops;
  int64_t blocks = (total + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  const size_t lds = generic_lds_bytes(pl);
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)istft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TSSEP_E_UNSUPPORTED;
  }
  hipLaunchKernelGGL(istft_generic_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, (const float2*)X, rows, T,
                     size, shift, pad_left, N, wsyn, (const float2*)tw, y, pl);
  return tssep_launch_status();
}
