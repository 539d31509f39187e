// STFT / inverse STFT / adjoint of the inverse STFT with paderbox semantics, size 1024.
//
// Replaces fe.stft (tssep/train/model.py:503-504) and fe.istft (tssep/train/model.py:661-664;
// third-party padertorch/paderbox code, restated in oracle/stft.py) and the autograd backward
// of the latter.
//
// FFT plan: a 1024-point real FFT is a 512-point complex FFT of z[n] = x[2n] + i x[2n+1]
// plus a butterfly pass.  The 512-point transform is a Stockham autosort radix-8 x 3
// (Govindaraju et al. formulation): ONE WAVE per frame, 8 complex points per lane in
// registers, three in-register 8-point DFTs, the two transposes in between through a
// per-wave LDS line.  The line index is padded (i + i/8) so the stride-8 scatter of the
// first stage is bank-conflict free for ds_write_b64.
//
// All three kernels are HBM/L2 streaming kernels in the roofline sense
// (5 128 B per frame for the STFT, K x 5 128 B per frame for the inverse).
#include <math.h>
#include "common.h"

namespace {

// Buffer resources for the frame streams: one 4-SGPR descriptor per row (base + byte range) and ONE 32-bit lane
// offset; the r-th access of a lane is an immediate offset.  Out-of-range elements (the fading zeros in front of a
// row, the padding behind it) read as 0 by the range check -- no per-load masks, no per-load 64-bit addresses.
typedef __amdgpu_buffer_rsrc_t srd_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ srd_t make_srd(const void* p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffff ? 0x7fffffff : bytes), 0x00020000);
}
__device__ __forceinline__ float2 bload2(srd_t r, unsigned voff) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}

__device__ __forceinline__ float bload1(srd_t r, unsigned voff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void bstore1(srd_t r, unsigned voff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, 0, 0);
}

constexpr int NH = 512;          // complex FFT length
constexpr int LINE = NH + NH / 8;  // padded LDS line (float2)
#define PADI(i) ((i) + ((i) >> 3))
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

// in-register 8-point DFT (forward, exp(-2 pi i nk/8)), natural order in and out
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
  const float h = 0.70710678118654752440f;
  float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  a5 = make_float2(h * (a5.x + a5.y), h * (a5.y - a5.x));   // * exp(-i pi/4)
  a6 = mul_mi(a6);                                          // * (-i)
  a7 = make_float2(h * (a7.y - a7.x), -h * (a7.x + a7.y));  // * exp(-3 i pi/4)
  float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
  float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = mul_mi(csub(a5, a7));
  v[0] = cadd(b0, b1); v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
  v[1] = cadd(b4, b5); v[5] = csub(b4, b5);
  v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}

// tw2[lane + 64 r] = exp(-2 pi i (lane + 64 r) / 1024) = tw2[lane] * exp(-2 pi i r / 16): one per-lane value
// loaded once per kernel and eight compile-time constants, instead of eight table loads from global memory
// in every frame's dependent chain
__device__ __forceinline__ float2 tw16(int r) {
  constexpr float c[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                          0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
  constexpr float sn[8] = {0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                           -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
  return make_float2(c[r], sn[r]);
}

// w[r] = tw2_lane * exp(-2 pi i r / 16), r < 8: three complex products (r = 1, 2, 3); r + 4 is a further factor -i
__device__ __forceinline__ void lane_twiddles(float2 tw2_lane, float2 (&w)[8]) {
  w[0] = tw2_lane;
#pragma unroll
  for (int r = 1; r < 4; ++r) w[r] = cmul(tw2_lane, tw16(r));
#pragma unroll
  for (int r = 0; r < 4; ++r) w[r + 4] = mul_mi(w[r]);
}

// 512-point forward FFT by one wave.  In: v[r] = z[lane + 64 r].  Out: buf[PADI(k)] = Z[k].
// twl[k] = exp(-2 pi i k / 512) (LDS copy).  The line is private to the wave: LDS operations of one
// wave execute in order, so the hand-offs between lanes need only a wave-level fence (the compiler
// must not reorder them) -- block-wide barriers here coupled the 4 independent waves of a block
// five times per frame.
__device__ __forceinline__ void fft512_wave(float2 (&v)[8], float2* buf, const float2* twl, int lane) {
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[PADI(8 * lane + r)] = v[r];
  WAVE_SYNC();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[PADI(lane + 64 * r)];
  WAVE_SYNC();
  {
    const int k = lane & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], twl[k * r * 8]);
    fft8(v);
    const int d = (lane >> 3) * 64 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[PADI(d + 8 * r)] = v[r];
  }
  WAVE_SYNC();
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[PADI(lane + 64 * r)];
  WAVE_SYNC();
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], twl[lane * r]);
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[PADI(lane + 64 * r)] = v[r];
  WAVE_SYNC();
}

// frames -> rfft.  Used as the STFT (window = analysis window, scales 1) and as the adjoint of
// the inverse STFT (window = synthesis window, interior bins x 2/1024, DC/Nyquist x 1/1024).
// tw: [512] exp(-2 pi i k/512) followed by [513] exp(-2 pi i k/1024).
//
// MASKED = true fuses the mask head's backward into the adjoint (tssep/train/net.py:983 sigmoid +
// tssep/train/enhancer.py:98-100 Masking, backward): a frame's d(estimate) bin never leaves the wave;
// the epilogue reads the logit (4 B) and the observation bin (8 B, shared by the K speakers of an
// utterance: L2) and writes d(logit) = Re(conj(Obs) dEst) m (1 - m), m = sigmoid(logit) -- the chain
// adjoint -> mask head moves 8 K F + 8 F bytes per frame instead of 24 K F + 8 F.
#ifndef TSSEP_RFFT_OCC
#define TSSEP_RFFT_OCC 3
#endif
template <bool MASKED>
__global__ __launch_bounds__(256, TSSEP_RFFT_OCC) void rfft_frames_kernel(
    const float* __restrict__ x, int64_t rows, int64_t N, int64_t T, int shift, int pad_left,
    const float* __restrict__ window, const float2* __restrict__ tw, float2* __restrict__ X,
    float s_in, float s_edge, int iters, const float* __restrict__ logit,
    const float2* __restrict__ obs, float* __restrict__ dlogit, int64_t Kspk,
    const float* __restrict__ tgt, const float* __restrict__ sums, const float* __restrict__ gout,
    const int32_t* __restrict__ iperm, int bt_major) {
  // MASKED with tgt != NULL: x is the time-domain ESTIMATE and the frame's samples are the loss gradient
  // d LogMAE / d est = gout[b] sign(est - tgt) / (N ln10 sums[b])  (sums == NULL: MAE, gout[b] sign / N;
  // tssep/train/loss.py:214-216, 244-247) formed while they are loaded -- tssep_logmae_bwd and its [B,K,N]
  // gradient leave the step.  bt_major: d(logit) is stored where the final Linear's backward reads it, rows
  // (b,t) x (speaker position iperm[b,k], f) -- the inverse of the store remap of the forward
  // (net.py:637-641, 957-967) -- instead of [B,K,T,F] + tssep_logit_map_bwd.
  __shared__ float2 twl[NH];
  __shared__ float2 wl[NH];               // the window as sample pairs: read per frame from LDS -- as global loads the
  __shared__ float2 line[4][LINE];        // compiler kept eight 64-bit per-lane addresses alive across the frame loop
  const int tid = threadIdx.x, lane = tid & 63;
  // (wave-uniform; through readfirstlane so that the per-frame buffer resources below are built in SGPRs -- from a
  // VGPR the compiler wraps EVERY buffer access in a waterfall loop: ~10 extra instructions and a serialisation each)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < NH; i += 256) { twl[i] = tw[i]; wl[i] = reinterpret_cast<const float2*>(window)[i]; }
  __syncthreads();
  const float2 tw2_lane = tw[NH + lane];
  const int64_t total = rows * T;
  for (int it = 0; it < iters; ++it) {
    const int64_t fidx = ((int64_t)blockIdx.x * iters + it) * 4 + wave;
    const bool valid = fidx < total;
    const int64_t row = valid ? fidx / T : 0;
    const int64_t t = valid ? fidx - row * T : 0;
    const int64_t base = t * shift - pad_left;
    const float* xr = x + row * N;
    // Sample loads, branch-free per lane: the eight loads of a frame go out back to back and are waited for
    // once.  (With per-lane branches around them the compiler waited for each load before issuing the
    // next: eight serialised round trips per frame.)  Wave-uniform fast path: N even, x 8-byte aligned and
    // the frame start even -> every sample pair is one aligned float2 that lies inside the row or outside
    // it as a whole; otherwise two clamped 4-byte loads per pair.
    const bool lossgrad = MASKED && tgt != nullptr;          // kernel-uniform
    const float* tr_ = lossgrad ? tgt + row * N : xr;
    float coef = 0.f;
    if (lossgrad) {
      const int64_t b = row / Kspk;
      coef = sums ? gout[b] / ((float)N * 2.30258509299404568402f * sums[b]) : gout[b] / (float)N;
    }
    auto lg = [&](float e, float t_) { const float d = e - t_; return d > 0.f ? coef : (d < 0.f ? -coef : 0.f); };
    const bool fast = ((N & 1) == 0) && ((base & 1) == 0) && ((((uintptr_t)x) & 7u) == 0) &&
                      (!lossgrad || (((uintptr_t)tgt) & 7u) == 0);
    float2 v[8];
    if (!valid) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = make_float2(0.f, 0.f);
    } else if (fast) {
      // (base + 2 lane + 128 r) may be negative or beyond the row: as an unsigned byte offset it then lies outside
      // the descriptor's range and the pair reads as (0, 0)
      const unsigned vo = (unsigned)((base + 2 * lane) * 4);
      const srd_t sx = make_srd(xr, N * 4);
      float2 xv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) xv[r] = bload2(sx, vo + 512u * r);
      if (lossgrad) {
        const srd_t st_ = make_srd(tr_, N * 4);
        float2 tv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tv[r] = bload2(st_, vo + 512u * r);
#pragma unroll
        for (int r = 0; r < 8; ++r) xv[r] = make_float2(lg(xv[r].x, tv[r].x), lg(xv[r].y, tv[r].y));
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float2 wv = wl[lane + 64 * r];
        v[r] = make_float2(xv[r].x * wv.x, xv[r].y * wv.y);
      }
    } else {
      float xa[8], xb[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t i0 = base + 2 * (lane + 64 * r), i1 = i0 + 1;
        const int64_t c0 = i0 < 0 ? 0 : (i0 >= N ? N - 1 : i0), c1 = i1 < 0 ? 0 : (i1 >= N ? N - 1 : i1);
        xa[r] = xr[c0];
        xb[r] = xr[c1];
        if (lossgrad) { xa[r] = lg(xa[r], tr_[c0]); xb[r] = lg(xb[r], tr_[c1]); }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t i0 = base + 2 * (lane + 64 * r), i1 = i0 + 1;
        const float2 wv = wl[lane + 64 * r];
        v[r] = make_float2((i0 >= 0 && i0 < N) ? xa[r] * wv.x : 0.f, (i1 >= 0 && i1 < N) ? xb[r] * wv.y : 0.f);
      }
    }
    float2* buf = line[wave];
    // MASKED: row = (utterance b, speaker); the observation frame is the utterance's.  The epilogue's logit and
    // observation bins do not depend on the transform: requested BEFORE it (24 registers), their latency hides
    // behind the FFT instead of ending every frame with a round trip to memory
    const float* Lr = MASKED ? logit + fidx * (NH + 1) : nullptr;
    const float2* Or = MASKED ? obs + ((row / Kspk) * T + t) * (NH + 1) : nullptr;
    float lg8[8];
    float2 ob8[8];
    if (MASKED && valid) {
      __builtin_amdgcn_sched_barrier(0);      // (not above the sample loads: the kernel is at its register ceiling there)
      const srd_t sl = make_srd(Lr, (NH + 1) * 4), so = make_srd(Or, (NH + 1) * 8);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        ob8[r] = bload2(so, (unsigned)(lane * 8 + 512 * r));
        lg8[r] = bload1(sl, (unsigned)(lane * 4 + 256 * r));
      }
    }
    fft512_wave(v, buf, twl, lane);
    if (valid) {
      float2* Xo = MASKED ? nullptr : X + fidx * (NH + 1);
      float* Dr = MASKED ? dlogit + fidx * (NH + 1) : nullptr;
      if (MASKED && bt_major) {
        const int64_t b = row / Kspk, j = row - b * Kspk;
        const int64_t kpos = iperm ? (int64_t)iperm[b * Kspk + j] : j;
        Dr = dlogit + ((b * T + t) * Kspk + kpos) * (NH + 1);
      }
      auto emit = [&](int k, float2 o, float lgk, float2 ob) {
        if (MASKED) {
          const float m = sigmoidf_mask(lgk);
          Dr[k] = (ob.x * o.x + ob.y * o.y) * m * (1.0f - m);
        } else {
          Xo[k] = o;
        }
      };
      float2 wr[8];
      lane_twiddles(tw2_lane, wr);
      const float hs_in = 0.5f * s_in;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const float2 zk = buf[PADI(k)];
        const int km = (NH - k) & (NH - 1);
        float2 zm = buf[PADI(km)];
        zm.y = -zm.y;
        const float2 u = cmul(wr[r], csub(zk, zm));
        float2 o = make_float2(zk.x + zm.x + u.y, zk.y + zm.y - u.x);
        if (k == 0) {
          o.x *= 0.5f * s_edge; o.y = 0.f;
        } else {
          o.x *= hs_in; o.y *= hs_in;
        }
        emit(k, o, MASKED ? lg8[r] : 0.f, MASKED ? ob8[r] : make_float2(0.f, 0.f));
      }
      if (lane == 0) {
        const float2 z0 = buf[0];
        emit(NH, make_float2((z0.x - z0.y) * s_edge, 0.f), MASKED ? Lr[NH] : 0.f, MASKED ? Or[NH] : make_float2(0.f, 0.f));
      }
    }
    WAVE_SYNC();          // the line is rewritten by this wave's next frame
  }
}

// inverse STFT: a workgroup walks `hcb` consecutive output hops of one row.  Its 4 waves inverse-transform
// 4 consecutive frames per iteration into a RING of 7 windowed frames in LDS (4 being written + the 3
// older ones the pending hops still need); after each iteration the hops whose 4 contributing frames are
// complete are emitted: every output sample sums its <= 4 frame contributions in a fixed order
// (deterministic overlap-add).  Only the first 3 frames of a chunk are transformed twice (by the
// neighbouring chunk as well): 66 transforms per 63 hops.  Round 1 parked HC + 3 = 12 frames per 9 hops
// (a third of all transforms redundant, 48 KB of LDS, two workgroups per CU); the ring needs 28 KB and
// three workgroups fit.
constexpr int RING = 7;
constexpr int HCB_MAX = 64;              // hops per workgroup (upper bound; the launcher balances the chunks)

//
// MASKED = true fuses the mask head in front (net.py:983 + enhancer.py:98-100): a frame's spectrum is
// formed in the wave's LDS line as sigmoid(logit) * Obs (logit 4 B per bin; the 8-B observation bin is
// shared by the K speakers of an utterance: L2) -- neither the mask nor the masked STFT is written;
// the chain mask head -> inverse STFT moves (4 K F + 8 F) + 4 K 256 bytes per frame instead of
// 16 K F + 8 F + 8 K F + 4 K 256.  (Measured, batch 768: 2.07 ms against 1.44 + 1.98 ms; the kernel is bound
// by the per-wave FFT chain, not by bytes: the hardware exp / rcp sigmoid instead of the accurate one changed
// nothing, three resident workgroups instead of two gained 20 %.)
#ifndef TSSEP_ISTFT_OCC
#define TSSEP_ISTFT_OCC 3            // waves per SIMD the register allocation aims at (A/B: build with -D...=2)
#endif
template <bool MASKED>
__global__ __launch_bounds__(256, TSSEP_ISTFT_OCC) void istft_kernel(
    const float2* __restrict__ X, int64_t T, int shift_, int64_t N,
    const float* __restrict__ wsyn, const float2* __restrict__ tw, float* __restrict__ y,
    const float* __restrict__ tgt, float* __restrict__ abs_partial, int nchunks, int hcb,
    const float* __restrict__ logit, const float2* __restrict__ obs, int64_t Kspk) {
  __shared__ float2 twl[NH];
  __shared__ float2 line[4][LINE];
  __shared__ __attribute__((aligned(16))) float fr[RING][1024];       // (51 KB with the rest: three workgroups per CU --
                                                                      // no room for the window table the rfft kernel keeps)
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63;
  // (wave-uniform; through readfirstlane so that the per-frame buffer resources below are built in SGPRs -- from a
  // VGPR the compiler wraps EVERY buffer access in a waterfall loop: ~10 extra instructions and a serialisation each)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row = blockIdx.y;
  const int c = blockIdx.x;
  for (int i = tid; i < NH; i += 256) twl[i] = tw[i];
  __syncthreads();
  const float2 tw2_lane = tw[NH + lane];
  const int64_t t_lo = (int64_t)c * hcb;   // hop h of this chunk sums frames t_lo + h .. t_lo + h + 3
  const int64_t n0 = t_lo * 256;
  const int64_t total_hops = (N + 255) / 256;
  const int hops = (int)(total_hops - t_lo < hcb ? total_hops - t_lo : hcb);
  const int iters = (hops + 3 + 3) / 4;    // frames 0 .. hops + 2
  const float hinv = 0.5f / 512.0f;
  float* yr = y + row * N;
  const float* tr = tgt ? tgt + row * N : nullptr;
  // the chunk's samples [n0, min(N, n0 + 256 hops)) of this row
  const int64_t chunk_bytes = ((N - n0 < (int64_t)hops * 256 ? N - n0 : (int64_t)hops * 256)) * 4;
  const srd_t sy = make_srd(yr + n0, chunk_bytes), str_ = make_srd(tr ? tr + n0 : yr + n0, chunk_bytes);
  float asum = 0.f;
  float2* buf = line[wave];
  // MASKED: the logits and observation bins of a frame are requested one iteration ahead (27 registers), right after
  // the previous frame's have been consumed: their latency hides behind that frame's transform and the hop emission
  // instead of opening every iteration (out-of-range frames get an empty resource: the loads return 0)
  float lg8[8], lgn = 0.f;
  float2 ob8[8], obn = make_float2(0.f, 0.f);
  const int64_t urow = MASKED ? row / Kspk : 0;        // the utterance of this (utterance, speaker) row
  auto request = [&](int it_) __attribute__((always_inline)) {
    const int lf_ = it_ * 4 + wave;
    const int64_t t_ = t_lo + lf_;
    const bool ok = it_ < iters && t_ < T && lf_ < hops + 3;
    const int64_t tt = ok ? t_ : 0;
    const srd_t sl = make_srd(logit + (row * T + tt) * (NH + 1), ok ? (NH + 1) * 4 : 0);
    const srd_t so = make_srd(obs + (urow * T + tt) * (NH + 1), ok ? (NH + 1) * 8 : 0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      lg8[r] = bload1(sl, (unsigned)(lane * 4 + 256 * r));
      ob8[r] = bload2(so, (unsigned)(lane * 8 + 512 * r));
    }
    lgn = bload1(sl, (unsigned)(NH * 4));          // bin 512 (every lane the same address; lane 0 uses it)
    obn = bload2(so, (unsigned)(NH * 8));
  };
  if (MASKED) request(0);
  for (int it = 0; it < iters; ++it) {
    const int lf = it * 4 + wave;
    const int64_t t = t_lo + lf;
    const bool valid = t < T && lf < hops + 3;
    float* slot = fr[lf % RING];
    float2 xnyq = make_float2(0.f, 0.f);              // MASKED: X[512], needed by lane 0 only
    if (MASKED) {
      if (valid) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int k = lane + 64 * r;
          const float m = sigmoidf_mask(lg8[r]);
          buf[PADI(k)] = make_float2(ob8[r].x * m, ob8[r].y * m);
        }
        const float m = sigmoidf_mask(lgn);
        xnyq = make_float2(obn.x * m, obn.y * m);
        WAVE_SYNC();
      }
      __builtin_amdgcn_sched_barrier(0);
      request(it + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (valid) {
      float2 v[8];
      const float2* Xr = MASKED ? nullptr : X + (row * T + t) * (NH + 1);
      float2 wr[8];
      lane_twiddles(tw2_lane, wr);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        float2 xk = MASKED ? buf[PADI(k)] : Xr[k];
        float2 xm = MASKED ? (k == 0 ? xnyq : buf[PADI(NH - k)]) : Xr[NH - k];
        xm.y = -xm.y;
        if (k == 0) { xk.y = 0.f; xm.y = 0.f; }
        // (the 1/512 of the inverse transform rides on the 1/2 of the even / odd split: a power of two, exact)
        const float2 e = make_float2(hinv * (xk.x + xm.x), hinv * (xk.y + xm.y));
        const float2 w = make_float2(wr[r].x, -wr[r].y);
        const float2 o = cmul(make_float2(hinv * (xk.x - xm.x), hinv * (xk.y - xm.y)), w);
        // Zi = E + i O ; feed conj(Zi) to the forward FFT
        v[r] = make_float2(e.x - o.y, -(e.y + o.x));
      }
      if (MASKED) WAVE_SYNC();        // every lane has read the spectrum before the FFT reuses the line
      fft512_wave(v, buf, twl, lane);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        const float2 z = buf[PADI(n)];
        const float2 w = *reinterpret_cast<const float2*>(wsyn + 2 * n);
        *reinterpret_cast<float2*>(&slot[2 * n]) = make_float2(z.x * w.x, -z.y * w.y);
      }
      WAVE_SYNC();          // the wave's line is reused by its next frame; ring slots are disjoint
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        *reinterpret_cast<float2*>(&slot[2 * (lane + 64 * r)]) = make_float2(0.f, 0.f);
    }
    __syncthreads();        // frames <= 4 it + 3 are in the ring
    // emit the hops completed by this iteration: h in [4 it - 3, 4 it] (4 x 256 samples, 4 per thread).
    // The target samples are requested for all four hops first: interleaved with the stores of y the
    // compiler kept each load behind the previous store and waited for it (four serialised round trips
    // per iteration, a third of a workgroup's time).
    {
      // y and the target through buffer resources whose range ends with this chunk's samples: hops in front of
      // the chunk (h < 0 wraps to a huge offset) or behind it load 0 / are not stored -- no per-lane branches
      float tv[4], sv[4];
      bool em[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = 4 * it - 3 + q;
        em[q] = h >= 0 && h < hops && n0 + (int64_t)h * 256 + tid < N;
        tv[q] = tr ? bload1(str_, (unsigned)(((4 * it - 3 + q) * 256 + tid) * 4)) : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = 4 * it - 3 + q + RING * 2;           // (non-negative index into the ring; RING * 2 > 3)
        float s = fr[h % RING][tid + 768];
        s += fr[(h + 1) % RING][tid + 512];
        s += fr[(h + 2) % RING][tid + 256];
        s += fr[(h + 3) % RING][tid];
        sv[q] = s;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bstore1(sy, (unsigned)(((4 * it - 3 + q) * 256 + tid) * 4), sv[q]);
        if (tr && em[q]) asum += fabsf(sv[q] - tv[q]);
      }
    }
    __syncthreads();        // the next iteration overwrites the slots of frames <= 4 it
  }
  if (abs_partial) {
    asum = wave_sum(asum);
    if (lane == 0) red[wave] = asum;
    __syncthreads();
    if (tid == 0) abs_partial[row * nchunks + c] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

}  // namespace

extern "C" int64_t tssep_stft_frames(int64_t N, int size, int shift, int window_length, int pad,
                                     int fading) {
  if (window_length <= 0) window_length = size;
  int64_t n = N;
  if (fading) n += 2 * (int64_t)(window_length - shift);
  if (pad) {
    int64_t num = n - window_length;
    int64_t q = num <= 0 ? 0 : (num + shift - 1) / shift;
    return q + 1;
  }
  return (n - window_length) / shift + 1;
}

// the general plan (stft_generic.hip): every other size / shift the reference's `fe` slot may name
int tssep_generic_plan_supported(int size, int shift);
int tssep_generic_twiddles(int size, float* host_out);
int tssep_generic_rfft(const float* x, int64_t rows, int64_t N, int size, int shift, int pad_left, const float* window,
                       const float* tw, float* X, int64_t T, float s_in, float s_edge, void* stream);
int tssep_generic_istft(const float* X, int64_t rows, int64_t T, int size, int shift, int pad_left, const float* wsyn,
                        const float* tw, float* y, int64_t N, void* stream);

extern "C" int tssep_fft_twiddles(int size, float* host_out) {
  if (!host_out) return TSSEP_E_NULL;
  if (size != 1024) return tssep_generic_twiddles(size, host_out);      // (the same table layout for every plan)
  const int nh = size / 2;
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

static int check_plan(int size, int shift) {
  if (size != 1024 || shift != 256) return TSSEP_E_UNSUPPORTED;
  return TSSEP_OK;
}
static bool generic_plan(int size, int shift) { return check_plan(size, shift) != TSSEP_OK && tssep_generic_plan_supported(size, shift); }
extern "C" int tssep_stft_plan(int size, int shift) {
  return check_plan(size, shift) == TSSEP_OK ? 1 : (tssep_generic_plan_supported(size, shift) ? 2 : 0);
}

extern "C" int tssep_stft_fwd(const float* x, int64_t rows, int64_t N, int size, int shift,
                              int fading, const float* window, const float* tw, float* X,
                              int64_t T, void* stream) {
  if (!x || !window || !tw || !X) return TSSEP_E_NULL;
  if (rows <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if ((((uintptr_t)X) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)window) & 7u)) return TSSEP_E_ALIGN;
  if (generic_plan(size, shift))
    return tssep_generic_rfft(x, rows, N, size, shift, fading ? size - shift : 0, window, tw, X, T, 1.0f, 1.0f, stream);
  if (int e = check_plan(size, shift)) return e;
  const int iters = 4;
  const int64_t total = rows * T;
  const int64_t blocks = (total + 4 * iters - 1) / (4 * iters);
  hipLaunchKernelGGL(rfft_frames_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     x, rows, N, T, shift, fading ? size - shift : 0, window, (const float2*)tw,
                     (float2*)X, 1.0f, 1.0f, iters, (const float*)nullptr, (const float2*)nullptr,
                     (float*)nullptr, (int64_t)1, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const int32_t*)nullptr, 0);
  return tssep_launch_status();
}

extern "C" int tssep_istft_bwd(const float* dy, int64_t rows, int64_t N, int size, int shift,
                               int fading, const float* wsyn, const float* tw, float* dX,
                               int64_t T, void* stream) {
  if (!dy || !wsyn || !tw || !dX) return TSSEP_E_NULL;
  if (rows <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if ((((uintptr_t)dX) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)wsyn) & 7u)) return TSSEP_E_ALIGN;
  if (generic_plan(size, shift))
    return tssep_generic_rfft(dy, rows, N, size, shift, fading ? size - shift : 0, wsyn, tw, dX, T, 2.0f / (float)size,
                              1.0f / (float)size, stream);
  if (int e = check_plan(size, shift)) return e;
  const int iters = 4;
  const int64_t total = rows * T;
  const int64_t blocks = (total + 4 * iters - 1) / (4 * iters);
  hipLaunchKernelGGL(rfft_frames_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     dy, rows, N, T, shift, fading ? size - shift : 0, wsyn, (const float2*)tw,
                     (float2*)dX, 2.0f / (float)size, 1.0f / (float)size, iters, (const float*)nullptr,
                     (const float2*)nullptr, (float*)nullptr, (int64_t)1, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const int32_t*)nullptr, 0);
  return tssep_launch_status();
}

extern "C" int tssep_mask_istft_bwd(const float* dy, const float* logit, const float* obs, int64_t B,
                                    int64_t K, int64_t N, int size, int shift, int fading,
                                    const float* wsyn, const float* tw, float* dlogit, int64_t T,
                                    void* stream) {
  if (!dy || !logit || !obs || !wsyn || !tw || !dlogit) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if (int e = check_plan(size, shift)) return e;
  if ((((uintptr_t)obs) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)wsyn) & 7u)) return TSSEP_E_ALIGN;
  const int iters = 4;
  const int64_t rows = B * K, total = rows * T;
  const int64_t blocks = (total + 4 * iters - 1) / (4 * iters);
  hipLaunchKernelGGL(rfft_frames_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     dy, rows, N, T, shift, fading ? size - shift : 0, wsyn, (const float2*)tw,
                     (float2*)nullptr, 2.0f / (float)size, 1.0f / (float)size, iters, logit,
                     (const float2*)obs, dlogit, K, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const int32_t*)nullptr, 0);
  return tssep_launch_status();
}

extern "C" int tssep_mask_istft_bwd_loss(const float* est, const float* tgt, const float* sums, const float* gout,
                                         const float* logit, const float* obs, int64_t B, int64_t K, int64_t N,
                                         int size, int shift, int fading, const float* wsyn, const float* tw,
                                         const int32_t* iperm, int bt_major, float* dlogit, int64_t T,
                                         void* stream) {
  // tgt == NULL: `est` is dy itself (only the output layout is folded); sums == NULL: MAE
  if (!est || (tgt && !gout) || !logit || !obs || !wsyn || !tw || !dlogit) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if (int e = check_plan(size, shift)) return e;
  if ((((uintptr_t)obs) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)wsyn) & 7u)) return TSSEP_E_ALIGN;
  const int iters = 4;
  const int64_t rows = B * K, total = rows * T;
  const int64_t blocks = (total + 4 * iters - 1) / (4 * iters);
  hipLaunchKernelGGL(rfft_frames_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     est, rows, N, T, shift, fading ? size - shift : 0, wsyn, (const float2*)tw,
                     (float2*)nullptr, 2.0f / (float)size, 1.0f / (float)size, iters, logit,
                     (const float2*)obs, dlogit, K, tgt, sums, gout, iperm, bt_major);
  return tssep_launch_status();
}

// chunks per row: balanced, at most HCB_MAX hops each
extern "C" int64_t tssep_istft_chunks(int64_t N) {
  const int64_t hops = (N + 255) / 256;
  return (hops + HCB_MAX - 1) / HCB_MAX;
}
static int istft_hops_per_chunk(int64_t N, int nchunks) {
  const int64_t hops = (N + 255) / 256;
  return (int)((hops + nchunks - 1) / nchunks);
}

extern "C" int tssep_istft_fwd(const float* X, int64_t rows, int64_t T, int size, int shift,
                               int fading, const float* wsyn, const float* tw, float* y, int64_t N,
                               const float* tgt, float* abs_partial, void* stream) {
  if (!X || !wsyn || !tw || !y) return TSSEP_E_NULL;
  if (rows <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if (generic_plan(size, shift)) {
    // (the general plan has no fused |estimate - target| sums: a time-domain loss reads the estimate itself)
    if (tgt || abs_partial) return TSSEP_E_UNSUPPORTED;
    if ((((uintptr_t)X) & 7u) || (((uintptr_t)tw) & 7u)) return TSSEP_E_ALIGN;
    return tssep_generic_istft(X, rows, T, size, shift, fading ? size - shift : 0, wsyn, tw, y, N, stream);
  }
  if (int e = check_plan(size, shift)) return e;
  if (!fading) return TSSEP_E_UNSUPPORTED;
  if ((((uintptr_t)X) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)wsyn) & 7u)) return TSSEP_E_ALIGN;
  if (rows > 65535) return TSSEP_E_SHAPE;
  const int nchunks = (int)tssep_istft_chunks(N);
  hipLaunchKernelGGL(istft_kernel<false>, dim3((unsigned)nchunks, (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)X, T, shift, N, wsyn, (const float2*)tw, y,
                     tgt, abs_partial, nchunks, istft_hops_per_chunk(N, nchunks), (const float*)nullptr,
                     (const float2*)nullptr, (int64_t)1);
  return tssep_launch_status();
}

extern "C" int tssep_mask_istft_fwd(const float* logit, const float* obs, int64_t B, int64_t K,
                                    int64_t T, int size, int shift, int fading, const float* wsyn,
                                    const float* tw, float* y, int64_t N, const float* tgt,
                                    float* abs_partial, void* stream) {
  if (!logit || !obs || !wsyn || !tw || !y) return TSSEP_E_NULL;
  if (B <= 0 || K <= 0 || N <= 0 || T <= 0) return TSSEP_E_SHAPE;
  if (int e = check_plan(size, shift)) return e;
  if (!fading) return TSSEP_E_UNSUPPORTED;
  if ((((uintptr_t)obs) & 7u) || (((uintptr_t)tw) & 7u) || (((uintptr_t)wsyn) & 7u)) return TSSEP_E_ALIGN;
  const int64_t rows = B * K;
  if (rows > 65535) return TSSEP_E_SHAPE;
  const int nchunks = (int)tssep_istft_chunks(N);
  hipLaunchKernelGGL(istft_kernel<true>, dim3((unsigned)nchunks, (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)nullptr, T, shift, N, wsyn, (const float2*)tw, y,
                     tgt, abs_partial, nchunks, istft_hops_per_chunk(N, nchunks), logit, (const float2*)obs, K);
  return tssep_launch_status();
}
