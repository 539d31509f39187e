// Split-bf16 ("bf16x3") GEMM on the gfx950 bf16 matrix cores: fp32 operands in memory, fp32
// accumulation, fp32-class accuracy at 16/3 x the rate of the exact-fp32 MFMA.
//
// Same call sites as gemm.hip (nn.Linear / LSTM input projection and their backward GEMMs,
// tssep/train/rnnp.py:88-96,146-161; tssep/train/net.py:663-666).  Every fp32 operand element x
// is split while it is staged into LDS:  hi = bf16_rne(x), lo = bf16_rne(x - hi)  (x - hi is
// exact in fp32), and a product is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulators.  The dropped a_lo*b_lo term and the rounding
// of lo bound the per-product relative error by ~2^-16 -- three orders of magnitude inside the
// 1e-3 parity bar, and unlike plain bf16 it does not eat the LSTM's dynamic range.
//
// Two kernels:
//  * gemm_bf16x3_pipe_kernel -- 128x128 output tile, K staged 32 at a time, 256 threads (waves 2x2,
//    each 2x2 MFMA tiles -> 64 accumulator registers), every operand layout (row-major, k-major,
//    time-shifted k-major), split-K, all epilogues.
//  * gemm_bf16x3_tall_kernel -- 256x128 tile for row-major x row-major with M >= 1024.
// Both: two LDS stages (hi+lo, A+B), ONE barrier per K tile, register prefetch of the next-but-one
// tile, and the split / ds_write / global loads interleaved between the MFMAs by
// sched_group_barrier.  LDS rows are [row][BK bf16] with a pitch of 2*BK+16 bytes: the 16-byte MFMA
// fragment reads (lane = row) and the staging writes are bank-conflict free.  k-major
// ("transposed") operands are read from global memory lane = column (coalesced 4-byte loads), BK/2
// consecutive k per lane, so the transpose costs no extra LDS traffic.
// History (measured, profiles/r1_gemm_microbench_bf16x3.jsonl): a first version with one LDS
// stage and two barriers per tile ran 113-267 TFLOP/s; BK = 64 and naive double buffering changed
// nothing; the pipelined loop + row epilogue + tall tile run 190-320 TFLOP/s.
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;
// bytes per LDS row = 2*BK + 16: the 16 rows of a ds_read_b128 lane group land on 16 distinct
// 16-byte slots (5r mod 16 for BK = 32)
template <int BK> struct Cfg {
  static constexpr int PITCH = 2 * BK + 16;
  static constexpr int ARR = BM * PITCH;
  static constexpr int NL = BK / 8;              // float4 loads per thread, row operand
  static constexpr int KQ = BK / 4;              // float4 per row
  static constexpr int NC = BK / 2;              // scalar loads per thread, col operand
};

// ------------------------------------------------------------------------------------------------
// Software pipeline: two LDS stages in SEPARATE arrays (so the compiler knows that the fragment
// reads of stage t and the staging writes of stage t+1 never alias), one barrier per K tile, and
// the split (VALU) + ds_write of tile t+1 and the global loads of tile t+2 interleaved between the
// MFMAs of tile t.  A 32x32x16 MFMA occupies the matrix pipe for 32 cycles but its issue slot for
// 4: the ~110 VALU/LDS/VMEM instructions a wave needs per tile fit in the shadow of its 24 MFMAs.
template <int BK>
__device__ __forceinline__ void row_store_n(char* hi, char* lo, int tid, const f32x4 (&v)[Cfg<BK>::NL]) {
  constexpr int KQ = Cfg<BK>::KQ;
#pragma unroll
  for (int i = 0; i < Cfg<BK>::NL; ++i) {
    const int off = (tid / KQ + (NTHREADS / KQ) * i) * Cfg<BK>::PITCH + ((tid % KQ) << 3);
    unsigned h0, l0, h1, l1;
    split2n(v[i][0], v[i][1], h0, l0);
    split2n(v[i][2], v[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(hi + off) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(lo + off) = u32x2{l0, l1};
  }
}
template <int BK>
__device__ __forceinline__ void col_store_n(char* hi, char* lo, int tid, const float (&v)[Cfg<BK>::NC]) {
  constexpr int NC = Cfg<BK>::NC;
  const int off = (tid & 127) * Cfg<BK>::PITCH + (tid >> 7) * NC * 2;
#pragma unroll
  for (int c = 0; c < NC / 8; ++c) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2n(v[8 * c + 2 * e], v[8 * c + 2 * e + 1], h[e], l[e]);
    *reinterpret_cast<u32x4*>(hi + off + 16 * c) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(lo + off + 16 * c) = u32x4{l[0], l[1], l[2], l[3]};
  }
}

// ---- loads for the pipelined kernel: one uniform (SGPR) base per tile + 32-bit per-thread byte
// offsets, so a load is `global_load v, voff, s[base]` with no 64-bit VALU address arithmetic
template <int BK>
struct RowLoadU {
  const char* base;                 // P + r0 * ld   (uniform)
  unsigned off[Cfg<BK>::NL];        // ((row_i clamped) - r0) * ld + kq, bytes
  int kq;
};
template <int BK>
__device__ __forceinline__ RowLoadU<BK> make_row_load_u(const float* P, int64_t ld, int64_t R,
                                                        int64_t r0, int tid) {
  constexpr int NL = Cfg<BK>::NL, KQ = Cfg<BK>::KQ;
  RowLoadU<BK> d;
  d.kq = (tid % KQ) << 2;
  d.base = reinterpret_cast<const char*>(P + r0 * ld);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int64_t r = r0 + tid / KQ + (NTHREADS / KQ) * i;
    r = r > R - 1 ? R - 1 : r;
    d.off[i] = (unsigned)(((r - r0) * ld + d.kq) * 4);
  }
  return d;
}
template <int BK, bool TAIL>
__device__ __forceinline__ void row_load_u(const RowLoadU<BK>& d, int64_t k0, int64_t K,
                                           f32x4 (&v)[Cfg<BK>::NL]) {
  const int64_t k = k0 + d.kq;
  // (tail) a 16-byte load that starts at or beyond K would leave the row: read the row start
  const char* b = d.base + ((!TAIL || k < K) ? k0 : -(int64_t)d.kq) * 4;
#pragma unroll
  for (int i = 0; i < Cfg<BK>::NL; ++i) {
    f32x4 x = *reinterpret_cast<const f32x4*>(b + d.off[i]);
    if (TAIL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (k + e < K) ? x[e] : 0.f;
    }
    v[i] = x;
  }
}
struct ColLoadU {
  const char* base;        // P + c0   (uniform)
  unsigned off;            // ((c clamped) - c0 + khalf * NC * ld) * 4
  int khalf;
  int ph0;                 // SHIFT: phase (k % kperiod) of the first row of the NEXT tile to load
  bool cvalid;
  bool ones;               // this thread's column is the virtual all-ones column
  int ph_tile;             // phase of the first row of the tile currently held in registers
  bool premasked;          // that tile was masked while it was loaded (tail / generic shift)
};
template <int BK>
__device__ __forceinline__ ColLoadU make_col_load_u(const float* P, int64_t ld, int64_t C, int64_t c0,
                                                    int tid, int64_t k_first, int64_t kperiod,
                                                    bool ones_col = false) {
  ColLoadU d;
  int64_t c = c0 + (tid & 127);
  d.cvalid = c < C;
  d.ones = ones_col && c == C - 1;
  if (ones_col) C -= 1;                         // real columns of the matrix in memory
  if (c > C - 1) c = C - 1;
  d.khalf = tid >> 7;
  const int64_t cb = c0 < C ? c0 : C - 1;      // a tile may start AT the virtual column
  d.base = reinterpret_cast<const char*>(P + cb);
  d.off = (unsigned)(((c - cb) + d.khalf * (Cfg<BK>::NC) * ld) * 4);
  d.ph0 = kperiod > 0 ? (int)(k_first % kperiod) : 0;      // phase of the tile's first row (uniform)
  return d;
}
// tiles must be requested in increasing order, one call per tile (ph0 is advanced here).
// SHIFT: 0 = none; 1 = generic time shift (any period, any shift); 2 = fast path for |kshift| = 1
// and kperiod >= BK: a tile then holds at most ONE row whose shifted partner lies outside its
// sequence, at the tile-uniform position rbad.
// Full tiles without the generic shift are loaded RAW and masked by col_mask when they are staged
// one iteration later: a select at load time sits on the loop-carried value and makes the wave wait
// for its own prefetch inside the iteration that issued it (measured: the k-major GEMMs lost up to
// half their rate to that wait).
template <int BK, bool TAIL, int SHIFT>
__device__ __forceinline__ void col_load_u(ColLoadU& d, int64_t ld, int64_t k0, int64_t K,
                                           int kshift, int kperiod, float (&v)[Cfg<BK>::NC]) {
  constexpr int NC = Cfg<BK>::NC;
  d.ph_tile = d.ph0;
  if constexpr (TAIL) {           // last K tile only: per-thread 64-bit addresses, rows clamped
    const char* col = d.base + (d.off - (unsigned)((int64_t)d.khalf * NC * ld * 4));
    int ph = SHIFT ? (d.ph0 + d.khalf * NC) % kperiod : 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int64_t k = k0 + d.khalf * NC + i;
      bool ok = d.cvalid && k < K;
      int64_t kk = k < K ? k : 0;
      if (SHIFT) {
        const int q = ph + kshift;
        const bool in = q >= 0 && q < kperiod;
        if (ok && in) kk += kshift;
        ok = ok && in;
        ph = ph + 1 == kperiod ? 0 : ph + 1;
      }
      const float x = *reinterpret_cast<const float*>(col + kk * ld * 4);
      v[i] = d.ones ? (ok ? 1.f : 0.f) : (ok ? x : 0.f);
    }
    d.premasked = true;
  } else if constexpr (SHIFT == 1) {
    // generic shift: the address depends on the row's phase, so the mask is known here anyway
    const int64_t bias = (int64_t)(kshift < 0 ? -kshift : kshift) * ld * 4;
    const char* b = d.base + k0 * ld * 4 - bias;          // keeps the lane offsets non-negative
    const unsigned off0 = d.off + (unsigned)bias;
    const unsigned off_in = (unsigned)((int64_t)off0 + (int64_t)kshift * ld * 4);
    int ph = (d.ph0 + d.khalf * NC) % kperiod;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int q = ph + kshift;
      const bool in = q >= 0 && q < kperiod;
      const float x = *reinterpret_cast<const float*>(b + (int64_t)i * ld * 4 + (in ? off_in : off0));
      v[i] = (d.cvalid && in) ? x : 0.f;
      ph = ph + 1 == kperiod ? 0 : ph + 1;
    }
    d.premasked = true;
  } else {
    // raw: SHIFT == 2 reads every row at its shifted position (the caller -- the pipelined loop --
    // only asks for tiles whose shifted rows all lie inside the matrix)
    const char* p = d.base + (k0 + (SHIFT == 2 ? kshift : 0)) * ld * 4 + d.off;
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = *reinterpret_cast<const float*>(p + (int64_t)i * ld * 4);
    d.premasked = false;
  }
  if (SHIFT) d.ph0 = (d.ph0 + BK) % kperiod;
}
// masks of a tile that was loaded raw (see col_load_u)
template <int BK, int SHIFT>
__device__ __forceinline__ void col_mask(const ColLoadU& d, int kshift, int kperiod,
                                         float (&v)[Cfg<BK>::NC]) {
  constexpr int NC = Cfg<BK>::NC;
  const bool okc = d.cvalid && !d.ones;
  const float fill = d.ones ? 1.f : 0.f;            // the virtual column: every valid k reads 1
  // SHIFT == 2: position (relative to this lane's first row) of the tile's only bad row
  const int rbad = SHIFT == 2 ? (kshift < 0 ? (kperiod - d.ph_tile) % kperiod : kperiod - 1 - d.ph_tile) -
                                    d.khalf * NC
                              : -1;
#pragma unroll
  for (int i = 0; i < NC; ++i) v[i] = (okc && i != rbad) ? v[i] : fill;
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, SID)

template <int BK, bool A_KMAJOR, bool B_KMAJOR, int SHIFT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16x3_pipe_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M,
    int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t b_kshift, int64_t kperiod,
    const float* __restrict__ bias, int act, int accumulate, StoreMap sm, int splitk,
    int64_t c_split_stride, TileMap tmap, int b_ones_col) {
  static_assert(BK == 32, "the interleave pattern below is written for two 16-wide k-steps");
  constexpr int PITCH = Cfg<BK>::PITCH, ARR = Cfg<BK>::ARR;
  __shared__ __attribute__((aligned(16))) char lds0[4 * ARR];     // stage 0: A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) char lds1[4 * ARR];     // stage 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;

  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  RowLoadU<BK> ra_d, rb_d;
  ColLoadU ca_d, cb_d;
  if (!A_KMAJOR) ra_d = make_row_load_u<BK>(A, lda, M, m0, tid);
  else ca_d = make_col_load_u<BK>(A, lda, M, m0, tid, 0, 0);
  if (!B_KMAJOR) rb_d = make_row_load_u<BK>(B, ldb, N, n0, tid);
  else cb_d = make_col_load_u<BK>(B, ldb, N, n0, tid, kt_begin * BK, SHIFT ? kperiod : 0, b_ones_col != 0);
  const int kshift = (int)b_kshift, kper = (int)kperiod;

  // ONE register set for the raw (fp32) tile in flight: a second set (loads issued a whole
  // iteration earlier) was measured 10-40 % slower -- the kernel is not load-latency bound and the
  // extra 32 VGPRs cost more in schedule quality than the deeper prefetch returns.
  f32x4 ra[Cfg<BK>::NL], rb[Cfg<BK>::NL];
  float ca[Cfg<BK>::NC], cb[Cfg<BK>::NC];
  auto gload_full = [&](int64_t kt) {           // steady state: raw tiles (masked when staged)
    const int64_t k0 = kt * BK;
    if (!A_KMAJOR) row_load_u<BK, false>(ra_d, k0, K, ra); else col_load_u<BK, false, 0>(ca_d, lda, k0, K, 0, 1, ca);
    if (!B_KMAJOR) row_load_u<BK, false>(rb_d, k0, K, rb); else col_load_u<BK, false, SHIFT>(cb_d, ldb, k0, K, kshift, kper, cb);
  };
  constexpr int SHIFT_ANY = SHIFT == 2 ? 1 : SHIFT;      // prologue / drain: any tile, generic shift
  auto gload_any = [&](int64_t kt) {
    const int64_t k0 = kt * BK;
    if (kt < kt_full) {
      if (!A_KMAJOR) row_load_u<BK, false>(ra_d, k0, K, ra); else col_load_u<BK, false, 0>(ca_d, lda, k0, K, 0, 1, ca);
      if (!B_KMAJOR) row_load_u<BK, false>(rb_d, k0, K, rb); else col_load_u<BK, false, SHIFT_ANY>(cb_d, ldb, k0, K, kshift, kper, cb);
    } else {
      if (!A_KMAJOR) row_load_u<BK, true>(ra_d, k0, K, ra); else col_load_u<BK, true, 0>(ca_d, lda, k0, K, 0, 1, ca);
      if (!B_KMAJOR) row_load_u<BK, true>(rb_d, k0, K, rb); else col_load_u<BK, true, SHIFT_ANY>(cb_d, ldb, k0, K, kshift, kper, cb);
    }
  };
  auto stage = [&](char* st) {
    if (!A_KMAJOR) row_store_n<BK>(st, st + ARR, tid, ra); else col_store_n<BK>(st, st + ARR, tid, ca);
    if (!B_KMAJOR) row_store_n<BK>(st + 2 * ARR, st + 3 * ARR, tid, rb); else col_store_n<BK>(st + 2 * ARR, st + 3 * ARR, tid, cb);
  };
  auto sstore_raw = [&](char* st) {             // tiles loaded by gload_full (never premasked
    if (A_KMAJOR) col_mask<BK, 0>(ca_d, 0, 1, ca);                  // except the generic shift)
    if (B_KMAJOR && SHIFT != 1) col_mask<BK, SHIFT>(cb_d, kshift, kper, cb);
    stage(st);
  };
  auto sstore = [&](char* st) {                 // tiles loaded by gload_any
    if (A_KMAJOR && !ca_d.premasked) col_mask<BK, 0>(ca_d, 0, 1, ca);
    // (a raw B tile of a fast-shift kernel can only come from gload_full: gload_any premasks)
    if (B_KMAJOR && !cb_d.premasked) col_mask<BK, (SHIFT == 1 ? 0 : SHIFT)>(cb_d, kshift, kper, cb);
    stage(st);
  };
  const int foff = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int aoff = (wm * 64) * PITCH + foff, boff = 2 * ARR + (wn * 64) * PITCH + foff;
  // 24 MFMAs of one K tile out of stage `st`; consecutive MFMAs hit different accumulators
  auto compute = [&](const char* st) {
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];            // [k-step][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[ks][i] = *reinterpret_cast<const bf16x8*>(st + aoff + i * 32 * PITCH + ks * 32);
        al[ks][i] = *reinterpret_cast<const bf16x8*>(st + ARR + aoff + i * 32 * PITCH + ks * 32);
        bh[ks][i] = *reinterpret_cast<const bf16x8*>(st + boff + i * 32 * PITCH + ks * 32);
        bl[ks][i] = *reinterpret_cast<const bf16x8*>(st + ARR + boff + i * 32 * PITCH + ks * 32);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    }
  };
  // steady state: compute tile kt (stage cur), stage tile kt+1 (registers -> nxt), fetch tile kt+2
#define PIPE(cur, nxt, kt_, SID_)                                                               \
  do {                                                                                          \
    constexpr int SID = SID_;                                                                   \
    compute(cur);                                                                               \
    sstore_raw(nxt);                                                                            \
    gload_full((kt_) + 2);                                                                      \
    SGB(0x100, 8);                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x100, 1); SGB(0x002, 5); SGB(0x200, 1); } \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x002, 5); SGB(0x200, 1); }               \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x020, 4); }                              \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  if (kt_begin < kt_end) {
    gload_any(kt_begin);
    sstore(lds0);
    if (kt_begin + 1 < kt_end) gload_any(kt_begin + 1);
    __syncthreads();
    int64_t kt = kt_begin;
    // pipelined pairs while tiles kt+2 and kt+3 exist and are full
    int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;
    // fast shift: the pipelined loads read row k + kshift unconditionally, so they must stay clear
    // of the matrix's last tile (they never see the first: they start at tile kt_begin + 2)
    if (SHIFT == 2 && lim > (K - 1) / BK - 4) lim = (K - 1) / BK - 4;
    for (; kt < lim; kt += 2) {
      PIPE(lds0, lds1, kt, 1);
      PIPE(lds1, lds0, kt + 1, 2);
    }
    // drain (also the K tail): same stages, conditional work
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) sstore(nxt);
      if (kt + 2 < kt_end) gload_any(kt + 2);
      __syncthreads();
    }
  }
#undef PIPE
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  if (!sm.remap) {
    // both stages are free after the last barrier: waves 0,1 use stage 0, waves 2,3 stage 1
    static_assert(2 * 64 * EPITCH * 4 <= 4 * ARR, "epilogue scratch must fit in one stage");
    float* stage = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
    gemm_epilogue_rows(acc, stage, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane, bias,
                       act, accumulate, sm.ldc, splitk == 1, sm.aux, sm.ldaux);
    return;
  }
  {
    float* stage = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
    if (remap_vec_ok(sm, Cz))
      gemm_epilogue_rows_remap_vec(acc, stage, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane, bias,
                                   act, accumulate, sm);
    else if (remap_wide_ok(sm))
      gemm_epilogue_rows_remap_wide(acc, stage, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane,
                                    splitk == 1 ? bias : nullptr, splitk == 1 ? act : 0, accumulate, sm);
    else
    gemm_epilogue_rows_remap(acc, stage, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane,
                             splitk == 1 ? bias : nullptr, splitk == 1 ? act : 0, accumulate, sm);
  }
}
#undef SGB


// ------------------------------------------------------------------------------------------------
// "Tall" variant for row-major x row-major (both operands k-contiguous): 256 x 128 output tile,
// wave tile 128 x 64 (4 x 2 MFMA tiles, 128 accumulator registers), K staged 16 at a time.  Per
// MFMA it moves 25 % fewer operand bytes (global -> LDS -> fragments) and issues 25 % fewer
// split / ds_write / ds_read instructions than the 128 x 128 kernel, with the same 24 MFMAs per
// wave between barriers.  LDS rows are [row][16 bf16] with a 48-byte pitch (3 slots of 16 B: the 16
// rows of a ds_read_b128 lane group land on 16 distinct slots).  Measured +3..15 % on the shapes of
// the step.  Per-tile s_memtime profile at K = 320 (61 k cycles): prologue 4 %, K loop 54 % (1850
// cycles per stage for 768 MFMA cycles per wave: two resident workgroups keep the matrix pipe 83 %
// busy while both are in their loops), drain 5 %, epilogue 36 % (the 128 KB fp32 C tile: stores
// back-pressured by HBM).  De-phasing the two workgroups of a CU by a start delay changed nothing,
// neither did starting the first resident round in 8 phases spread over one tile time chip-wide;
// a persistent grid (512 workgroups walking the tile list) gained 5-9 % standalone on the K <= 600
// shapes and nothing in the training step; a BK = 16 variant of the 128 x 128 kernel at three
// workgroups per CU (152 VGPRs, 48 KB LDS) was 0-10 % slower than BK = 32 at two; half-width (128 x 64)
// edge tiles for N = 513 / 514 / 300 (20-25 % of the columns of the last 128-wide tile are padding)
// cost as much as the padding they remove -- as a second launch they also serialise behind the main
// one (+5..12 % on the split-K shapes).  Skipping only the MFMAs of the padded 32 x 32 sub-tiles inside the
// edge tile (N = 513: three quarters of its matrix work) changed nothing either (alternating A/B): an edge
// workgroup still stages the full A panel, and the stage is bounded by staging and barriers, not by MFMA.
// A K tile of 32 (whole 128-byte lines instead of half lines: half the L2 -> L1 line traffic) with ONE LDS
// stage of 61 KB and two barriers per tile (48 MFMAs between them, 246 VGPRs, no spills) measured 3-10 %
// SLOWER on the N = 2400 shapes and equal on the rest (alternating A/B): the second barrier costs more than the
// line traffic saves.
// WN = 4 ("wide", round 2): the same wave tile in a 256 x 256 workgroup tile of 8 waves (one workgroup per
// CU instead of two: the same 2 waves per SIMD).  The A panel -- the streamed activation matrix, of which
// a K tile of 16 fp32 uses only half of every 128-byte line it pulls through the L1 -- is then fetched
// once per 256 output columns instead of once per 128: a third less L2 -> L1 line traffic per MFMA
// (A + B lines per 16-k step and 256 x 256 outputs: 64 KB against 96 KB), and a thread stages 16
// instead of 24 raw values.  Used where the padding of N to a multiple of 256 costs < 10 %.
// (A second raw-tile register set in this variant -- loads of tile t + 3 issued during stage t, two stages of
// latency budget, 246 VGPRs, no spills -- measured 0..8 % SLOWER per shape and 0.4 % slower per step in an
// alternating A/B: the loop is not load-latency bound, as round 1 found for the 128 x 128 kernel.)
constexpr int TBM = 256, TBK = 16, TPITCH = 48;
constexpr int TARR_A = TBM * TPITCH;

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, SID)
// XCOL (wide tile only, N = 256 q + 1: the 513 frequency bins of `dgrad birnn0 dx` and of the pre-net
// projection): the MFMA tiles cover the first N - 1 columns -- two 256-wide tiles instead of five 128-wide ones
// of which the fifth staged the whole A panel once more for ONE column -- and column N - 1 is computed on the
// VALU from the raw fp32 A values every workgroup stages anyway (8 FMAs per thread and K tile against row
// N - 1 of B, exact fp32; every workgroup computes it -- no branch inside the loop body the scheduler
// interleaves, a uniform branch there measured 10 % slower -- and those of the last column tile store it).
// The 256 x 128 variant has no registers left for this (248 VGPRs: it spilled inside the loop).
// HACK (experiment builds only, -DTSSEP_GEMM_EXP, results are garbage -- TIMING probes of where a tile's life goes):
//   1 = operands addressed as if they were k-tile-major [K/16][rows][16] (every wave load = whole 128-B lines)
//   2 = no epilogue stores      4 = no MFMAs      8 = no global loads
//  16 = row-major full-line loads: half the rows, 32 k per row, alternating row halves (the access stream of a
//       BK = 32 pipeline with half-stage staging)
template <int WN, bool XCOL = false, int HACK = 0>
__global__ __launch_bounds__(128 * WN, 2) void gemm_bf16x3_tall_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M,
    int64_t Nfull, int64_t K, int64_t lda, int64_t ldb, const float* __restrict__ bias, int act,
    int accumulate, StoreMap sm, TileMap tmap) {
  const int64_t N = XCOL ? Nfull - 1 : Nfull;          // columns of the MFMA tiles
  constexpr int NT = 128 * WN, TBN = 64 * WN, TARR_B = TBN * TPITCH;
  constexpr int TSTAGE = 2 * TARR_A + 2 * TARR_B;        // A hi, A lo, B hi, B lo = 36 864 B (WN = 2) / 49 152 B
  constexpr int NA = TBM * 4 / NT, NB = TBN * 4 / NT;    // 16-byte loads per thread and K tile: 4 + 2 / 2 + 2
  constexpr int RSTEP = NT / 4;                          // rows covered by one load of all threads
  __shared__ __attribute__((aligned(16))) char lds0[TSTAGE];
  __shared__ __attribute__((aligned(16))) char lds1[TSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * TBM, n0 = (int64_t)nt * TBN;
  const int64_t ktiles = (K + TBK - 1) / TBK, kt_full = K / TBK;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: thread <-> (row tid/4 + 64 i, 4 consecutive k at (tid%4)*4); uniform base + 32-bit offsets
  // rows of an 8-row block are visited 0,2,4,6,1,3,5,7: the 4 rows one ds_write_b64 lane group stages
  // are then 96 B apart and cover all 32 banks once (consecutive rows at the 48-byte pitch overlap:
  // SQ_LDS_BANK_CONFLICT was 11 % of the kernel's CU-busy cycles)
  const int kq = (tid & 3) << 2;
  const int lrow = ((tid >> 2) & ~7) | (((tid >> 2) & 3) << 1) | ((tid >> 4) & 1);
  const char* abase = reinterpret_cast<const char*>(A + m0 * ((HACK & 1) ? 16 : lda));
  const char* bbase = reinterpret_cast<const char*>(B + n0 * ((HACK & 1) ? 16 : ldb));
  unsigned aoffs[NA], boffs[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int64_t r = m0 + lrow + RSTEP * i;
    r = r > M - 1 ? M - 1 : r;
    aoffs[i] = (unsigned)(((r - m0) * ((HACK & 1) ? 16 : lda) + kq) * 4);
    if (HACK & 16) aoffs[i] = (unsigned)((((tid >> 3) + (NT / 8) * i) * lda + ((tid & 7) << 2)) * 4);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int64_t r = n0 + lrow + RSTEP * i;
    r = r > N - 1 ? N - 1 : r;
    boffs[i] = (unsigned)(((r - n0) * ((HACK & 1) ? 16 : ldb) + kq) * 4);
    if (HACK & 16) boffs[i] = (unsigned)((((tid >> 3) + (NT / 8) * i) * ldb + ((tid & 7) << 2)) * 4);
  }
  f32x4 ra[NA], rb[NB];
  const bool xwg = XCOL && nt == tmap.NT - 1;           // workgroup-uniform
  const srd_t asrd = make_srd(abase), bsrd = make_srd(bbase), xsrd = make_srd(B + N * ldb);
  f32x4 rx = {0.f, 0.f, 0.f, 0.f};
  float xacc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) xacc[i] = 0.f;
  auto gload = [&](int64_t kt, bool tail, auto xtag) {
    const int64_t k0 = kt * TBK, k = k0 + kq;
    // (tail) a 16-byte load that starts at or beyond K would leave the row: read the row start
    const unsigned fix = (!tail || k < K) ? 0u : (unsigned)(-(k0 + kq) * 4);     // per lane, wraps with the offsets
    int so = (int)(k0 * 4), sob = so;
    if (HACK & 1) { so = (int)(kt * M * 64); sob = (int)(kt * Nfull * 64); }
    if (HACK & 16) {      // rows (kt & 1) * TBM/2 .. of the tile, k = 32 (kt / 2) .. + 31
      so = (int)((kt >> 1) * 128 + (kt & 1) * (TBM / 2) * lda * 4);
      sob = (int)((kt >> 1) * 128 + (kt & 1) * (TBN / 2) * ldb * 4);
    }
    if (!(HACK & 8)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = bload4(asrd, aoffs[i] + fix, so);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = bload4(bsrd, boffs[i] + fix, sob);
    }
    if constexpr (decltype(xtag)::value) rx = bload4(xsrd, (unsigned)(kq * 4) + fix, so);   // (ra is zeroed in the tail)
    if (tail) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = k + e < K;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i][e] = ok ? ra[i][e] : 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i][e] = ok ? rb[i][e] : 0.f;
      }
    }
  };
  const int soff = lrow * TPITCH + ((tid & 3) << 3);
  auto sstore = [&](char* st, auto xtag) {
    if constexpr (decltype(xtag)::value) {
#pragma unroll
      for (int i = 0; i < NA; ++i)
        xacc[i] = fmaf(ra[i][3], rx[3], fmaf(ra[i][2], rx[2], fmaf(ra[i][1], rx[1], fmaf(ra[i][0], rx[0], xacc[i]))));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      unsigned h0, l0, h1, l1;
      split2n(ra[i][0], ra[i][1], h0, l0);
      split2n(ra[i][2], ra[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soff + i * RSTEP * TPITCH) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + TARR_A + soff + i * RSTEP * TPITCH) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      unsigned h0, l0, h1, l1;
      split2n(rb[i][0], rb[i][1], h0, l0);
      split2n(rb[i][2], rb[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + 2 * TARR_A + soff + i * RSTEP * TPITCH) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + 2 * TARR_A + TARR_B + soff + i * RSTEP * TPITCH) = u32x2{l0, l1};
    }
  };
  const int foff = (lane & 31) * TPITCH + (lane >> 5) * 16;
  const int aoff = (wm * 128) * TPITCH + foff, boff = 2 * TARR_A + (wn * 64) * TPITCH + foff;
  auto compute = [&](const char* st) {
    bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(st + aoff + i * 32 * TPITCH);
      al[i] = *reinterpret_cast<const bf16x8*>(st + TARR_A + aoff + i * 32 * TPITCH);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * TPITCH);
      bl[j] = *reinterpret_cast<const bf16x8*>(st + TARR_B + boff + j * 32 * TPITCH);
    }
    if (HACK & 4) {       // keep the fragment reads alive without matrix work
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[i][0][0] += (float)ah[i][0] + (float)al[i][1]; }
#pragma unroll
      for (int j = 0; j < 2; ++j) { acc[0][j][1] += (float)bh[j][0] + (float)bl[j][1]; }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };
#define TPIPE(cur, nxt, kt_, SID_)                                                              \
  do {                                                                                          \
    constexpr int SID = SID_;                                                                   \
    compute(cur);                                                                               \
    sstore(nxt, xtag);                                                                          \
    gload((kt_) + 2, false, xtag);                                                              \
    SGB(0x100, 6);                                                                              \
    if (WN == 2) {                                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) { SGB(0x008, 1); SGB(0x100, 1); SGB(0x002, 4); SGB(0x200, 1); } \
      _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) { SGB(0x008, 1); SGB(0x002, 4); SGB(0x200, 1); }              \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x020, 1); }                              \
    } else {      /* 8 ds_writes, 4 global loads, two thirds of the split VALU per thread */   \
      _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) { SGB(0x008, 1); SGB(0x100, 1); SGB(0x002, 4); SGB(0x200, 1); } \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) { SGB(0x008, 1); SGB(0x002, 4); SGB(0x200, 1); }               \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x002, 3); }                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { SGB(0x008, 1); SGB(0x020, 1); }                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { SGB(0x008, 1); }                                             \
    }                                                                                           \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  const std::integral_constant<bool, XCOL> xtag{};
  gload(0, 0 >= kt_full, xtag);
  sstore(lds0, xtag);
  if (1 < ktiles) gload(1, 1 >= kt_full, xtag);
  __syncthreads();
  int64_t kt = 0;
  const int64_t lim = kt_full - 3;
  for (; kt < lim; kt += 2) {
    TPIPE(lds0, lds1, kt, 1);
    TPIPE(lds1, lds0, kt + 1, 2);
  }
  for (int par = 0; kt < ktiles; ++kt, par ^= 1) {
    const char* cur = par ? lds1 : lds0;
    char* nxt = par ? lds0 : lds1;
    compute(cur);
    if (kt + 1 < ktiles) sstore(nxt, xtag);
    if (kt + 2 < ktiles) gload(kt + 2, kt + 2 >= kt_full, xtag);
    __syncthreads();
  }
  if (XCOL && xwg) {
    // column N of the full matrix: the 4 lanes of a row hold its four k quarters
    const int64_t n = N;
    const float bv = bias ? bias[n] : 0.f;
    const int64_t cq = sm.remap ? n / sm.cm : 0, cr = sm.remap ? n - cq * sm.cm : n;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float v = xacc[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const int64_t m = m0 + lrow + RSTEP * i;
      if ((tid & 3) == 0 && m < M) {
        int64_t a;
        if (sm.remap) {
          const int64_t t = m % sm.T, q = m / sm.T;
          const int64_t k = q % sm.K, b = q / sm.K;
          const int64_t cqq = sm.perm ? (int64_t)sm.perm[b * sm.perm_ld + cq] : cq;
          a = b * sm.sb + k * sm.sk + t * sm.st + cqq * sm.co + cr;
        } else {
          a = m * sm.ldc + n;
        }
        v += bv;
        if (act == 1) v = gemm_tanh(v);
        if (accumulate) v += C[a];
        C[a] = v;
      }
    }
  }
#undef TPIPE
  // epilogue: every wave transposes its 64 x 64 blocks through a private LDS scratch (17 KB); a stage
  // array holds two of them, so the 8 waves of the wide tile take turns in two phases of 4
  static_assert(2 * 64 * EPITCH * 4 <= TSTAGE, "epilogue scratch must fit in one stage");
  float* stage = reinterpret_cast<float*>((wave & 2) ? lds1 : lds0) + (wave & 1) * 64 * EPITCH;
  if ((HACK & 2) && K >= 0) {        // no stores (K >= 0 always: the accumulators stay live)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
    if (sacc == 123.456f) C[tid] = sacc;
    return;
  }
#pragma unroll 1
  for (int phase = 0; phase < WN / 2; ++phase) {
    if (WN > 2 && phase) __syncthreads();
    if ((wave >> 2) != phase) continue;
#pragma unroll
    for (int ih = 0; ih < 2; ++ih) {
      f32x16 a2[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) a2[i][j] = acc[2 * ih + i][j];
      if (!sm.remap)
        gemm_epilogue_rows(a2, stage, C, M, (HACK & 32) ? 0 : N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64, lane,
                           bias, act, (HACK & 64) ? 2 : accumulate, sm.ldc, true, sm.aux, sm.ldaux);
      else if (remap_vec_ok(sm, C))
        gemm_epilogue_rows_remap_vec(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64,
                                     lane, bias, act, accumulate, sm);
      else if (remap_wide_ok(sm))
        gemm_epilogue_rows_remap_wide(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64,
                                      lane, bias, act, accumulate, sm);
      else
        gemm_epilogue_rows_remap(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64,
                                 lane, bias, act, accumulate, sm);
    }
  }
}
#undef SGB


// ------------------------------------------------------------------------------------------------
// Weight-gradient variant: C[M,N] = A^T B with BOTH operands k-major (A = dY [K, M], B = X [K, N],
// rows = time steps).  The generic kernel above reads such operands lane = column with 4-byte loads
// and transposes them in registers (32 scalar loads per wave and tile).  Here a tile is read the
// way it lies in memory -- 16-byte loads along m / n, one k row per 32 lanes -- split, and written
// UNTRANSPOSED to LDS as [k][m] bf16 rows (pitch 320 B: 4 consecutive k rows cover all 64 banks
// once); the MFMA fragments (lane = m, 8 consecutive k) are produced by the LDS transpose read
// ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block, 4 contiguous m per lane, and lane
// i receives column i (4 k values) -- two reads per 8-k fragment.  Masks (K tail, column tail, time
// shift, virtual ones column) are applied when a tile is staged, never on the load path.
// Time shift: row k of B is replaced by row k + kshift of the same period (zero outside it).
// The loop body of the unshifted variant is left to the compiler's scheduler: a hand-written
// sched_group_barrier interleave (MFMA / transpose read / 4 VALU / ds_write ...) measured 5-8 % SLOWER
// there in an alternating A/B of two builds (three other interleaves 0-3 % slower than none; MFMA busy
// 44 -> 54 % in the step); the shifted variant keeps the interleave (44 % with, 40 % without).
constexpr int TNP = 320;                 // bytes per k row: 128 m x 2 B + 64
constexpr int TNARR = 32 * TNP;          // one array (32 k rows) = 10 240 B
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * TNP));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, SID)
// TWO = true (opt-in, TSSEP_WGRAD_PRODUCTS=2): the a_lo * b_hi product is dropped -- dY enters as plain
// bf16, X keeps hi + lo -- 16 instead of 24 MFMAs per K tile, no lo plane of A staged.  Every term of a
// weight-gradient sum then carries a relative error of up to 2^-9 instead of 2^-16; over 2e5 .. 8e5 rows
// the errors average, measured: DESIGN.md section 5.
template <bool SHIFT, bool TWO = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16x3_tn_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M,
    int64_t N, int64_t K, int64_t lda, int64_t ldb, int kshift, int kperiod, int accumulate,
    int64_t ldc, int splitk, int64_t c_split_stride, TileMap tmap, int b_ones_col) {
  constexpr int BK = 32;
  __shared__ __attribute__((aligned(16))) char lds0[4 * TNARR];     // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) char lds1[4 * TNARR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * BM, n0 = (int64_t)nt * BN;
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: thread <-> (k row tid/32 + 8 i, columns 4 (tid%32) .. +3)
  const int krow = tid >> 5, cq = (tid & 31) << 2;
  const int64_t Nreal = N - (b_ones_col ? 1 : 0);
  // column starts clamped so that the 16-byte load stays inside the row (host: round_up(M,4) <= lda)
  const int64_t Mp = (M + 3) & ~(int64_t)3, Np = (Nreal + 3) & ~(int64_t)3;
  const int64_t ca = m0 + cq <= Mp - 4 ? m0 + cq : Mp - 4;
  const int64_t cb = n0 + cq <= Np - 4 ? n0 + cq : Np - 4;
  const char* abase = reinterpret_cast<const char*>(A);
  const char* bbase = reinterpret_cast<const char*>(B);
  const int64_t aoff0 = (krow * lda + ca) * 4, boff0 = (krow * ldb + cb) * 4;       // + k0 * ld * 4
  // per-element column masks of this thread's four columns
  bool am[4], bm[4], bone[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    am[e] = m0 + cq + e < M;
    bm[e] = n0 + cq + e < Nreal;
    bone[e] = b_ones_col && n0 + cq + e == N - 1;
  }
  // time shift: phase (k mod kperiod) of this thread's rows of the tile held in registers
  int ph[4] = {0, 0, 0, 0};
  const int phstep = SHIFT ? BK % kperiod : 0;
  if (SHIFT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ph[i] = (int)((kt_begin * BK + krow + 8 * i) % kperiod);
  }
  bool kok[4] = {true, true, true, true};       // row < K, of the tile held in registers

  f32x4 ra[4], rb[4];
  // steady state: rows k0 .. k0+31 (and their shifted partners) all lie inside the matrix
  auto gload_full = [&](int64_t kt) {
    const char* pa = abase + kt * BK * lda * 4 + aoff0;
    const char* pb = bbase + (kt * BK + (SHIFT ? kshift : 0)) * ldb * 4 + boff0;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(pa + (int64_t)i * 8 * lda * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb + (int64_t)i * 8 * ldb * 4);
  };
  // prologue / drain: any tile; rows clamped into the matrix (the masks zero what was clamped)
  auto gload_any = [&](int64_t kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = kt * BK + krow + 8 * i;
      const int64_t ka = k < K ? k : K - 1;
      int64_t kb = ka + (SHIFT ? kshift : 0);
      kb = kb < 0 ? 0 : (kb > K - 1 ? K - 1 : kb);
      ra[i] = *reinterpret_cast<const f32x4*>(abase + (ka * lda + ca) * 4);
      rb[i] = *reinterpret_cast<const f32x4*>(bbase + (kb * ldb + cb) * 4);
    }
  };
  // bookkeeping for the tile that was just requested (phases advance tile by tile)
  int64_t held = kt_begin - 1;
  auto note_tile = [&](int64_t kt, bool full) {
    if (SHIFT && held >= kt_begin) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ph[i] += phstep; ph[i] = ph[i] >= kperiod ? ph[i] - kperiod : ph[i]; }
    }
    held = kt;
#pragma unroll
    for (int i = 0; i < 4; ++i) kok[i] = full || kt * BK + krow + 8 * i < K;
  };
  const int soff = krow * TNP + (tid & 31) * 8;
  // EDGE = false: an interior tile in the steady state -- every row < K and every column valid, only
  // the time shift still masks rows (64 selects per thread and tile less)
  auto stage = [&](char* st, auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 a = ra[i], b = rb[i];
      if constexpr (EDGE) {
        bool okb = kok[i];
        if (SHIFT) { const int q = ph[i] + kshift; okb = okb && q >= 0 && q < kperiod; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = (kok[i] && am[e]) ? a[e] : 0.f;
          b[e] = (okb && bm[e]) ? b[e] : ((bone[e] && kok[i]) ? 1.f : 0.f);
        }
      } else if constexpr (SHIFT) {
        const int q = ph[i] + kshift;
        const bool okb = q >= 0 && q < kperiod;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = okb ? b[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split2n(a[0], a[1], h0, l0);
      split2n(a[2], a[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soff + i * 8 * TNP) = u32x2{h0, h1};
      if (!TWO) *reinterpret_cast<u32x2*>(st + TNARR + soff + i * 8 * TNP) = u32x2{l0, l1};
      split2n(b[0], b[1], h0, l0);
      split2n(b[2], b[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + 2 * TNARR + soff + i * 8 * TNP) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + 3 * TNARR + soff + i * 8 * TNP) = u32x2{l0, l1};
    }
  };
  // fragment address of this lane: 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int foff = (8 * hk + (ii >> 2)) * TNP + (16 * g2 + 4 * (ii & 3)) * 2;
  const int aoff = foff + wm * 64 * 2, boff = 2 * TNARR + foff + wn * 64 * 2;
  auto compute = [&](const char* st) {
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];            // [k-step][tile]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[ks][i] = tr_frag(st + aoff + ks * 16 * TNP + i * 64);
        if (!TWO) al[ks][i] = tr_frag(st + TNARR + aoff + ks * 16 * TNP + i * 64);
        bh[ks][i] = tr_frag(st + boff + ks * 16 * TNP + i * 64);
        bl[ks][i] = tr_frag(st + TNARR + boff + ks * 16 * TNP + i * 64);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (!TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    }
  };
#define TNPIPE(cur, nxt, kt_, SID_, EDGE_)                                                      \
  do {                                                                                          \
    compute(cur);                                                                               \
    stage(nxt, std::integral_constant<bool, EDGE_>{});                                          \
    gload_full((kt_) + 2);                                                                      \
    note_tile((kt_) + 2, true);                                                                 \
    if constexpr (SHIFT) {     /* (the shifted variant: 44 % MFMA busy with this interleave, 40 % without) */ \
      constexpr int SID = SID_;                                                                 \
      SGB(0x100, 16);                                                                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { SGB(0x008, 1); SGB(0x100, 1); SGB(0x002, 4); SGB(0x200, 1); } \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { SGB(0x008, 1); SGB(0x002, 4); SGB(0x020, 1); }               \
    }                                                                                           \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  if (kt_begin < kt_end) {
    gload_any(kt_begin);
    note_tile(kt_begin, kt_begin < kt_full);
    stage(lds0, std::true_type{});
    if (kt_begin + 1 < kt_end) { gload_any(kt_begin + 1); note_tile(kt_begin + 1, kt_begin + 1 < kt_full); }
    __syncthreads();
    int64_t kt = kt_begin;
    int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;
    // the pipelined loads read row k + kshift unconditionally: stay clear of the matrix's last tiles
    // (they never see the first ones: they start at tile kt_begin + 2)
    if (SHIFT && lim > (K - 1) / BK - 4) lim = (K - 1) / BK - 4;
    // (the tile staged first in the loop, kt + 1, was loaded by gload_any: rows < K there as well)
    // (the shifted variant always takes the masked loop: its second copy of the loop cost 3 points of
    // MFMA-busy time -- code size -- and its row masks stay anyway)
    const bool edge = SHIFT || m0 + BM > M || n0 + BN > Nreal;
    if (edge) {
      for (; kt < lim; kt += 2) {
        TNPIPE(lds0, lds1, kt, 1, true);
        TNPIPE(lds1, lds0, kt + 1, 2, true);
      }
    } else {
      for (; kt < lim; kt += 2) {
        TNPIPE(lds0, lds1, kt, 3, false);
        TNPIPE(lds1, lds0, kt + 1, 4, false);
      }
    }
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) stage(nxt, std::true_type{});
      if (kt + 2 < kt_end) { gload_any(kt + 2); note_tile(kt + 2, kt + 2 < kt_full); }
      __syncthreads();
    }
  }
#undef TNPIPE
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  static_assert(2 * 64 * EPITCH * 4 <= 4 * TNARR, "epilogue scratch must fit in one stage");
  float* stg = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
  gemm_epilogue_rows(acc, stg, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane, nullptr, 0,
                     accumulate, ldc, splitk == 1);
}
#undef SGB

// ------------------------------------------------------------------------------------------------
// "Tall" weight-gradient variant: 256 (m) x 128 (n) output tile, wave tile 128 x 64, K staged 16 rows at a time
// -- the row x row kernel's geometry applied to the k-major x k-major case.  Per MFMA it stages a quarter fewer
// operand values (6 instead of 8 16-byte loads per thread and 24 MFMAs), issues a quarter fewer transpose reads
// (24 instead of 32) and splits a quarter fewer values than the 128 x 128 kernel above; the d(gates) panel
// (M = 2400 / 1200: every LSTM weight gradient) is split once per 128 output columns as before, the X panel once
// per 256 instead of once per 128 gate columns.  Same staging ([k][m] bf16 rows, pitch 576 B for the 256-wide
// A rows / 320 B for B, transpose reads), same masks, same pipeline (two LDS stages, one barrier per K tile,
// loads of tile t + 2 in flight), same k order per output element -> bit-identical to the 128 x 128 kernel.
// Used for the time-shifted dW_hh GEMMs (M = 1200: padding to 256 costs 6.7 %), where it takes 2.3 ms off the
// step; for the unshifted ones (M = 2400) it measured 4-10 % slower standalone and 0.5 ms slower per step.
constexpr int TTM = 256, TTK = 16;
constexpr int TPA = 576;                 // bytes per k row of an A plane: 256 m x 2 B + 64
constexpr int TTARR_A = TTK * TPA, TTARR_B = TTK * TNP;
constexpr int TTSTAGE = 2 * 64 * EPITCH * 4;      // 34 816 B: the planes need 28 672, the epilogue two 64 x 64 scratches
static_assert(2 * TTARR_A + 2 * TTARR_B <= TTSTAGE, "planes must fit in a stage");

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag_p(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool SHIFT, bool TWO = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16x3_tn_tall_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M,
    int64_t N, int64_t K, int64_t lda, int64_t ldb, int kshift, int kperiod, int accumulate,
    int64_t ldc, int splitk, int64_t c_split_stride, TileMap tmap, int b_ones_col) {
  constexpr int BK = TTK;
  __shared__ __attribute__((aligned(16))) char lds0[TTSTAGE];      // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) char lds1[TTSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * TTM, n0 = (int64_t)nt * BN;
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: A thread <-> (k row tid/64 + 4 i, columns 4 (tid%64) .. +3), i < 4;
  //        B thread <-> (k row tid/32 + 8 i, columns 4 (tid%32) .. +3), i < 2
  const int krA = tid >> 6, cqA = (tid & 63) << 2;
  const int krB = tid >> 5, cqB = (tid & 31) << 2;
  const int64_t Nreal = N - (b_ones_col ? 1 : 0);
  const int64_t Mp = (M + 3) & ~(int64_t)3, Np = (Nreal + 3) & ~(int64_t)3;
  const int64_t ca = m0 + cqA <= Mp - 4 ? m0 + cqA : Mp - 4;
  const int64_t cb = n0 + cqB <= Np - 4 ? n0 + cqB : Np - 4;
  const char* abase = reinterpret_cast<const char*>(A);
  const char* bbase = reinterpret_cast<const char*>(B);
  const int64_t aoff0 = (krA * lda + ca) * 4, boff0 = (krB * ldb + cb) * 4;       // + k0 * ld * 4
  bool am[4], bm[4], bone[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    am[e] = m0 + cqA + e < M;
    bm[e] = n0 + cqB + e < Nreal;
    bone[e] = b_ones_col && n0 + cqB + e == N - 1;
  }
  int ph[2] = {0, 0};                      // phase (k mod kperiod) of this thread's B rows of the tile in registers
  const int phstep = SHIFT ? BK % kperiod : 0;
  if (SHIFT) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ph[i] = (int)((kt_begin * BK + krB + 8 * i) % kperiod);
  }
  bool kokA[4] = {true, true, true, true}, kokB[2] = {true, true};    // row < K, of the tile held in registers

  f32x4 ra[4], rb[2];
  auto gload_full = [&](int64_t kt) {
    const char* pa = abase + kt * BK * lda * 4 + aoff0;
    const char* pb = bbase + (kt * BK + (SHIFT ? kshift : 0)) * ldb * 4 + boff0;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(pa + (int64_t)i * 4 * lda * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb + (int64_t)i * 8 * ldb * 4);
  };
  auto gload_any = [&](int64_t kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = kt * BK + krA + 4 * i;
      ra[i] = *reinterpret_cast<const f32x4*>(abase + ((k < K ? k : K - 1) * lda + ca) * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t k = kt * BK + krB + 8 * i;
      int64_t kb = (k < K ? k : K - 1) + (SHIFT ? kshift : 0);
      kb = kb < 0 ? 0 : (kb > K - 1 ? K - 1 : kb);
      rb[i] = *reinterpret_cast<const f32x4*>(bbase + (kb * ldb + cb) * 4);
    }
  };
  int64_t held = kt_begin - 1;
  auto note_tile = [&](int64_t kt, bool full) {
    if (SHIFT && held >= kt_begin) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { ph[i] += phstep; ph[i] = ph[i] >= kperiod ? ph[i] - kperiod : ph[i]; }
    }
    held = kt;
#pragma unroll
    for (int i = 0; i < 4; ++i) kokA[i] = full || kt * BK + krA + 4 * i < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) kokB[i] = full || kt * BK + krB + 8 * i < K;
  };
  const int soffA = krA * TPA + (tid & 63) * 8, soffB = 2 * TTARR_A + krB * TNP + (tid & 31) * 8;
  auto stage = [&](char* st, auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 a = ra[i];
      if constexpr (EDGE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (kokA[i] && am[e]) ? a[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split2n(a[0], a[1], h0, l0);
      split2n(a[2], a[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffA + i * 4 * TPA) = u32x2{h0, h1};
      if (!TWO) *reinterpret_cast<u32x2*>(st + TTARR_A + soffA + i * 4 * TPA) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 b = rb[i];
      if constexpr (EDGE) {
        bool okb = kokB[i];
        if (SHIFT) { const int q = ph[i] + kshift; okb = okb && q >= 0 && q < kperiod; }
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (okb && bm[e]) ? b[e] : ((bone[e] && kokB[i]) ? 1.f : 0.f);
      } else if constexpr (SHIFT) {
        const int q = ph[i] + kshift;
        const bool okb = q >= 0 && q < kperiod;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = okb ? b[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split2n(b[0], b[1], h0, l0);
      split2n(b[2], b[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffB + i * 8 * TNP) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + TTARR_B + soffB + i * 8 * TNP) = u32x2{l0, l1};
    }
  };
  // fragment address of this lane: 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * TPA + fcol + wm * 128 * 2, boff = 2 * TTARR_A + frow * TNP + fcol + wn * 64 * 2;
  auto compute = [&](const char* st) {
    bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = tr_frag_p<TPA>(st + aoff + i * 64);
      if (!TWO) al[i] = tr_frag_p<TPA>(st + TTARR_A + aoff + i * 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = tr_frag_p<TNP>(st + boff + j * 64);
      bl[j] = tr_frag_p<TNP>(st + TTARR_B + boff + j * 64);
    }
    if (!TWO) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };
#define TTPIPE(cur, nxt, kt_, EDGE_)                                                            \
  do {                                                                                          \
    compute(cur);                                                                               \
    stage(nxt, std::integral_constant<bool, EDGE_>{});                                          \
    gload_full((kt_) + 2);                                                                      \
    note_tile((kt_) + 2, true);                                                                 \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  if (kt_begin < kt_end) {
    gload_any(kt_begin);
    note_tile(kt_begin, kt_begin < kt_full);
    stage(lds0, std::true_type{});
    if (kt_begin + 1 < kt_end) { gload_any(kt_begin + 1); note_tile(kt_begin + 1, kt_begin + 1 < kt_full); }
    __syncthreads();
    int64_t kt = kt_begin;
    int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;
    // the pipelined loads read row k + kshift unconditionally (|kshift| <= 16 here): stay clear of the matrix's
    // last tiles (they never see the first ones: they start at tile kt_begin + 2)
    if (SHIFT && lim > (K - 1) / BK - 4) lim = (K - 1) / BK - 4;
    const bool edge = SHIFT || m0 + TTM > M || n0 + BN > Nreal;
    if (edge) {
      for (; kt < lim; kt += 2) {
        TTPIPE(lds0, lds1, kt, true);
        TTPIPE(lds1, lds0, kt + 1, true);
      }
    } else {
      for (; kt < lim; kt += 2) {
        TTPIPE(lds0, lds1, kt, false);
        TTPIPE(lds1, lds0, kt + 1, false);
      }
    }
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) stage(nxt, std::true_type{});
      if (kt + 2 < kt_end) { gload_any(kt + 2); note_tile(kt + 2, kt + 2 < kt_full); }
      __syncthreads();
    }
  }
#undef TTPIPE
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  float* stg = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {
    f32x16 a2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) a2[i][j] = acc[2 * ih + i][j];
    gemm_epilogue_rows(a2, stg, Cz, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64, lane, nullptr, 0,
                       accumulate, ldc, splitk == 1);
  }
}

}  // namespace

int tssep_gemm_bf16x3_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk,
                             gemm_detail::GemmCall& call) {
  // The candidates in the order they are tried; `gemm_try(call, kernel, rule)`: in automatic mode the rule decides,
  // a forced call (tssep_gemm_f32_on) tries exactly the kernel it names.  A candidate's own launcher checks what
  // the kernel REQUIRES and returns TSSEP_E_UNSUPPORTED otherwise (the next candidate is tried in automatic mode).
  // The rules are stated in quantities of the request (padding waste of a tile shape, K stages per tile, bytes of
  // the C stream) -- profiles/r4_gemm_shape_sweep.jsonl holds, per shape of a sweep over units / projs / speakers,
  // the time of every candidate next to the one the rules pick.
  hipStream_t s = (hipStream_t)call.stream;
  const GemmSwitches sw = gemm_switches();
  const bool shift = g->kperiod > 0;
  const bool two = g->precision == 2;            // weight gradients only: the dY_lo * X_hi product dropped
  if (g->b_ones_col && (!g->b_kmajor || shift || g->N < 2)) return TSSEP_E_UNSUPPORTED;
  if (two && !(g->a_kmajor && g->b_kmajor)) return TSSEP_E_UNSUPPORTED;
#define TAKEN(KID_) do { call.chosen = (KID_); if (call.dry) return TSSEP_OK; } while (0)
  if (sw.tall && !g->a_kmajor && !g->b_kmajor && splitk == 1 && g->M >= 4 * TBM) {
    const int64_t n256 = (g->N + 255) / 256 * 256;
    const bool xcol_shape = g->N > 256 && g->N % 256 == 1 && (g->K & 3) == 0;      // N = 256 q + 1: q tiles + a VALU column
    const bool pads_to_256 = g->N >= 1024 && n256 * 10 <= g->N * 11;               // < 10 % column padding
    const bool pads_to_256_any = g->N >= 256 && n256 * 10 <= g->N * 11;            // ... also one or two column tiles (projs = 256)
    const bool short_k = g->K < 448;               // a tile's life is mostly its C store below this
    // Occupancy (round 4, the 8-utterance shard of the 8-GPU configuration: M = 2024 / 8096 rows): a kernel whose tiles do
    // not fill three quarters of the CUs once leaves the chip idle -- 128 x 128 tiles on two workgroups per CU then run up
    // to 2.7 x faster (profiles/r4_gemm_shape_sweep_b8.jsonl: 168 against 70 TFLOP/s at 8096 x 320 x 2400).  The
    // persistent streaming kernel balances its own tile list: half the CUs suffice there.
    const int64_t mt256 = (g->M + 255) / 256;
    auto fills = [&](int64_t col_tiles, int64_t need) { return mt256 * col_tiles >= need; };
    // ... and a one-workgroup-per-CU kernel whose last resident round is mostly empty loses it whole: 380 tiles = 1.48
    // rounds of 256 run at 74 % (the 8-speaker pre-net, 97 152 x 256 x 512: 261 against 312 TFLOP/s on the persistent kernel)
    auto rounds_ok = [&](int64_t col_tiles) { const int64_t t = mt256 * col_tiles; return t * 5 >= (t + 255) / 256 * 256 * 4; };
    // (the persistent big tile against the persistent 256 x 128 tile, whose list is twice as fine: 90 %)
    auto rounds_ok9 = [&](int64_t col_tiles) { const int64_t t = mt256 * col_tiles; return t * 10 >= (t + 255) / 256 * 256 * 9; };
    // big-tile kernel (gemm_bf16x3_big.hip) where the 256-wide tile applies.  (K < 448: a tile's life is mostly its
    // C store there -- the streaming kernel, which hides it, measured 4.14 against 4.56 ms at K = 320, N = 2400; from
    // K = 513 up this kernel wins: 6.55 / 6.63, 3.10 / 3.42 at K = 1280, 2.75 / 3.29 at K = 2400, N = 1280; sw.big 2 =
    // regardless of K)
    // Round 4 (profiles/r4_gemm_shape_sweep.jsonl, units x projs x speakers): N = 256 / 512 (projs = 256) belong here too
    // (302 against 244 TFLOP/s at 777 216 x 256 x 1024 with the Tanh store); the folded Tanh backward with the
    // un-combining remap reads its aux operand in the unhidden epilogue -- below K = 1536 the eight-wave tiles win
    // (280 against 230 at 194 304 x 1280 x 1024; at K = 2400, the default size, this kernel leads 338 to 300)
    const bool aux_remap_short = g->act == 2 && sm.remap && g->K < 1536;
    // 192 x 320 persistent tile (gemm_bf16x3_bigp320.hip, round 4) where 320-wide column tiles compute at least 10 % fewer
    // columns than 256-wide ones (N = 320: the Tanh projections and d(input) of birnn1; N = 600: 640 against 768 columns)
    {
      const int64_t n320 = (g->N + 319) / 320 * 320, mt192 = (g->M + 191) / 192;
      const int64_t t320 = mt192 * (n320 / 320);
      if (gemm_try(call, TSSEP_GEMM_BIG_P320, sw.big_p320 && n320 * 11 <= n256 * 10 && n320 * 10 <= g->N * 11 && !g->accumulate && sm.remap <= 1 &&
                                               t320 >= 192 && t320 * 10 >= (t320 + 255) / 256 * 256 * 9)) {
        const int rc = tssep_gemm_bf16x3_bigp320_launch(g, sm, call);
        if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_BIG_P320; return rc; }
      }
    }
    // persistent big tile (gemm_bf16x3_bigp.hip, round 4): the same tile without the per-tile drain / dispatch / prologue and
    // with a four times cheaper transposition -- plain / bias / Tanh stores, also remapped (the logit layer), any K.  profiles/r4_ab_gemm_big_p.jsonl: 5.57
    // against 6.46 ms (big) at 777 216 x 2400 x 513, 3.53 against 4.15 (stream) at K = 320; the shorter fixed part of a tile
    // also pays for more column padding than the tiled kernel's 10 %: N = 600 (768 columns computed) 1.05 against 1.12 ms on
    // the streaming kernel
    const bool pads_to_256_loosely = g->N >= 256 && n256 * 100 <= g->N * 130;
    // (N = 256 q + 1 with M a multiple of 256: q tiles + the VALU column, as in the tiled kernel)
    const int64_t ct_p = xcol_shape ? (g->N - 1) / 256 : n256 / 256;
    if (gemm_try(call, TSSEP_GEMM_BIG_P, sw.big_p && (pads_to_256_loosely || (xcol_shape && g->M % 256 == 0 && !sm.remap)) && sm.remap <= 1 && !g->accumulate &&
                                         fills(ct_p, 192) && rounds_ok9(ct_p))) {
      const int rc = tssep_gemm_bf16x3_bigp_launch(g, sm, call);
      if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_BIG_P; return rc; }
    }
    if (gemm_try(call, TSSEP_GEMM_BIG, sw.big && (!short_k || sw.big == 2) && (pads_to_256_any || xcol_shape) && !aux_remap_short &&
                                       fills(xcol_shape ? (g->N - 1) / 256 : n256 / 256, 192) &&
                                       rounds_ok(xcol_shape ? (g->N - 1) / 256 : n256 / 256))) {
      const int rc = tssep_gemm_bf16x3_big_launch(g, sm, call);
      if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_BIG; return rc; }
    }
    // persistent streaming kernel (gemm_bf16x3_stream.hip): plain row-major stores; the N = 256 q + 1 shapes keep
    // the wide tile with its VALU column
    if (gemm_try(call, TSSEP_GEMM_STREAM, sw.stream && !sm.remap && !(g->N > 256 && g->N % 256 == 1) && fills((g->N + 127) / 128, 128))) {
      const int rc = tssep_gemm_bf16x3_stream_launch(g, sm, call);
      if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_STREAM; return rc; }
    }
    // 256 x 160 tile where 160-wide column tiles waste >= 10 % fewer columns than 128-wide ones (N = 320: the Tanh
    // projections and d(input) of birnn1, which the two kernels above do not take)
    {
      // (a looser rule -- 4 % fewer columns at short K, for the logit layer's N = 2052 -- looked 19 % better in the first,
      // incumbent-first sweep and measured equal to slower in the interleaved one: not adopted)
      const int64_t n160 = (g->N + 159) / 160 * 160, n128 = (g->N + BN - 1) / BN * BN;
      if (gemm_try(call, TSSEP_GEMM_NT_W160, sw.nt_w160 && n160 * 11 <= n128 * 10 && fills(n160 / 160, 192))) {
        const int rc = tssep_gemm_bf16x3_nt_w160_launch(g, sm, call);
        if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_NT_W160; return rc; }
      }
    }
    // wide (256 x 256) eight-wave tile where rounding N up to 256 wastes < 10 % of the columns
    // (not below K = 448: 145 against 207 TFLOP/s on the four-wave tile for the 8-speaker logit layer, 97 152 x 4104 x 256;
    // from K = 448 up the big-tile kernel above has taken the request unless it cannot address it)
    if (gemm_try(call, TSSEP_GEMM_TALL4, sw.wide && pads_to_256 && !short_k && !aux_remap_short && fills(n256 / 256, 192))) {
      TAKEN(TSSEP_GEMM_TALL4);
      const TileMap tm4 = make_tile_map((g->M + TBM - 1) / TBM, n256 / 256, 1);
#ifdef TSSEP_GEMM_EXP
      {
#define HK(H_) case H_: hipLaunchKernelGGL((gemm_bf16x3_tall_kernel<4, false, H_>), dim3((unsigned)tile_map_blocks(tm4)), dim3(512), 0, s, \
                         g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act, g->accumulate, sm, tm4); return tssep_launch_status();
        switch (sw.hack) { HK(1) HK(2) HK(3) HK(4) HK(6) HK(8) HK(10) HK(12) HK(14) HK(16) HK(18) HK(32) HK(64) default: break; }
#undef HK
      }
#endif
      hipLaunchKernelGGL(gemm_bf16x3_tall_kernel<4>, dim3((unsigned)tile_map_blocks(tm4)), dim3(512), 0, s,
                         g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act,
                         g->accumulate, sm, tm4);
      return tssep_launch_status();
    }
    if (xcol_shape && gemm_try(call, TSSEP_GEMM_TALL4_XCOL, sw.xcol != 0 && fills((g->N - 1) / 256, 192))) {
      TAKEN(TSSEP_GEMM_TALL4_XCOL);
      const TileMap tmx = make_tile_map((g->M + TBM - 1) / TBM, (g->N - 1) / 256, 1);
      hipLaunchKernelGGL((gemm_bf16x3_tall_kernel<4, true>), dim3((unsigned)tile_map_blocks(tmx)), dim3(512), 0, s,
                         g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act,
                         g->accumulate, sm, tmx);
      return tssep_launch_status();
    }
    if (gemm_try(call, TSSEP_GEMM_TALL2, fills((g->N + BN - 1) / BN, 192))) {
      TAKEN(TSSEP_GEMM_TALL2);
      const TileMap tm2 = make_tile_map((g->M + TBM - 1) / TBM, (g->N + BN - 1) / BN, 1);
      hipLaunchKernelGGL(gemm_bf16x3_tall_kernel<2>, dim3((unsigned)tile_map_blocks(tm2)), dim3(NTHREADS), 0, s,
                         g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act,
                         g->accumulate, sm, tm2);
      return tssep_launch_status();
    }
  }
  const TileMap tmap = make_tile_map((g->M + BM - 1) / BM, (g->N + BN - 1) / BN, splitk);
  dim3 grid((unsigned)tile_map_blocks(tmap));
  {
    // weight gradients: both operands k-major, plain store, 16-byte rows (see gemm_bf16x3_tn_kernel)
    const int64_t nreal = g->N - (g->b_ones_col ? 1 : 0);
    const int64_t ks = g->b_kshift < 0 ? -g->b_kshift : g->b_kshift;
    if (sw.tn && g->a_kmajor && g->b_kmajor && !sm.remap && !g->bias && g->act == 0 && (g->lda & 3) == 0 &&
        (g->ldb & 3) == 0 && aligned16(g->A) && aligned16(g->B) && ((g->M + 3) & ~(int64_t)3) <= g->lda &&
        nreal >= 1 && ((nreal + 3) & ~(int64_t)3) <= g->ldb && g->M >= 4 && (!shift || (ks <= 32 && ks < g->K))) {
      // Occupancy of a weight gradient: its tiles times the splits its K allows (>= 8 K tiles of 16 rows per split, <= 64
      // splits) must fill three quarters of the CUs, else the next smaller tile is tried (the 8-utterance shard: K = 2024
      // rows -> 15 splits; the 256 x 160 tile of dW_hh then has 150 workgroups at most: 54 against 89 TFLOP/s on 128 x 128)
      const int64_t max_splits = std::min<int64_t>(64, std::max<int64_t>(1, ((g->K + 15) / 16) / 8));
      auto tn_fills = [&](int64_t tiles) { return tiles * max_splits >= 192; };
      {   // 256 x 320 workgroups of gemm_bf16x3_tn_w160.hip (round 5) for the dW_ih GEMMs whose input width is a multiple of
          // 320 (+ the ones column): birnn1 (N = 321) 3.14 against 3.79 ms on the 192 x 320 tile, birnn2 (N = 1281) 3.00 against
          // 3.61 ms on the 512 x 128 tile (tools/exp_wgrad_w320.py) -- 31 % / 28 % fewer staged bytes per MFMA; M pads to 256
          // by at most 8 % (the logit layer's M = 2052 stays on the 192-row tile)
        const int64_t nr = g->N - (g->b_ones_col ? 1 : 0);
        const int64_t m256w = (g->M + 255) / 256 * 256;
        const int wide = tn_w160_wide(g);
        // (256 x 256 workgroups, round 5: dW_ih of birnn0, N = 513 + 1 -- two column tiles + two VALU columns, 20 % fewer staged
        // bytes per MFMA than the 512 x 128 tile)
        const int64_t ncols = wide ? tn_w160_wide_cols(g, wide) : 0;
        const bool takes = wide == 4 || (wide == 5 && (ncols + 319) / 320 * 320 <= (g->N + 127) / 128 * 128);
        const int64_t tilesw = (m256w / 256) * ((ncols + (wide == 4 ? 255 : 319)) / (wide == 4 ? 256 : 320));
        // (swapped operands, round 5: the projection weight gradients -- 320 x 600 + 1 -- 0.92 against 1.34 ms on the 320 x 128 tile)
        // (from K = 81 920 rows: the three tiles of the swapped problem need ~80 splits of >= 64 K tiles each to fill the chip)
        if (gemm_try(call, TSSEP_GEMM_TN_W160, sw.tn_w160 && wide == 7 && !two && g->K >= 80 * 64 * 16)) {
          const int rc = tssep_gemm_bf16x3_tn_w160_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_W160; return rc; }
        }
        // (M pads to 256-row tiles by at most 13 %: the logit layer's 2052 -> 2304, 0.69 against 0.83 ms on the 192 x 320 tile)
        if (gemm_try(call, TSSEP_GEMM_TN_W160, sw.tn_w160 && !shift && takes && g->M >= 1024 && (m256w - g->M) * 100 <= 13 * g->M &&
                                                   tn_fills(tilesw))) {
          const int rc = tssep_gemm_bf16x3_tn_w160_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_W160; return rc; }
        }
      }
      {   // 192 x 320 tile (gemm_bf16x3_tn_p320.hip, round 4) where it computes at least 10 % less than the 512 x 128 tile:
          // N = 320 (+ the ones column) -- dW_ih of birnn1: 2496 x 320 against 2560 x 384, the logit layer's weight gradient
          // (M = 2052): 2112 x 320 against 2560 x 384
        const int64_t nr = g->N - (g->b_ones_col ? 1 : 0);
        const int64_t a320 = ((g->M + 191) / 192 * 192) * ((nr + 319) / 320 * 320);
        const int64_t rem = g->N % 128, xcn = (g->N > 128 && rem >= 1 && rem <= 2 && rem - (g->b_ones_col ? 1 : 0) <= 1) ? g->N / 128 * 128 : (g->N + 127) / 128 * 128;
        const int64_t a512 = ((g->M + 511) / 512 * 512) * xcn;
        if (gemm_try(call, TSSEP_GEMM_TN_P320, sw.tn_p320 && !shift && !two && g->M >= 768 && a320 * 100 <= a512 * 90 &&
                                                   tn_fills(((g->M + 191) / 192) * ((nr + 319) / 320)))) {
          const int rc = tssep_gemm_bf16x3_tn_p320_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_P320; return rc; }
        }
      }
      {   // big-tile weight-gradient kernel: unshifted, M padded to 512 by at most a quarter (the dW_ih GEMMs: M = 8 units; round 4:
          // the logit layer's M = speakers x 513 = 2052 / 4104 too -- 297 against 232 and 278 against 211 TFLOP/s on the tiles
          // the 10 % rule of round 3 left them, profiles/r4_gemm_shape_sweep.jsonl)
        const int64_t m512 = (g->M + 511) / 512 * 512;
        if (gemm_try(call, TSSEP_GEMM_TN_BIG, sw.tn_big && !shift && g->M >= 1024 && m512 * 4 <= g->M * 5 &&
                                                  tn_fills((m512 / 512) * ((g->N + 127) / 128)))) {
          const int rc = tssep_gemm_bf16x3_tn_big_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_BIG; return rc; }
        }
      }
      {   // 256 x 160 tile where 160-wide column tiles waste >= 10 % fewer columns than 128-wide ones (dW_hh: N = units = 300)
        const int64_t n160 = (g->N + 159) / 160 * 160, n128 = (g->N + BN - 1) / BN * BN;
        if (gemm_try(call, TSSEP_GEMM_TN_W160, sw.tn_w160 && n160 * 11 <= n128 * 10 && tn_fills(((g->M + 255) / 256) * (n160 / 160)))) {
          const int rc = tssep_gemm_bf16x3_tn_w160_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_W160; return rc; }
        }
      }
      {   // 320 x 128 tile where 320-row tiles waste >= 10 % fewer rows than 128-row ones (the projection weight
          // gradients: M = projs = 320, N = 2 units + 1: 1.18 vs 1.38 ms at each kernel's best split count), four column
          // tiles or more (one column tile: 0.13 vs 0.08 ms, profiles/r3_wgrad_h160_sweep.jsonl)
        const int64_t m320 = (g->M + 319) / 320 * 320, m128 = (g->M + BM - 1) / BM * BM;
        if (gemm_try(call, TSSEP_GEMM_TN_H160, sw.tn_h160 && !shift && m320 * 11 <= m128 * 10 && g->N > 3 * BN &&
                                                   tn_fills((m320 / 320) * ((g->N + 127) / 128)))) {
          const int rc = tssep_gemm_bf16x3_tn_h160_launch(g, sm, splitk, two ? 1 : 0, call);
          if (rc != TSSEP_E_UNSUPPORTED) { call.chosen = TSSEP_GEMM_TN_H160; return rc; }
        }
      }
      const int64_t m256 = (g->M + TTM - 1) / TTM * TTM;
      // rule 4: the time-shifted dW_hh GEMMs (-2.3 ms per step, alternating A/B) and, round 3, the unshifted ones
      // with at most 3 or at least 9 column tiles (dW_ih of birnn1: N = 321, birnn2: N = 1281 -- 4.89 vs 5.27 ms and
      // 4.28 vs 4.55 ms with the split counts tssep_gemm_wgrad_splits gives them, profiles/r3_wgrad_tile_sweep.jsonl);
      // the 5-column-tile shapes (N = 514 / 554) stay on the 128 x 128 tile: there the larger tile measured equal or
      // slower at every split count.  (1: all eligible, 2: shifted only, 3: unshifted only, 0: off)
      const int tmode = sw.tn_tall;
      const int64_t ntl = (g->N + BN - 1) / BN;
      const bool want = tmode == 1 || (tmode == 2 && shift) || (tmode == 3 && !shift) ||
                        (tmode == 4 && (shift || ntl <= 3 || ntl >= 9));
      if (g->M >= 1024 && (m256 - g->M) * 100 <= 8 * g->M && (!shift || ks <= 16) &&
          gemm_try(call, TSSEP_GEMM_TN_TALL, want && tn_fills((m256 / TTM) * ntl))) {
        TAKEN(TSSEP_GEMM_TN_TALL);
        const TileMap tmt = make_tile_map(m256 / TTM, (g->N + BN - 1) / BN, splitk);
        dim3 gridt((unsigned)tile_map_blocks(tmt));
#define TT_LAUNCH(SH, TW, KS, KP, ONES) hipLaunchKernelGGL((gemm_bf16x3_tn_tall_kernel<SH, TW>), gridt, dim3(NTHREADS), 0, s, \
            g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, KS, KP, g->accumulate, sm.ldc, splitk, g->c_split_stride, tmt, ONES)
        if (shift) { if (two) TT_LAUNCH(true, true, (int)g->b_kshift, (int)g->kperiod, 0); else TT_LAUNCH(true, false, (int)g->b_kshift, (int)g->kperiod, 0); }
        else { if (two) TT_LAUNCH(false, true, 0, 1, g->b_ones_col); else TT_LAUNCH(false, false, 0, 1, g->b_ones_col); }
#undef TT_LAUNCH
        return tssep_launch_status();
      }
      if (gemm_try(call, TSSEP_GEMM_TN, true)) {
        TAKEN(TSSEP_GEMM_TN);
#define TN_LAUNCH(SH, TW, ...) hipLaunchKernelGGL((gemm_bf16x3_tn_kernel<SH, TW>), grid, dim3(NTHREADS), 0, s, __VA_ARGS__)
        if (shift) {
          if (two) TN_LAUNCH(true, true, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, (int)g->b_kshift,
                             (int)g->kperiod, g->accumulate, sm.ldc, splitk, g->c_split_stride, tmap, 0);
          else TN_LAUNCH(true, false, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, (int)g->b_kshift,
                         (int)g->kperiod, g->accumulate, sm.ldc, splitk, g->c_split_stride, tmap, 0);
        } else {
          if (two) TN_LAUNCH(false, true, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, 0, 1, g->accumulate,
                             sm.ldc, splitk, g->c_split_stride, tmap, g->b_ones_col);
          else TN_LAUNCH(false, false, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, 0, 1, g->accumulate,
                         sm.ldc, splitk, g->c_split_stride, tmap, g->b_ones_col);
        }
#undef TN_LAUNCH
        return tssep_launch_status();
      }
    }
  }
  if (two || !gemm_try(call, TSSEP_GEMM_PIPE, true)) return TSSEP_E_UNSUPPORTED;
  TAKEN(TSSEP_GEMM_PIPE);
#define LAUNCH(AK, BKM, SH)                                                                      \
  hipLaunchKernelGGL((gemm_bf16x3_pipe_kernel<32, AK, BKM, SH>), grid, dim3(NTHREADS), 0, s,     \
                     g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->b_kshift,            \
                     g->kperiod, g->bias, g->act, g->accumulate, sm, splitk, g->c_split_stride,  \
                     tmap, g->b_ones_col)
  const bool fast = shift && g->kperiod >= 32 && (g->b_kshift == 1 || g->b_kshift == -1);
  if (!g->a_kmajor && !g->b_kmajor) LAUNCH(false, false, 0);
  else if (!g->a_kmajor && shift) { if (fast) LAUNCH(false, true, 2); else LAUNCH(false, true, 1); }
  else if (!g->a_kmajor) LAUNCH(false, true, 0);
  else if (shift) { if (fast) LAUNCH(true, true, 2); else LAUNCH(true, true, 1); }
  else LAUNCH(true, true, 0);
#undef LAUNCH
#undef TAKEN
  return tssep_launch_status();
}
