// Weight-gradient GEMMs on 64 x 160 / 64 x 128 wave tiles: C[M,N] (split-K partials) = A^T B, both operands k-major
// (A = d(gates) [K rows][M], B = h or X [K rows][N]) -- the time-shifted dW_hh GEMMs of every BLSTM layer and, since round
// 5, the dW_ih GEMMs (tssep/train/rnnp.py:88-96, backward: M = 4 H = 1200 gate columns of one direction against N = H = 300
// hidden units, rows = time steps paired with their neighbour t -/+ 1; M = 8 H = 2400 against the layer's input width).
// TWO kernels under one dispatcher id (tn_w160):
//  (1) gemm_bf16x3_tn_w160_kernel (round 3): 256 (m) x 160 (n) tile, FOUR waves stacked along m, two workgroups per CU.
//      N = 300 pads to 3 x 128 = 384 (78 %) but to 2 x 160 = 320 (94 %).  Staging, transpose reads, time shift (phase of
//      each staged B row inside its sequence) and the two-stage pipeline are those of gemm_bf16x3_tn_tall_kernel
//      (gemm_bf16x3.hip); the B planes keep the 320-byte pitch (= 64 B mod 256 B: conflict-free transpose reads).  Round 5:
//      masks by out-of-range loads (OOB) and a slot-by-slot stage body.  Still takes what (2) cannot: an odd number of
//      160-column tiles, M or N not multiples of 4, a ones column on N != 320 q + 1.
//  (2) gemm_bf16x3_tn_w8_kernel (round 5): EIGHT waves as 4 (m) x 2 (n), one workgroup per CU, workgroup tile 256 x 320
//      (JW = 5) or 256 x 256 (JW = 4): the same wave tile, but the staged d(gates) rows are shared by twice the columns --
//      dW_hh 2.05 -> 1.70 ms per launch, dW_ih of birnn0 / 1 / 2 6.3 -> 5.5, 3.8 -> 3.2, 3.6 -> 3.1 ms (they ran on the
//      512 x 128 and 192 x 320 tiles).  See the comment in front of it.
//  Same k order and MFMA sequence per output element as the other tn kernels -> bit-identical results for equal split
//  counts.  -DTNW160_PROBE=1..5: timing probes (wrong results) behind profiles/r5_gemm_probes.jsonl; -DTNW160_SLOTTED=0 /
//  -DTNW160_WIDE=0: the previous stage body / no eight-wave kernel, for A/B builds.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int VM = 256, VN = 160, VBK = 16, VNT = 256;
constexpr int VPA = VM * 2 + 64;                // 576 bytes per k row of an A plane
constexpr int VPB = VN * 2;                     // 320
constexpr int VARR_A = VBK * VPA, VARR_B = VBK * VPB;
constexpr int VSTAGE = 2 * 64 * EPITCH * 4;     // 34 816 B: the planes need 28 672, the epilogue two 64 x 64 scratches
static_assert(2 * VARR_A + 2 * VARR_B <= VSTAGE, "planes must fit in a stage");

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int PITCH>
__device__ __forceinline__ bf16x8 trv(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// OOB (round 5): every mask of the steady state is a property of a whole four-column piece (M and N multiples of 4, no
// ones column) or of a whole k row (the time shift's sequence boundary), so it is applied by the LOAD -- bit 31 of the
// buffer offset puts the piece out of range and the hardware returns zeros -- instead of 56 v_cndmask + 18 compares per
// 16-k stage on the VALU (249 -> 160 VALU instructions per two stages).  The time-shifted GEMMs always run this way.
constexpr unsigned VOOR = 0x80000000u;
#ifndef TNW160_PROBE
#define TNW160_PROBE 0
#endif
#ifndef TNW160_SLOTTED
#define TNW160_SLOTTED 1
#endif
#ifndef TNW160_WIDE
#define TNW160_WIDE 1
#endif
template <bool SHIFT, bool TWO, bool OOB>
__global__ __launch_bounds__(VNT, 2) void gemm_bf16x3_tn_w160_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N,
    int64_t K, int64_t lda, int64_t ldb, int kshift, int kperiod, int accumulate, int64_t ldc, int splitk,
    int64_t c_split_stride, TileMap tmap, int b_ones_col) {
  constexpr int BK = VBK;
  __shared__ __attribute__((aligned(16))) char lds0[VSTAGE];      // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) char lds1[VSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * VM, n0 = (int64_t)nt * VN;
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;

  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: A thread <-> (k row tid/64 + 4 i, columns 4 (tid%64) .. +3), i < 4;
  //        B piece p = tid + 256 i (i < 3, p < 640) <-> (k row p / 40, columns 4 (p % 40) .. +3)
  const int krA = tid >> 6, cqA = (tid & 63) << 2;
  int krB[3], cqB[3];
  bool pv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = tid + 256 * i;
    pv[i] = p < 640;
    krB[i] = pv[i] ? p / 40 : 0;
    cqB[i] = pv[i] ? (p % 40) << 2 : 0;
  }
  const bool ones = !SHIFT && b_ones_col;         // (the fused ones column only exists for unshifted GEMMs)
  const int64_t Nreal = N - (ones ? 1 : 0);
  const int64_t Mp = (M + 3) & ~(int64_t)3, Np = (Nreal + 3) & ~(int64_t)3;
  const int64_t ca = (OOB || m0 + cqA <= Mp - 4) ? m0 + cqA : Mp - 4;
  // buffer loads relative to the first row of this split (the launcher bounds a split's bytes by 2^31): scalar base
  // and stage offset, 32-bit lane offsets
  const int64_t k_begin = kt_begin * BK;
  const srd_t asrd = make_srd(A + k_begin * lda);
  const srd_t bsrd = make_srd(B + (k_begin + (SHIFT ? kshift : 0)) * ldb);
  const unsigned avo = (unsigned)((krA * lda + ca) * 4) | ((OOB && m0 + cqA >= M) ? VOOR : 0u);
  unsigned bvo[3], bm[3], bone[3];           // bit e of bm / bone: column e of the piece is a real column / the ones column
  unsigned am = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) am |= (m0 + cqA + e < M ? 1u : 0u) << e;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int64_t cb = (OOB || n0 + cqB[i] <= Np - 4) ? n0 + cqB[i] : Np - 4;
    bvo[i] = (unsigned)((krB[i] * ldb + cb) * 4) | ((OOB && n0 + cqB[i] >= Nreal) ? VOOR : 0u);
    bm[i] = 0; bone[i] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bm[i] |= (n0 + cqB[i] + e < Nreal ? 1u : 0u) << e;
      bone[i] |= ((ones && n0 + cqB[i] + e == N - 1) ? 1u : 0u) << e;
    }
  }
  int ph[3] = {0, 0, 0};                   // phase (k mod kperiod) of this thread's B rows of the tile in registers
  const int phstep = SHIFT ? BK % kperiod : 0;
  if (SHIFT) {
#pragma unroll
    for (int i = 0; i < 3; ++i) ph[i] = (int)((k_begin + krB[i]) % kperiod);
  }
  bool kokA[4] = {true, true, true, true}, kokB[3] = {true, true, true};    // row < K, of the tile held in registers

  f32x4 ra[4], rb[3];
  // OOB: `ph` is the phase of the tile being LOADED (note_tile runs in front of the load), rows outside their sequence
  // are put out of range
  auto phase_oob = [&](int i) __attribute__((always_inline)) -> unsigned {      // (one unsigned compare: 0 <= q < kperiod)
    if constexpr (OOB && SHIFT) return (unsigned)(ph[i] + kshift) < (unsigned)kperiod ? 0u : VOOR;
    return 0u;
  };
  auto gload_full = [&](int64_t kt) __attribute__((always_inline)) {
#if TNW160_PROBE == 1 || TNW160_PROBE == 4       // timing probe (wrong results): every stage re-reads the first rows of the split -- cache hits
    const int soa = (int)((kt & 1) * BK * lda * 4), sob = (int)((kt & 1) * BK * ldb * 4);
#else
    const int soa = (int)((kt - kt_begin) * BK * lda * 4), sob = (int)((kt - kt_begin) * BK * ldb * 4);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(asrd, avo + (unsigned)(i * 4 * lda * 4), soa);
#pragma unroll
    for (int i = 0; i < 3; ++i) rb[i] = bload4(bsrd, bvo[i] | phase_oob(i), sob);
  };
  auto gload_any = [&](int64_t kt) __attribute__((always_inline)) {       // rows clamped into the matrix
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = kt * BK + krA + 4 * i;
      ra[i] = bload4(asrd, (unsigned)((((k < K ? k : K - 1) - k_begin) * lda + ca) * 4) | (avo & VOOR), 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int64_t k = kt * BK + krB[i];
      int64_t kb = (k < K ? k : K - 1) + (SHIFT ? kshift : 0);
      kb = kb < 0 ? 0 : (kb > K - 1 ? K - 1 : kb);
      rb[i] = bload4(bsrd, (bvo[i] + (unsigned)((kb - (SHIFT ? kshift : 0) - k_begin - krB[i]) * ldb * 4)) | phase_oob(i), 0);
    }
  };
  int64_t held = kt_begin - 1;
  auto note_tile = [&](int64_t kt, bool full) __attribute__((always_inline)) {
    if (SHIFT && held >= kt_begin) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {               // ph + phstep mod kperiod: the smaller of x and x - kperiod as unsigned
        const unsigned x = (unsigned)(ph[i] + phstep), y = x - (unsigned)kperiod;
        ph[i] = (int)(x < y ? x : y);
      }
    }
    held = kt;
#pragma unroll
    for (int i = 0; i < 4; ++i) kokA[i] = full || kt * BK + krA + 4 * i < K;
#pragma unroll
    for (int i = 0; i < 3; ++i) kokB[i] = full || kt * BK + krB[i] < K;
  };
  const int soffA = krA * VPA + (tid & 63) * 8;
  int soffB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) soffB[i] = 2 * VARR_A + krB[i] * VPB + cqB[i] * 2;
  // the 640 B pieces of a stage are three per thread for waves 0-1 and two for waves 2-3: the third piece of waves 2-3 is
  // a dummy (out-of-range load where the variant masks that way, 1 KB of the stage's spare bytes as its target), so
  // that the stage body is ONE basic block the scheduler can interleave
  static_assert(2 * VARR_A + 3 * VARR_B + 1024 <= VSTAGE, "dummy piece (hi at the end of the planes, lo VARR_B behind it) must fit");
  if (!pv[2]) { soffB[2] = 2 * VARR_A + 2 * VARR_B + (tid - 128) * 8; if (OOB) bvo[2] |= VOOR; }
#if TNW160_PROBE == 3 || TNW160_PROBE == 4       // timing probes (wrong results): no split arithmetic
#define PSPLIT(x, y, h, l) do { h = __builtin_bit_cast(unsigned, x); l = __builtin_bit_cast(unsigned, y); } while (0)
#else
#define PSPLIT(x, y, h, l) split2n(x, y, h, l)
#endif
#if TNW160_PROBE == 5                             // timing probe (wrong results): the split, but no LDS writes
#define PWRITE(ptr, v) asm volatile("" :: "v"(v))
#else
#define PWRITE(ptr, v) *reinterpret_cast<u32x2*>(ptr) = v
#endif
  auto stage = [&](char* st, auto edge_tag, auto full_tag) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;       // every k row of the tile in registers is a row of the matrix
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 a = ra[i];
      if constexpr (EDGE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = ((FULL || kokA[i]) && ((am >> e) & 1)) ? a[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      PSPLIT(a[0], a[1], h0, l0);
      PSPLIT(a[2], a[3], h1, l1);
      PWRITE(st + soffA + i * 4 * VPA, (u32x2{h0, h1}));
      if (!TWO) PWRITE(st + VARR_A + soffA + i * 4 * VPA, (u32x2{l0, l1}));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x4 b = rb[i];
      if constexpr (EDGE) {
        const bool kok = FULL || kokB[i];
        bool okb = kok;
        if (SHIFT) { const int q = ph[i] + kshift; okb = okb && q >= 0 && q < kperiod; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          b[e] = (okb && ((bm[i] >> e) & 1)) ? b[e] : ((((bone[i] >> e) & 1) && kok) ? 1.f : 0.f);
      } else if constexpr (SHIFT && !OOB) {
        const int q = ph[i] + kshift;
        const bool okb = q >= 0 && q < kperiod;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = okb ? b[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      PSPLIT(b[0], b[1], h0, l0);
      PSPLIT(b[2], b[3], h1, l1);
      PWRITE(st + soffB[i], (u32x2{h0, h1}));
      PWRITE(st + VARR_B + soffB[i], (u32x2{l0, l1}));
    }
  };
  // fragment address of this lane: 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * VPA + fcol + wave * 64 * 2, boff = 2 * VARR_A + frow * VPB + fcol;
  auto compute = [&](const char* st) __attribute__((always_inline)) {
#if TNW160_PROBE == 2       // timing probe (wrong results): no fragment reads, no MFMAs -- the loads and the staging alone
    return;
#endif
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = trv<VPA>(st + aoff + i * 64);
      if (!TWO) al[i] = trv<VPA>(st + VARR_A + aoff + i * 64);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bf16x8 bh = trv<VPB>(st + boff + j * 64);
      const bf16x8 bl = trv<VPB>(st + VARR_B + boff + j * 64);
      if (!TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
    }
  };
  // ---- the steady-state stage of the OOB variants, written slot by slot (one MFMA + at most one other piece of work between
  // two scheduling barriers): B fragments one j ahead (b_hi double-buffered, b_lo re-read into its own registers once its
  // last product is issued), every staged piece as {two splits} / {two LDS writes + the reload of its registers for the
  // tile after the next}, so that the LDS writes and the loads are spread over the MFMA stream instead of following it
  // in one burst (probes of round 5: the burst of 13 ds_write_b64 per wave cost 20 % of the launch, the load latency
  // behind it 18 %).  Same products in the same order per accumulator as compute().
  auto slotted = [&](const char* cur, char* nxt, int64_t ktl) __attribute__((always_inline)) {
    note_tile(ktl, true);
    const int soa = (int)((ktl - kt_begin) * BK * lda * 4), sob = (int)((ktl - kt_begin) * BK * ldb * 4);
    bf16x8 ah[2], al[2], bhA, bhB, bl;
    unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
#define SLOT __builtin_amdgcn_sched_barrier(0)
#define MM(x, y, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i][j], 0, 0, 0)
#define MM1(x, y, i, j) if constexpr (!TWO) MM(x, y, i, j)
#define SA12(i) split2n(ra[i][0], ra[i][1], sh0, sl0); split2n(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) *reinterpret_cast<u32x2*>(nxt + soffA + (i) * 4 * VPA) = u32x2{sh0, sh1};                          \
               if constexpr (!TWO) *reinterpret_cast<u32x2*>(nxt + VARR_A + soffA + (i) * 4 * VPA) = u32x2{sl0, sl1}; \
               ra[i] = bload4(asrd, avo + (unsigned)((i) * 4 * lda * 4), soa)
#define SB12(i) split2n(rb[i][0], rb[i][1], sh0, sl0); split2n(rb[i][2], rb[i][3], sh1, sl1)
#define SB3(i) *reinterpret_cast<u32x2*>(nxt + soffB[i]) = u32x2{sh0, sh1};                                       \
               *reinterpret_cast<u32x2*>(nxt + VARR_B + soffB[i]) = u32x2{sl0, sl1};                              \
               rb[i] = bload4(bsrd, bvo[i] | phase_oob(i), sob)
#define FBH(dst, j) dst = trv<VPB>(cur + boff + (j) * 64)
#define FBL(j) bl = trv<VPB>(cur + VARR_B + boff + (j) * 64)
#define JBLOCK(j, bh_, X0, X1, X2, X3, X4, X5)                \
    MM1(al[0], bh_, 0, j); X0; SLOT;                          \
    MM1(al[1], bh_, 1, j); X1; SLOT;                          \
    MM(ah[0], bl, 0, j); X2; SLOT;                            \
    MM(ah[1], bl, 1, j); X3; SLOT;                            \
    MM(ah[0], bh_, 0, j); X4; SLOT;                           \
    MM(ah[1], bh_, 1, j); X5; SLOT
    if constexpr (!TWO) al[0] = trv<VPA>(cur + VARR_A + aoff);
    FBH(bhA, 0);
    if constexpr (!TWO) al[1] = trv<VPA>(cur + VARR_A + aoff + 64);
    ah[0] = trv<VPA>(cur + aoff);
    FBL(0);
    ah[1] = trv<VPA>(cur + aoff + 64);
    SLOT;
    JBLOCK(0, bhA, FBH(bhB, 1), SA12(0), SA3(0), SA12(1), FBL(1), SA3(1));
    JBLOCK(1, bhB, FBH(bhA, 2), SA12(2), SA3(2), SA12(3), FBL(2), SA3(3));
    JBLOCK(2, bhA, FBH(bhB, 3), SB12(0), SB3(0), SB12(1), FBL(3), SB3(1));
    JBLOCK(3, bhB, FBH(bhA, 4), SB12(2), SB3(2), (void)0, FBL(4), (void)0);
    JBLOCK(4, bhA, (void)0, (void)0, (void)0, (void)0, (void)0, (void)0);
#undef JBLOCK
#undef FBL
#undef FBH
#undef SB3
#undef SB12
#undef SA3
#undef SA12
#undef MM1
#undef MM
#undef SLOT
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };
#define VPIPE(cur, nxt, kt_, EDGE_)                                                             \
  do {                                                                                          \
    compute(cur);                                                                               \
    stage(nxt, std::integral_constant<bool, EDGE_>{}, std::true_type{});                        \
    note_tile((kt_) + 2, true);                                                                 \
    gload_full((kt_) + 2);                                                                      \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  if (kt_begin < kt_end) {
    note_tile(kt_begin, kt_begin < kt_full);        // (in front of the load: the loaders of the OOB variant mask by the phase)
    gload_any(kt_begin);
    stage(lds0, std::true_type{}, std::false_type{});
    if (kt_begin + 1 < kt_end) { note_tile(kt_begin + 1, kt_begin + 1 < kt_full); gload_any(kt_begin + 1); }
    __syncthreads();
    int64_t kt = kt_begin;
    int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;
    // the pipelined loads read row k + kshift unconditionally (|kshift| <= 16 here): stay clear of the matrix's
    // last tiles (they never see the first ones: they start at tile kt_begin + 2)
    if (SHIFT && lim > (K - 1) / BK - 4) lim = (K - 1) / BK - 4;
    const bool edge = !OOB && (SHIFT || m0 + VM > M || n0 + VN > Nreal);
    if constexpr (OOB && TNW160_SLOTTED) {
      for (; kt < lim; kt += 2) {
        slotted(lds0, lds1, kt + 2);
        slotted(lds1, lds0, kt + 3);
      }
    } else if (edge) {
      for (; kt < lim; kt += 2) {
        VPIPE(lds0, lds1, kt, true);
        VPIPE(lds1, lds0, kt + 1, true);
      }
    } else {
      for (; kt < lim; kt += 2) {
        VPIPE(lds0, lds1, kt, false);
        VPIPE(lds1, lds0, kt + 1, false);
      }
    }
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) stage(nxt, std::true_type{}, std::false_type{});
      if (kt + 2 < kt_end) { note_tile(kt + 2, kt + 2 < kt_full); gload_any(kt + 2); }
      __syncthreads();
    }
  }
#undef VPIPE
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  float* stg = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
  const int64_t nlim = n0 + VN < N ? n0 + VN : N;       // (the third 64-column block is half a block)
#pragma unroll
  for (int jh = 0; jh < 3; ++jh) {
    f32x16 a2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a2[i][0] = acc[i][2 * jh];
      if (jh < 2) a2[i][1] = acc[i][2 * jh + 1];
      else {
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[i][1][e] = 0.f;
      }
    }
    gemm_epilogue_rows(a2, stg, Cz, M, nlim, m0 + (int64_t)wave * 64, n0 + jh * 64, lane, nullptr, 0, accumulate, ldc,
                       splitk == 1);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Eight-wave variants, 512 threads, ONE workgroup per CU (two waves per SIMD as before): waves as 4 (m) x 2 (n), wave tile
// 64 x 32 JW -- JW = 5: workgroup tile 256 x 320, the two 160-column tiles of one row block in one workgroup (dW_hh; the
// dW_ih GEMMs with N = 320 q); JW = 4: 256 x 256 (dW_ih of birnn0: N = 512 + one more real column + the ones column).
// A stage stages 256 + 320 columns instead of 2 x (256 + 160): 31 % fewer bytes loaded, split and written to LDS per
// MFMA (JW = 4: 20 % fewer than the 512 x 128 tile).  Why that matters (round 5 probes on the 256 x 160 kernel at 3 072
// sequences, 1.86 ms per launch): no split arithmetic -2.7 %, loads that hit the cache -18 %, no LDS writes -21 % -- the
// kernel pays for the bytes it moves through the vector registers, not for the VALU.  Masks by out-of-range loads only
// (M and the MFMA columns multiples of 4), every stage of the steady state written slot by slot like `slotted` above;
// prologue and tail use a general loader (rows beyond K, time shift at the ends of the matrix: all out of range).  Same k
// order and products per output element: bit-identical to the other tn kernels for equal split counts.
// XC = 10 XR + XO (unshifted only, as in gemm_bf16x3_tn_big.hip): N = Nm + XR + XO -- the MFMA tiles cover the first Nm
// columns (a multiple of 4; the last column tile may be ragged: pieces beyond Nm are out of range); then XR <= 1 real
// columns of B and XO <= 1 virtual ones column (the bias gradient) are accumulated on the VALU
// from the raw fp32 A pieces every thread stages anyway (exact fp32 chains in the order of the k rows, the eight row groups
// of a workgroup reduced through LDS in a fixed order: deterministic; the workgroups of the last column tile write them).
template <int JW> struct W8 {
  static constexpr int WN = 64 * JW;                     // workgroup columns: 320 / 256
  static constexpr int PB = WN * 2 + 64;                 // bytes per k row of a B plane: 704 / 576 (= 48 / 16 dwords mod 64: conflict-free transpose reads)
  static constexpr int ARRB = VBK * PB;
  static constexpr int STAGE = 2 * VARR_A + 2 * ARRB;    // A hi, A lo, B hi, B lo: 40 960 / 36 864 B
  static constexpr int PPR = WN / 4;                     // 16-byte pieces per k row of B: 80 / 64
  static constexpr int NPB = (VBK * PPR + 511) / 512;    // B pieces per thread: 3 (waves 4-7: two and a repeat) / 2
};
constexpr int XNT = 512;

// SW (round 5, the projection weight gradients: M = 320 outputs against N = 600 + 1 inputs, which no eight-wave tile fits
// the way they are asked): the launcher SWAPS the operands -- the kernel's A is the caller's X (M' = 600 (+ 1) rows of the
// transposed result in three 256-row tiles), its B the caller's dY (N' = 320: one 320-column tile) -- and the kernel
// stores its accumulators TRANSPOSED, straight from the registers (a lane's four consecutive accumulator registers are
// four consecutive columns of one row of the caller's C: 16-byte stores, no LDS transposition).  The ones column of the
// caller (XO) is then ROW M' of the kernel's A: the A piece that holds it is set to {1, 0, 0, 0} by value when staged (rows
// inside K only), the MFMAs do the rest -- the column sums of dY exactly as the other kernels' ones column computes them.
// The three products of a k-step keep the CALLER's order per accumulator (a_hi b_lo first: the caller's dY_lo X_hi), so
// the result is bit-identical to the unswapped kernels.  1.34 -> 0.92 ms per launch at 777 216 rows (tn_h160: 320 x 128
// tiles, two four-wave workgroups per CU, matrix pipe busy 0.52).
template <bool SHIFT, bool TWO, int XC, int JW, bool SW = false>
__global__ __launch_bounds__(XNT, 1) void gemm_bf16x3_tn_w8_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t Nfull,
    int64_t K, int64_t lda, int64_t ldb, int kshift, int kperiod, int accumulate, int64_t ldc, int splitk,
    int64_t c_split_stride, TileMap tmap) {
  using G = W8<JW>;
  constexpr int XR = XC / 10, XO = XC % 10, XN_ = XR + XO;
  static_assert(!(SHIFT && XC), "extra columns exist for unshifted GEMMs only");
  static_assert(!SW || (!SHIFT && !TWO && XR == 0), "swapped operands: unshifted, three products, ones row only");
  static_assert(4 * 64 * EPITCH * 4 <= 2 * G::STAGE, "four epilogue scratches at a time");
  const int64_t N = SW ? Nfull : Nfull - XN_;       // columns of the MFMA tiles (all real)
  constexpr int BK = VBK, NPB = G::NPB, PB = G::PB, ARRB = G::ARRB;
  __shared__ __attribute__((aligned(16))) char lds[2 * G::STAGE];
  char* const lds0 = lds;
  char* const lds1 = lds + G::STAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;        // (waves w and w + 4 share a SIMD: same rows, the two column halves)
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * VM, n0 = (int64_t)nt * G::WN;
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;
  const int64_t k_begin = kt_begin * BK;

  f32x16 acc[2][JW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // pieces (four columns of one k row): A thread <-> (k row tid / 64 + 8 i, columns 4 (tid % 64) ..), i < 2;
  // B piece p = tid + 512 i <-> (k row p / PPR, columns 4 (p % PPR) ..); JW = 5: 1280 pieces, waves 0-3 have three, waves
  // 4-7 two and a repeat of their second one (same values to the same place: the stage body stays branch-free)
  const int krA = tid >> 6, cqA = (tid & 63) << 2;
  int krB[NPB], cqB[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int p = tid + 512 * i < VBK * G::PPR ? tid + 512 * i : tid + 512;
    krB[i] = p / G::PPR;
    cqB[i] = (p % G::PPR) << 2;
  }
  const srd_t asrd = make_srd(A + k_begin * lda);
  const srd_t bsrd = make_srd(B + (k_begin + (SHIFT ? kshift : 0)) * ldb);
  const srd_t xsrd = make_srd(B + k_begin * ldb + (XR ? N : 0));      // the one more real column: column N of B
  const unsigned avo = (unsigned)((krA * lda + m0 + cqA) * 4) | (m0 + cqA >= M ? VOOR : 0u);
  const unsigned xvo = (unsigned)(krA * ldb * 4);
  unsigned bvo[NPB];
  int ph[NPB];                              // phase (k mod kperiod) of this thread's B rows of the tile loaded last
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    bvo[i] = (unsigned)((krB[i] * ldb + n0 + cqB[i]) * 4) | (n0 + cqB[i] >= N ? VOOR : 0u);
    ph[i] = SHIFT ? (int)((k_begin + krB[i]) % kperiod) : 0;
  }
  const int phstep = SHIFT ? BK % kperiod : 0;
  auto advance_phase = [&]() __attribute__((always_inline)) {
    if constexpr (SHIFT) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const unsigned x = (unsigned)(ph[i] + phstep), y = x - (unsigned)kperiod;
        ph[i] = (int)(x < y ? x : y);
      }
    }
  };
  auto phase_oob = [&](int i) __attribute__((always_inline)) -> unsigned {
    if constexpr (SHIFT) return (unsigned)(ph[i] + kshift) < (unsigned)kperiod ? 0u : VOOR;
    return 0u;
  };
  f32x4 ra[2], rb[NPB];
  float rx[2] = {0.f, 0.f};                   // XR: B[k row of A piece i][N]
  float xacc[XN_ > 0 ? XN_ : 1][4];            // sums over this thread's k rows, for its four A columns: [real column][ones column]
#pragma unroll
  for (int e = 0; e < (XN_ > 0 ? XN_ : 1); ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) xacc[e][c] = 0.f;
  // SW: this thread's A pieces hold row M of the kernel's A (the caller's ones column); rowok[i]: piece i's k row lies inside K
  const bool ones_piece = SW && XO > 0 && m0 + cqA == M;
  bool rowok[2] = {true, true};
#define XSUM(i) { if constexpr (SW) { if constexpr (XO > 0) ra[i][0] = (ones_piece && rowok[i]) ? 1.f : ra[i][0]; }          \
                  else { if constexpr (XR > 0) { _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) xacc[0][c_] = fmaf(ra[i][c_], rx[i], xacc[0][c_]); } \
                         if constexpr (XO > 0) { _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) xacc[XR][c_] += ra[i][c_]; } } }
  // general loader (prologue, tail): tile kt, rows beyond K out of range; `ph` is the phase of tile kt
  auto gload_any = [&](int64_t kt) __attribute__((always_inline)) {
    const int64_t rel = (kt - kt_begin) * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned tail = kt * BK + krA + 8 * i < K ? 0u : VOOR;
      rowok[i] = tail == 0u;
      ra[i] = bload4(asrd, (avo + (unsigned)((rel + 8 * i) * lda * 4)) | tail, 0);
      if constexpr (XR > 0) rx[i] = bload1(xsrd, (xvo + (unsigned)((rel + 8 * i) * ldb * 4)) | tail, 0);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      rb[i] = bload4(bsrd, (bvo[i] + (unsigned)(rel * ldb * 4)) | (kt * BK + krB[i] < K ? 0u : VOOR) | phase_oob(i), 0);
  };
  const int soffA = krA * VPA + (tid & 63) * 8;
  int soffB[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) soffB[i] = 2 * VARR_A + krB[i] * PB + cqB[i] * 2;
  auto stage_plain = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned h0, l0, h1, l1;
      XSUM(i);
      split2n(ra[i][0], ra[i][1], h0, l0);
      split2n(ra[i][2], ra[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffA + i * 8 * VPA) = u32x2{h0, h1};
      if (!TWO) *reinterpret_cast<u32x2*>(st + VARR_A + soffA + i * 8 * VPA) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      unsigned h0, l0, h1, l1;
      split2n(rb[i][0], rb[i][1], h0, l0);
      split2n(rb[i][2], rb[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffB[i]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + ARRB + soffB[i]) = u32x2{l0, l1};
    }
  };
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * VPA + fcol + wm * 64 * 2, boff = 2 * VARR_A + frow * PB + fcol + wn * 32 * JW * 2;
  auto compute = [&](const char* st) __attribute__((always_inline)) {      // tail stages
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = trv<VPA>(st + aoff + i * 64);
      if (!TWO) al[i] = trv<VPA>(st + VARR_A + aoff + i * 64);
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const bf16x8 bh = trv<PB>(st + boff + j * 64);
      const bf16x8 bl = trv<PB>(st + ARRB + boff + j * 64);
      if (SW) {      // (the caller's order: its a_lo b_hi is this kernel's a_hi b_lo)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
      }
      if (!TWO) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
      }
      if (!SW) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
    }
  };
  // steady state: the tile in `cur` is computed, the registers' tile staged into `nxt`, tile ktl (full, inside the split)
  // loaded.  The stage's ONE barrier stands in front of its LAST column block, not behind it: by then every LDS write to
  // `nxt` and every read of `cur` has been issued, so behind the barrier the last six MFMAs cover the reads of the NEXT
  // stage's first fragments (a_lo, the first b_hi, b_lo -- into the registers the last block has finished with; only the two
  // a_hi reads remain at a stage's start).  With the barrier at the end every wave of the CU waited there for six transpose
  // reads before its first MFMA.  PAR: which of the two b_hi registers holds a stage's first fragment (JW odd: they alternate).
  bf16x8 al[2], bhA, bhB, bl;
  auto first_frags = [&](const char* st) __attribute__((always_inline)) {      // in front of the first steady stage
    if constexpr (!TWO) { al[0] = trv<VPA>(st + VARR_A + aoff); al[1] = trv<VPA>(st + VARR_A + aoff + 64); }
    bhA = trv<PB>(st + boff);
    bl = trv<PB>(st + ARRB + boff);
  };
  auto slotted = [&](auto par_tag, const char* cur, char* nxt, int64_t ktl) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;
    advance_phase();
    const int soa = (int)((ktl - kt_begin) * BK * lda * 4), sob = (int)((ktl - kt_begin) * BK * ldb * 4);
    bf16x8 ah[2];
    unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
#define SLOT __builtin_amdgcn_sched_barrier(0)
#define MM(x, y, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i][j], 0, 0, 0)
#define MM1(x, y, i, j) if constexpr (!TWO) MM(x, y, i, j)
#define SA12(i) XSUM(i); split2n(ra[i][0], ra[i][1], sh0, sl0); split2n(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) *reinterpret_cast<u32x2*>(nxt + soffA + (i) * 8 * VPA) = u32x2{sh0, sh1};                          \
               if constexpr (!TWO) *reinterpret_cast<u32x2*>(nxt + VARR_A + soffA + (i) * 8 * VPA) = u32x2{sl0, sl1}; \
               ra[i] = bload4(asrd, avo + (unsigned)((i) * 8 * lda * 4), soa);                                     \
               if constexpr (XR > 0) rx[i] = bload1(xsrd, xvo + (unsigned)((i) * 8 * ldb * 4), sob)
#define SB12(i) split2n(rb[i][0], rb[i][1], sh0, sl0); split2n(rb[i][2], rb[i][3], sh1, sl1)
#define SB3(i) *reinterpret_cast<u32x2*>(nxt + soffB[i]) = u32x2{sh0, sh1};                                       \
               *reinterpret_cast<u32x2*>(nxt + ARRB + soffB[i]) = u32x2{sl0, sl1};                                \
               rb[i] = bload4(bsrd, bvo[i] | phase_oob(i), sob)
    // X = the b_hi register of even column blocks, Y = of odd ones
#define BX (PAR ? bhB : bhA)
#define BY (PAR ? bhA : bhB)
#define FBX(j) if constexpr (PAR) bhB = trv<PB>(cur + boff + (j) * 64); else bhA = trv<PB>(cur + boff + (j) * 64)
#define FBY(j) if constexpr (PAR) bhA = trv<PB>(cur + boff + (j) * 64); else bhB = trv<PB>(cur + boff + (j) * 64)
#define FBL(j) bl = trv<PB>(cur + ARRB + boff + (j) * 64)
    // the next stage's first fragments (its b_hi goes where the next stage's PAR expects it: JW odd -> Y, JW even -> X)
#define NAL(i) if constexpr (!TWO) al[i] = trv<VPA>(nxt + VARR_A + aoff + (i) * 64)
#define NBH() if constexpr ((JW & 1) ? !PAR : PAR) bhB = trv<PB>(nxt + boff); else bhA = trv<PB>(nxt + boff)
#define NBL() bl = trv<PB>(nxt + ARRB + boff)
#define NOP_ (void)0
    // (SW: a_hi b_lo in front of a_lo b_hi -- the caller's product order; b_lo is free two slots earlier, a_lo two later)
#define JBLOCK(j, bh_, X0, X1, X2, X3, X4, X5)                                                      \
    if constexpr (SW) { MM(ah[0], bl, 0, j); } else { MM1(al[0], bh_, 0, j); } X0; SLOT;            \
    if constexpr (SW) { MM(ah[1], bl, 1, j); } else { MM1(al[1], bh_, 1, j); } X1; SLOT;            \
    if constexpr (SW) { MM1(al[0], bh_, 0, j); } else { MM(ah[0], bl, 0, j); } X2; SLOT;            \
    if constexpr (SW) { MM1(al[1], bh_, 1, j); } else { MM(ah[1], bl, 1, j); } X3; SLOT;            \
    MM(ah[0], bh_, 0, j); X4; SLOT;                                                                 \
    MM(ah[1], bh_, 1, j); X5; SLOT
#define LASTBLOCK(j, bh_)                                                                            \
    if constexpr (SW) { JBLOCK(j, bh_, NBH(), NOP_, NBL(), NOP_, NAL(0), NAL(1)); }                  \
    else { JBLOCK(j, bh_, NBH(), NOP_, NAL(0), NAL(1), NBL(), NOP_); }
#define STAGE_BARRIER __syncthreads(); SLOT
    ah[0] = trv<VPA>(cur + aoff);
    ah[1] = trv<VPA>(cur + aoff + 64);
    SLOT;
    if constexpr (JW == 5) {
      JBLOCK(0, BX, FBY(1), SA12(0), SA3(0), NOP_, FBL(1), SA12(1));
      JBLOCK(1, BY, FBX(2), SA3(1), NOP_, SB12(0), FBL(2), SB3(0));
      JBLOCK(2, BX, FBY(3), SB12(1), SB3(1), NOP_, FBL(3), SB12(2));
      JBLOCK(3, BY, FBX(4), SB3(2), NOP_, NOP_, FBL(4), NOP_);
      STAGE_BARRIER;
      LASTBLOCK(4, BX);
    } else {
      JBLOCK(0, BX, FBY(1), SA12(0), SA3(0), NOP_, FBL(1), SA12(1));
      JBLOCK(1, BY, FBX(2), SA3(1), NOP_, SB12(0), FBL(2), SB3(0));
      JBLOCK(2, BX, FBY(3), SB12(1), SB3(1), NOP_, FBL(3), NOP_);
      STAGE_BARRIER;
      LASTBLOCK(3, BY);
    }
#undef STAGE_BARRIER
#undef LASTBLOCK
#undef JBLOCK
#undef NOP_
#undef NBL
#undef NBH
#undef NAL
#undef FBL
#undef FBY
#undef FBX
#undef BY
#undef BX
#undef SB3
#undef SB12
#undef SA3
#undef SA12
#undef MM1
#undef MM
#undef SLOT
  };

  if (kt_begin < kt_end) {
    gload_any(kt_begin);
    stage_plain(lds0);
    if (kt_begin + 1 < kt_end) { advance_phase(); gload_any(kt_begin + 1); }
    __syncthreads();
    int64_t kt = kt_begin;
    const int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;      // tiles kt + 2, kt + 3 full and inside the split
    if (kt < lim) {
      first_frags(lds0);
      using P0 = std::integral_constant<int, 0>;
      using P1 = std::integral_constant<int, (JW & 1)>;
      for (; kt < lim; kt += 2) {      // (ONE loop exit: the accumulators stay where they are)
        slotted(P0{}, lds0, lds1, kt + 2);
        slotted(P1{}, lds1, lds0, kt + 3);
      }
    }
    // tail: `lds0` holds tile kt, the registers tile kt + 1 (if inside the split); every wave is behind the last steady
    // stage's barrier, i.e. done with its reads of the buffer the first tail stage writes
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) stage_plain(nxt);
      if (kt + 2 < kt_end) { advance_phase(); gload_any(kt + 2); }
      __syncthreads();
    }
  }
#undef XSUM
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  if constexpr (SW) {
    // transposed store: element (m, n) of the kernel's tile is C[n][m] of the caller; lane = column n of a 32 x 32 MFMA
    // tile, registers 4 q .. 4 q + 3 = rows m, m + 1, m + 2, m + 3
    const int64_t mstore = (M + (XO > 0 ? 1 : 0) + 3) & ~(int64_t)3;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        const int64_t n = n0 + wn * 32 * JW + j * 32 + (lane & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t m = m0 + wm * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
          if (n < N && m < mstore) {
            float* dst = Cz + n * ldc + m;
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = v;
          }
        }
      }
    return;
  }
  if constexpr (XN_ > 0 && !SW) {
    if (nt == tmap.NT - 1) {                  // workgroup-uniform
      float* xs = reinterpret_cast<float*>(lds);            // [extra column][row group = wave][256 columns of A]
#pragma unroll
      for (int e = 0; e < XN_; ++e)
        *reinterpret_cast<f32x4*>(xs + (e * 8 + wave) * 256 + (tid & 63) * 4) = f32x4{xacc[e][0], xacc[e][1], xacc[e][2], xacc[e][3]};
      __syncthreads();
      if (tid < 256 && m0 + tid < M) {
#pragma unroll
        for (int e = 0; e < XN_; ++e) {
          float v = xs[e * 8 * 256 + tid];
#pragma unroll
          for (int w = 1; w < 8; ++w) v += xs[(e * 8 + w) * 256 + tid];
          float* dst = Cz + (m0 + tid) * ldc + N + e;
          *dst = accumulate ? *dst + v : v;
        }
      }
      __syncthreads();
    }
  }
  const int64_t nend = n0 + (wn + 1) * 32 * JW;
  const int64_t nlim = nend < N ? nend : N;
  for (int round = 0; round < 2; ++round) {          // four scratches of 17 KB at a time
    if (round) __syncthreads();
    if (wn == round) {
      float* stg = reinterpret_cast<float*>(lds) + wm * 64 * EPITCH;
#pragma unroll
      for (int jh = 0; jh < (JW + 1) / 2; ++jh) {
        f32x16 a2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a2[i][0] = acc[i][2 * jh];
          if (2 * jh + 1 < JW) a2[i][1] = acc[i][2 * jh + 1];
          else {
#pragma unroll
            for (int e = 0; e < 16; ++e) a2[i][1][e] = 0.f;
          }
        }
        gemm_epilogue_rows(a2, stg, Cz, M, nlim, m0 + (int64_t)wm * 64, n0 + wn * 32 * JW + jh * 64, lane, nullptr, 0, accumulate, ldc,
                           splitk == 1);
      }
    }
  }
}

}  // namespace

// Returns TSSEP_E_UNSUPPORTED where the geometry does not apply: the caller (gemm_bf16x3.hip) has already checked the
// operand layout of the tn kernels (k-major operands, 16-byte rows, no bias / activation / remapped store).
int tssep_gemm_bf16x3_tn_w160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two,
                                     const gemm_detail::GemmCall& call) {
  void* const stream = call.stream;
  const bool shift = g->kperiod > 0;
  const int64_t ks = g->b_kshift < 0 ? -g->b_kshift : g->b_kshift;
  const int64_t m256 = (g->M + VM - 1) / VM * VM;
  const int wide = gemm_detail::tn_w160_wide(g);
  if (wide == 7) {
    // swapped operands (see the kernel): A' = X [K][N - ones], B' = dY [K][M], the result stored transposed
    if (!TNW160_WIDE || two || (sm.ldc & 3) != 0) return TSSEP_E_UNSUPPORTED;
    const int64_t nr = g->N - (g->b_ones_col ? 1 : 0);
    if (sm.ldc < ((g->N + 3) & ~(int64_t)3)) return TSSEP_E_UNSUPPORTED;
    const int64_t ktiles = (g->K + VBK - 1) / VBK, per = (ktiles + splitk - 1) / splitk;
    const int64_t ldmax = g->lda > g->ldb ? g->lda : g->ldb;
    if ((per + 4) * VBK * ldmax * 4 >= ((int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
    if (call.dry) return TSSEP_OK;
    const TileMap tms = make_tile_map((g->N + 255) / 256, (g->M + 319) / 320, splitk);
    const dim3 grids((unsigned)tile_map_blocks(tms));
#define S_LAUNCH(XC_) hipLaunchKernelGGL((gemm_bf16x3_tn_w8_kernel<false, false, XC_, 5, true>), grids, dim3(XNT), 0, (hipStream_t)stream, \
      g->B, g->A, g->C, nr, g->M, g->K, g->ldb, g->lda, 0, 1, g->accumulate, sm.ldc, splitk, g->c_split_stride, tms)
    if (g->b_ones_col) S_LAUNCH(1); else S_LAUNCH(0);
#undef S_LAUNCH
    return tssep_launch_status();
  }
  if (g->M < 1024 || (m256 - g->M) * 100 > 13 * g->M || (shift && ks > 16) || (sm.ldc & 3) != 0) return TSSEP_E_UNSUPPORTED;
  // masks by out-of-range loads: whole four-column pieces only (the time-shifted kernel exists in this form alone)
  const bool oob = (g->M & 3) == 0 && (g->N & 3) == 0 && !g->b_ones_col;
  if (shift && !oob) return TSSEP_E_UNSUPPORTED;
  // 32-bit buffer offsets inside a split
  const int64_t ktiles = (g->K + VBK - 1) / VBK, per = (ktiles + splitk - 1) / splitk;
  const int64_t ldmax = g->lda > g->ldb ? g->lda : g->ldb;
  if ((per + 4) * VBK * ldmax * 4 >= ((int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  // an even number of 160-column tiles: pairs of them in one 512-thread workgroup (256 x 320)
  const int64_t nt160 = (g->N + VN - 1) / VN;
  if (TNW160_WIDE && wide) {
    // 0 = no; 5 -> 256 x 320 workgroups (XC = 0 / 1), 4 -> 256 x 256 (XC = 0 / 1 / 10 / 11)
    const int xo = g->b_ones_col ? 1 : 0;
    const int64_t ncols = gemm_detail::tn_w160_wide_cols(g, wide);
    const int xr = (int)(g->N - xo - ncols);
    const int xc = 10 * xr + xo;
    const int wn_ = wide == 5 ? 320 : 256;
    const TileMap tmw = make_tile_map(m256 / VM, (ncols + wn_ - 1) / wn_, splitk);
    const dim3 gridw((unsigned)tile_map_blocks(tmw));
#define W_LAUNCH(SH, TW, XC_, JW_) hipLaunchKernelGGL((gemm_bf16x3_tn_w8_kernel<SH, TW, XC_, JW_>), gridw, dim3(XNT), 0, (hipStream_t)stream, \
      g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, (int)g->b_kshift, shift ? (int)g->kperiod : 1, g->accumulate, sm.ldc, \
      splitk, g->c_split_stride, tmw)
#define W_TWO(SH, XC_, JW_) do { if (two) W_LAUNCH(SH, true, XC_, JW_); else W_LAUNCH(SH, false, XC_, JW_); } while (0)
    if (wide == 5) {
      if (shift) W_TWO(true, 0, 5);
      else if (xc == 11) W_TWO(false, 11, 5);
      else if (xc == 10) W_TWO(false, 10, 5);
      else if (xc == 1) W_TWO(false, 1, 5);
      else W_TWO(false, 0, 5);
    } else {
      if (xc == 11) W_TWO(false, 11, 4);
      else if (xc == 10) W_TWO(false, 10, 4);
      else if (xc == 1) W_TWO(false, 1, 4);
      else W_TWO(false, 0, 4);
    }
#undef W_TWO
#undef W_LAUNCH
    return tssep_launch_status();
  }
  const TileMap tm = make_tile_map(m256 / VM, nt160, splitk);
  const dim3 grid((unsigned)tile_map_blocks(tm));
#define V_LAUNCH(SH, TW, OB, KS, KP, ONES) hipLaunchKernelGGL((gemm_bf16x3_tn_w160_kernel<SH, TW, OB>), grid, dim3(VNT), 0, (hipStream_t)stream, \
      g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, KS, KP, g->accumulate, sm.ldc, splitk, g->c_split_stride, tm, ONES)
  if (shift) { if (two) V_LAUNCH(true, true, true, (int)g->b_kshift, (int)g->kperiod, 0); else V_LAUNCH(true, false, true, (int)g->b_kshift, (int)g->kperiod, 0); }
  else if (oob) { if (two) V_LAUNCH(false, true, true, 0, 1, 0); else V_LAUNCH(false, false, true, 0, 1, 0); }
  else { if (two) V_LAUNCH(false, true, false, 0, 1, g->b_ones_col); else V_LAUNCH(false, false, false, 0, 1, g->b_ones_col); }
#undef V_LAUNCH
  return tssep_launch_status();
}
