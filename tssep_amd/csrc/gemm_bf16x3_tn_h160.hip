// Weight-gradient GEMM with a 320 (m) x 128 (n) output tile: C[M,N] (split-K partials) = A^T B, both operands
// k-major (A = dZ [K rows][M], B = h [K rows][N]), for M = 320 -- the weight gradients of the BLSTM projections
// (tssep/train/rnnp.py:154-161, backward: M = 320 projection outputs, N = 600 hidden units + the ones column, rows =
// time steps), which pad to 3 x 128 = 384 rows (83 %) on the 128 x 128 tile.
//  * FOUR waves as 2 x 2, wave tile 160 (m) x 64 (n) = 5 x 2 MFMA tiles (160 accumulators; 14 fragment reads per 30
//    MFMAs against 16 per 24 of the 64 x 64 wave tile), two workgroups per CU;
//  * the mirror image of gemm_bf16x3_tn_w160.hip: a k row of A is 320 columns = 80 four-column pieces, the 16 rows of a
//    stage 1280 pieces = five per thread; B two per thread; `[k][m]` bf16 rows (pitch 704 B for A = -64 B mod 256 B:
//    conflict-free transpose reads like the +64 B pitches; 320 B for B), fragments by `ds_read_b64_tr_b16`, masks at
//    staging time (row tail, column tail, the virtual ones column), two LDS stages, one barrier per K tile;
//  * no time shift (the dW_hh GEMMs have their own 256 x 160 tile); same k order and MFMA sequence per output element
//    as the other tn kernels -> bit-identical results.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int HM = 320, HN = 128, HBK = 16, HNT = 256;
constexpr int HPA = HM * 2 + 64;                // 704 bytes per k row of an A plane
constexpr int HPB = HN * 2 + 64;                // 320
constexpr int HARR_A = HBK * HPA, HARR_B = HBK * HPB;
constexpr int HSTAGE = 2 * 64 * EPITCH * 4;     // 34 816 B: the planes need 32 768, the epilogue two 64 x 64 scratches
static_assert(2 * HARR_A + 2 * HARR_B <= HSTAGE, "planes must fit in a stage");

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int PITCH>
__device__ __forceinline__ bf16x8 trh(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool TWO>
__global__ __launch_bounds__(HNT, 2) void gemm_bf16x3_tn_h160_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N,
    int64_t K, int64_t lda, int64_t ldb, int accumulate, int64_t ldc, int splitk, int64_t c_split_stride,
    TileMap tmap, int b_ones_col) {
  constexpr int BK = HBK;
  __shared__ __attribute__((aligned(16))) char lds0[HSTAGE];      // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) char lds1[HSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * HM, n0 = (int64_t)nt * HN;
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;

  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: A columns 0..255: thread <-> (k row tid/64 + 4 i, columns 4 (tid%64) .. +3), i < 4; columns 256..319:
  //        thread <-> (k row tid/16, columns 256 + 4 (tid%16) .. +3) (i = 4);
  //        B thread <-> (k row tid/32 + 8 i, columns 4 (tid%32) .. +3), i < 2
  int krA[5], cqA[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) { krA[i] = (tid >> 6) + 4 * i; cqA[i] = (tid & 63) << 2; }
  krA[4] = tid >> 4; cqA[4] = 256 + ((tid & 15) << 2);
  const int krB = tid >> 5, cqB = (tid & 31) << 2;
  const bool ones = b_ones_col != 0;
  const int64_t Nreal = N - (ones ? 1 : 0);
  const int64_t Mp = (M + 3) & ~(int64_t)3, Np = (Nreal + 3) & ~(int64_t)3;
  // buffer loads relative to the first row of this split (the launcher bounds a split's bytes by 2^31)
  const int64_t k_begin = kt_begin * BK;
  const srd_t asrd = make_srd(A + k_begin * lda);
  const srd_t bsrd = make_srd(B + k_begin * ldb);
  unsigned avo[2], am[2];                      // (i < 4: avo[0] + 4 i rows, mask am[0]; i = 4: avo[1], am[1])
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t ca = m0 + cqA[4 * q] <= Mp - 4 ? m0 + cqA[4 * q] : Mp - 4;
    avo[q] = (unsigned)((krA[4 * q] * lda + ca) * 4);
    am[q] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) am[q] |= (m0 + cqA[4 * q] + e < M ? 1u : 0u) << e;
  }
  const int64_t cb = n0 + cqB <= Np - 4 ? n0 + cqB : Np - 4;
  const unsigned bvo = (unsigned)((krB * ldb + cb) * 4);
  unsigned bm = 0, bone = 0;                  // bit e: column e of the piece is a real column / the ones column
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bm |= (n0 + cqB + e < Nreal ? 1u : 0u) << e;
    bone |= ((ones && n0 + cqB + e == N - 1) ? 1u : 0u) << e;
  }
  bool kokA[5] = {true, true, true, true, true}, kokB[2] = {true, true};    // row < K, of the tile held in registers

  f32x4 ra[5], rb[2];
  auto gload_full = [&](int64_t kt) __attribute__((always_inline)) {
    const int soa = (int)((kt - kt_begin) * BK * lda * 4), sob = (int)((kt - kt_begin) * BK * ldb * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(asrd, avo[0] + (unsigned)(i * 4 * lda * 4), soa);
    ra[4] = bload4(asrd, avo[1], soa);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = bload4(bsrd, bvo + (unsigned)(i * 8 * ldb * 4), sob);
  };
  auto gload_any = [&](int64_t kt) __attribute__((always_inline)) {       // rows clamped into the matrix
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int64_t k = kt * BK + krA[i];
      ra[i] = bload4(asrd, avo[i >> 2] + (unsigned)(((k < K ? k : K - 1) - k_begin - krA[4 * (i >> 2)]) * lda * 4), 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t k = kt * BK + krB + 8 * i;
      rb[i] = bload4(bsrd, bvo + (unsigned)(((k < K ? k : K - 1) - k_begin - krB) * ldb * 4), 0);
    }
  };
  auto note_tile = [&](int64_t kt, bool full) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i) kokA[i] = full || kt * BK + krA[i] < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) kokB[i] = full || kt * BK + krB + 8 * i < K;
  };
  const int soffA0 = krA[0] * HPA + cqA[0] * 2, soffA4 = krA[4] * HPA + cqA[4] * 2;
  const int soffB = 2 * HARR_A + krB * HPB + (tid & 31) * 8;
  auto stage = [&](char* st, auto edge_tag, auto full_tag) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;       // every k row of the tile in registers is a row of the matrix
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      f32x4 a = ra[i];
      if constexpr (EDGE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = ((FULL || kokA[i]) && ((am[i >> 2] >> e) & 1)) ? a[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split2n(a[0], a[1], h0, l0);
      split2n(a[2], a[3], h1, l1);
      const int so = i < 4 ? soffA0 + i * 4 * HPA : soffA4;
      *reinterpret_cast<u32x2*>(st + so) = u32x2{h0, h1};
      if (!TWO) *reinterpret_cast<u32x2*>(st + HARR_A + so) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 b = rb[i];
      if constexpr (EDGE) {
        const bool kok = FULL || kokB[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (kok && ((bm >> e) & 1)) ? b[e] : ((((bone >> e) & 1) && kok) ? 1.f : 0.f);
      }
      unsigned h0, l0, h1, l1;
      split2n(b[0], b[1], h0, l0);
      split2n(b[2], b[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffB + i * 8 * HPB) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + HARR_B + soffB + i * 8 * HPB) = u32x2{l0, l1};
    }
  };
  // fragment address of this lane: 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * HPA + fcol + wm * 160 * 2, boff = 2 * HARR_A + frow * HPB + fcol + wn * 64 * 2;
  auto compute = [&](const char* st) __attribute__((always_inline)) {
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = trh<HPB>(st + boff + j * 64);
      bl[j] = trh<HPB>(st + HARR_B + boff + j * 64);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const bf16x8 ah = trh<HPA>(st + aoff + i * 64);
      if (!TWO) {
        const bf16x8 al = trh<HPA>(st + HARR_A + aoff + i * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
    }
  };
#define HPIPE(cur, nxt, kt_, EDGE_)                                                             \
  do {                                                                                          \
    compute(cur);                                                                               \
    stage(nxt, std::integral_constant<bool, EDGE_>{}, std::true_type{});                        \
    gload_full((kt_) + 2);                                                                      \
    note_tile((kt_) + 2, true);                                                                 \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  if (kt_begin < kt_end) {
    gload_any(kt_begin);
    note_tile(kt_begin, kt_begin < kt_full);
    stage(lds0, std::true_type{}, std::false_type{});
    if (kt_begin + 1 < kt_end) { gload_any(kt_begin + 1); note_tile(kt_begin + 1, kt_begin + 1 < kt_full); }
    __syncthreads();
    int64_t kt = kt_begin;
    const int64_t lim = (kt_end < kt_full ? kt_end : kt_full) - 3;
    const bool edge = m0 + HM > M || n0 + HN > Nreal;
    if (edge) {
      for (; kt < lim; kt += 2) {
        HPIPE(lds0, lds1, kt, true);
        HPIPE(lds1, lds0, kt + 1, true);
      }
    } else {
      for (; kt < lim; kt += 2) {
        HPIPE(lds0, lds1, kt, false);
        HPIPE(lds1, lds0, kt + 1, false);
      }
    }
    for (int par = 0; kt < kt_end; ++kt, par ^= 1) {
      const char* cur = par ? lds1 : lds0;
      char* nxt = par ? lds0 : lds1;
      compute(cur);
      if (kt + 1 < kt_end) stage(nxt, std::true_type{}, std::false_type{});
      if (kt + 2 < kt_end) { gload_any(kt + 2); note_tile(kt + 2, kt + 2 < kt_full); }
      __syncthreads();
    }
  }
#undef HPIPE
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  float* stg = reinterpret_cast<float*>(wave < 2 ? lds0 : lds1) + (wave & 1) * 64 * EPITCH;
  // the wave's 160 rows as two and a half 64-row blocks (rows beyond its slice belong to the other row wave)
  const int64_t mlim = m0 + (int64_t)wm * 160 + 160 < M ? m0 + (int64_t)wm * 160 + 160 : M;
#pragma unroll
  for (int ih = 0; ih < 3; ++ih) {
    f32x16 a2[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a2[0][j] = acc[2 * ih][j];
      if (ih < 2) a2[1][j] = acc[2 * ih + 1][j];
      else {
#pragma unroll
        for (int e = 0; e < 16; ++e) a2[1][j][e] = 0.f;
      }
    }
    gemm_epilogue_rows(a2, stg, Cz, mlim, N, m0 + (int64_t)wm * 160 + ih * 64, n0 + (int64_t)wn * 64, lane, nullptr, 0,
                       accumulate, ldc, splitk == 1);
  }
}

}  // namespace

// Returns TSSEP_E_UNSUPPORTED where the geometry does not apply: the caller (gemm_bf16x3.hip) has already checked the
// operand layout of the tn kernels (k-major operands, 16-byte rows, no bias / activation / remapped store, no shift).
int tssep_gemm_bf16x3_tn_h160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two,
                                     const gemm_detail::GemmCall& call) {
  void* const stream = call.stream;
  using namespace gemm_detail;
  if (g->kperiod > 0 || (sm.ldc & 3) != 0) return TSSEP_E_UNSUPPORTED;
  // 32-bit buffer offsets inside a split
  const int64_t ktiles = (g->K + HBK - 1) / HBK, per = (ktiles + splitk - 1) / splitk;
  const int64_t ldmax = g->lda > g->ldb ? g->lda : g->ldb;
  if ((per + 4) * HBK * ldmax * 4 >= ((int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + HM - 1) / HM, (g->N + HN - 1) / HN, splitk);
  const dim3 grid((unsigned)tile_map_blocks(tm));
#define H_LAUNCH(TW) hipLaunchKernelGGL((gemm_bf16x3_tn_h160_kernel<TW>), grid, dim3(HNT), 0, (hipStream_t)stream, g->A, g->B, g->C, \
      g->M, g->N, g->K, g->lda, g->ldb, g->accumulate, sm.ldc, splitk, g->c_split_stride, tm, g->b_ones_col)
  if (two) H_LAUNCH(true); else H_LAUNCH(false);
#undef H_LAUNCH
  return tssep_launch_status();
}
