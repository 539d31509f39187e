// Row x row GEMM with a 256 (m) x 160 (n) output tile: C[M,N] = A W^T (+ bias, Tanh / folded Tanh backward /
// accumulate / remapped stores), A [M,K] and W [N,K] row-major, for the column counts the 128-wide tiles fit badly --
// N = 320 (the projection size of the speaker BLSTMs, tssep/train/net.py:598-611: the two Tanh projections of a step and
// d(input) of birnn1) pads to 3 x 128 = 384 (83 %) but to 2 x 160 exactly.
//  * FOUR waves stacked along m, wave tile 64 x 160 = 2 x 5 MFMA tiles (160 accumulators; 14 fragment reads per 30 MFMAs),
//    two workgroups per CU; staging, LDS layout ([row][16 k] bf16 rows, 48-byte pitch, hi and lo planes), the two-stage
//    pipeline with one barrier per K tile and the tail handling are those of gemm_bf16x3_tall_kernel<2> (gemm_bf16x3.hip);
//    a K tile of W is 160 rows = 640 four-k pieces: three per thread for the first 128 threads, two for the rest;
//  * the shared row-transposed epilogues (plain, vector remap, scalar remap), 64-column blocks: two and a half per wave;
//  * same k order and MFMA sequence per output element as the other kernels -> bit-identical results.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int UM = 256, UN = 160, UBK = 16, UNT = 256, UPITCH = 48;
constexpr int UARR_A = UM * UPITCH, UARR_B = UN * UPITCH;          // 12 288, 7 680
constexpr int USTAGE = 2 * UARR_A + 2 * UARR_B;                    // A hi, A lo, B hi, B lo = 39 936 B
static_assert(2 * 64 * EPITCH * 4 <= USTAGE, "epilogue scratch must fit in one stage");

__global__ __launch_bounds__(UNT, 2) void gemm_bf16x3_nt_w160_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N, int64_t K,
    int64_t lda, int64_t ldb, const float* __restrict__ bias, int act, int accumulate, StoreMap sm, TileMap tmap) {
  __shared__ __attribute__((aligned(16))) char lds0[USTAGE];
  __shared__ __attribute__((aligned(16))) char lds1[USTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * UM, n0 = (int64_t)nt * UN;
  const int64_t ktiles = (K + UBK - 1) / UBK, kt_full = K / UBK;
  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // loads: thread <-> (row lrow + 64 i, 4 consecutive k at (tid % 4) * 4); rows of an 8-row block are visited
  // 0,2,4,6,1,3,5,7 (the four rows one ds_write_b64 lane group stages are then 96 B apart: all banks once)
  const int kq = (tid & 3) << 2;
  const int lrow = ((tid >> 2) & ~7) | (((tid >> 2) & 3) << 1) | ((tid >> 4) & 1);
  const bool b3 = lrow + 128 < UN;                         // third B piece: rows 128 .. 159 (threads 0 .. 127)
  unsigned aoffs[4], boffs[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t r = m0 + lrow + 64 * i;
    r = r > M - 1 ? M - 1 : r;
    aoffs[i] = (unsigned)(((r - m0) * lda + kq) * 4);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int64_t r = n0 + lrow + 64 * i;
    r = r > N - 1 ? N - 1 : r;
    boffs[i] = (unsigned)(((r - n0) * ldb + kq) * 4);
  }
  const srd_t asrd = make_srd(A + m0 * lda), bsrd = make_srd(B + n0 * ldb);
  f32x4 ra[4], rb[3];
  auto gload = [&](int64_t kt, bool tail) __attribute__((always_inline)) {
    const int64_t k0 = kt * UBK, k = k0 + kq;
    // (tail) a 16-byte load that starts at or beyond K would leave the row: read the row start
    const unsigned fix = (!tail || k < K) ? 0u : (unsigned)(-(k0 + kq) * 4);
    const int so = (int)(k0 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(asrd, aoffs[i] + fix, so);
#pragma unroll
    for (int i = 0; i < 3; ++i) rb[i] = bload4(bsrd, boffs[i] + fix, so);
    if (tail) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = k + e < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i][e] = ok ? ra[i][e] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) rb[i][e] = ok ? rb[i][e] : 0.f;
      }
    }
  };
  const int soff = lrow * UPITCH + ((tid & 3) << 3);
  auto sstore = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, l0, h1, l1;
      split2n(ra[i][0], ra[i][1], h0, l0);
      split2n(ra[i][2], ra[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soff + i * 64 * UPITCH) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + UARR_A + soff + i * 64 * UPITCH) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && !b3) break;
      unsigned h0, l0, h1, l1;
      split2n(rb[i][0], rb[i][1], h0, l0);
      split2n(rb[i][2], rb[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + 2 * UARR_A + soff + i * 64 * UPITCH) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + 2 * UARR_A + UARR_B + soff + i * 64 * UPITCH) = u32x2{l0, l1};
    }
  };
  const int foff = (lane & 31) * UPITCH + (lane >> 5) * 16;
  const int aoff = (wave * 64) * UPITCH + foff, boff = 2 * UARR_A + foff;
  auto compute = [&](const char* st) __attribute__((always_inline)) {
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(st + aoff + i * 32 * UPITCH);
      al[i] = *reinterpret_cast<const bf16x8*>(st + UARR_A + aoff + i * 32 * UPITCH);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * UPITCH);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + UARR_B + boff + j * 32 * UPITCH);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
    }
  };
#define UPIPE(cur, nxt, kt_)                                                                    \
  do {                                                                                          \
    compute(cur);                                                                               \
    sstore(nxt);                                                                                \
    gload((kt_) + 2, false);                                                                    \
    __syncthreads();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)

  gload(0, 0 >= kt_full);
  sstore(lds0);
  if (1 < ktiles) gload(1, 1 >= kt_full);
  __syncthreads();
  int64_t kt = 0;
  const int64_t lim = kt_full - 3;
  for (; kt < lim; kt += 2) {
    UPIPE(lds0, lds1, kt);
    UPIPE(lds1, lds0, kt + 1);
  }
  for (int par = 0; kt < ktiles; ++kt, par ^= 1) {
    const char* cur = par ? lds1 : lds0;
    char* nxt = par ? lds0 : lds1;
    compute(cur);
    if (kt + 1 < ktiles) sstore(nxt);
    if (kt + 2 < ktiles) gload(kt + 2, kt + 2 >= kt_full);
    __syncthreads();
  }
#undef UPIPE
  // epilogue: every wave transposes 64 x 64 blocks through a private LDS scratch; the third block is half a block
  float* stage = reinterpret_cast<float*>((wave & 2) ? lds1 : lds0) + (wave & 1) * 64 * EPITCH;
  const int64_t nlim = n0 + UN < N ? n0 + UN : N;
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
    const int64_t mr = m0 + (int64_t)wave * 64, nc = n0 + jh * 64;
    if (!sm.remap) gemm_epilogue_rows(a2, stage, C, M, nlim, mr, nc, lane, bias, act, accumulate, sm.ldc, true, sm.aux, sm.ldaux);
    else if (remap_vec_ok(sm, C)) gemm_epilogue_rows_remap_vec(a2, stage, C, M, nlim, mr, nc, lane, bias, act, accumulate, sm);
    else if (remap_wide_ok(sm)) gemm_epilogue_rows_remap_wide(a2, stage, C, M, nlim, mr, nc, lane, bias, act, accumulate, sm);
    else gemm_epilogue_rows_remap(a2, stage, C, M, nlim, mr, nc, lane, bias, act, accumulate, sm);
  }
}

}  // namespace

// Returns TSSEP_E_UNSUPPORTED where the geometry does not apply (the caller falls back to the 128-wide tiles).
int tssep_gemm_bf16x3_nt_w160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (g->a_kmajor || g->b_kmajor || g->M < 1024 || g->K < 48 || (g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B))
    return TSSEP_E_UNSUPPORTED;
  // 32-bit buffer offsets inside a row tile
  if ((int64_t)UM * g->lda * 4 + g->K * 4 >= ((int64_t)1 << 31) || (int64_t)UN * g->ldb * 4 + g->K * 4 >= ((int64_t)1 << 31))
    return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + UM - 1) / UM, (g->N + UN - 1) / UN, 1);
  hipLaunchKernelGGL(gemm_bf16x3_nt_w160_kernel, dim3((unsigned)tile_map_blocks(tm)), dim3(UNT), 0, (hipStream_t)stream, g->A, g->B,
                     g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act, g->accumulate, sm, tm);
  return tssep_launch_status();
}
