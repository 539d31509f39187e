// 192 x 320 weight-gradient GEMM: C[M,N] (split-K partials) = A^T B with BOTH operands k-major (A = dY [K rows][M],
// B = X [K rows][N], rows = time steps), unshifted, for N = 320 q (+ the virtual ones column): dW_ih of birnn1 (N = 320 + 1)
// and the logit layer's weight gradient (M = 2052, N = 320) -- tssep/train/rnnp.py:88-96, tssep/train/net.py:663-666,
// backward.
//
// Why: the 512 x 128 tile of gemm_bf16x3_tn_big.hip computes 2560 x 384 for 2400 x 321 (27.6 % padding) and 2560 x 384 for
// 2052 x 320 (50 %); 128-wide column tiles do not divide 320 and no 160-wide tile has a 128-row wave tile (320 accumulator
// registers).  This kernel uses the wave tile of gemm_bf16x3_bigp320.hip: four waves (one per SIMD) as 2 x 2, each 96 x 160
// = 3 x 5 MFMA tiles (240 accumulator registers), workgroup tile 192 (m) x 320 (n): 2496 x 320 + the ones column on the
// VALU for 2400 x 321 (4 %), 2112 x 320 for 2052 x 320 (3 %).
//  * K staged 16 rows at a time, as in the tn kernels: bf16 rows [k][A's 192 columns | B's 320 columns] (hi and lo planes,
//    1 088 B per k row) written UNTRANSPOSED, MFMA fragments by ds_read_b64_tr_b16; THREE LDS stages (34 KB each): stage
//    s + 2 is staged while stage s computes, so the first fragments of stage s + 1 are requested during the last third of
//    stage s -- no fragment-read bubble behind the barrier;
//  * loads: a stage is 16 x 128 pieces of 16 bytes = 32 wave-loads; wave-loads 0-11 carry A (16 rows x 48 pieces), 12-31
//    B (16 x 80): pass p of wave w takes wave-load 4 p + w, so passes 0-2 are A and 3-7 B for every wave (one buffer
//    resource per load instruction), lane l of a wave-load takes its piece l in row-major (k, piece) order;
//  * stage body generated (tools/gen/gen_tn_p320_body.py): 45 MFMAs, one transpose read or one third of a staged piece
//    per slot;
//  * masks: columns beyond M read as zero through out-of-range buffer offsets (M % 4 == 0); B pieces are masked by
//    value (column tail; the virtual ones column is computed on the VALU, not staged): N = 320 q + 1 with b_ones_col
//    runs q column tiles + the column sums of dY -- accumulated by the A pieces' threads of the LAST column tile's
//    workgroups in the fixed order of the k rows, reduced through LDS in a fixed order (deterministic);
//  * same k order and MFMA sequence per output element as the tn kernels: bit-identical for equal split counts.
// (A time-shifted variant for dW_hh -- M = 1200 pads to 1344 here -- measured 1.96 against the 256 x 160 tile's 2.00 ms at
// 3 072 sequences and 0.52 against 0.51 at 768: not kept.)
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int QM = 192, QN = 320, QBK = 16, QNT = 256;
constexpr int QP = (QM + QN) * 2 + 64;          // bytes per k row of a plane: 512 columns x 2 B + 64 = 1 088
constexpr int QPL = QBK * QP;                   // one plane of a stage: 17 408 B
constexpr int QSTAGE = 2 * QPL;                 // hi, lo = 34 816 B
constexpr int QAP = QM / 4, QBP = QN / 4;       // 16-byte pieces per k row: 48 of A, 80 of B
constexpr unsigned QOOR = 0x80000000u;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 trq(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * QP));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ONES: N = 320 q + 1 with b_ones_col -- the MFMA tiles cover the first N - 1 columns, column N - 1 = the column sums of A
template <bool ONES>
__global__ __launch_bounds__(QNT, 1) void gemm_bf16x3_tn_p320_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t Nfull,
    int64_t K, int64_t lda, int64_t ldb, int accumulate, int64_t ldc, int splitk, int64_t c_split_stride, TileMap tmap) {
  const int64_t N = ONES ? Nfull - 1 : Nfull;       // columns of the MFMA tiles (all real)
  __shared__ __attribute__((aligned(16))) char lds[3 * QSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * QM, n0 = (int64_t)nt * QN;
  const int64_t ktiles = K / QBK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int nst = kt_end > kt_begin ? (int)(kt_end - kt_begin) : 0;        // stages of this split

  // ---- the eight pieces of a thread (see the header): global byte offset within the stage's rows, LDS byte offset in the
  // hi plane, first column (B pieces: for the value mask)
  const srd_t asrd = make_srd(A + kt_begin * QBK * lda + m0), bsrd = make_srd(B + kt_begin * QBK * ldb + n0);
  unsigned goff[8];
  int loff[8], ncol[5];
  const int64_t Np4 = (N + 3) & ~(int64_t)3;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (p < 3) {
      const int P = (p * 4 + wave) * 64 + lane, k = P / QAP, q = P - k * QAP;
      goff[p] = (m0 + 4 * q < M) ? (unsigned)((k * lda + 4 * q) * 4) : QOOR;
      loff[p] = k * QP + 8 * q;
    } else {
      const int P = ((p - 3) * 4 + wave) * 64 + lane, k = P / QBP, q = P - k * QBP;
      goff[p] = (n0 + 4 * q < Np4) ? (unsigned)((k * ldb + 4 * q) * 4) : QOOR;
      loff[p] = k * QP + (QM + 4 * q) * 2;
      ncol[p - 3] = (int)(n0 + 4 * q);
    }
  }
  f32x4 rr[8];
  auto load_mask = [&](int st) __attribute__((always_inline)) -> unsigned { return st >= nst ? QOOR : 0u; };
  auto gload = [&](int st) __attribute__((always_inline)) {
    const unsigned tm = load_mask(st);
    const int soa = (int)((int64_t)st * QBK * lda * 4), sob = (int)((int64_t)st * QBK * ldb * 4);
#pragma unroll
    for (int p = 0; p < 8; ++p) rr[p] = p < 3 ? bload4(asrd, goff[p] | tm, soa) : bload4(bsrd, goff[p] | tm, sob);
  };
  auto maskb = [&](f32x4 b, int pb) __attribute__((always_inline)) -> f32x4 {      // columns beyond N are zero
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = ncol[pb] + e < N ? b[e] : 0.f;
    return b;
  };
  // column sums of A (ONES): this thread's three A pieces, summed over the stages in order
  float xacc[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) xacc[p][c] = 0.f;
  auto stage_all = [&](char* st, bool xsum) __attribute__((always_inline)) {      // prologue only
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const f32x4 v = p < 3 ? rr[p] : maskb(rr[p], p < 3 ? 0 : p - 3);
      if (ONES && xsum && p < 3) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xacc[p < 3 ? p : 0][c] += v[c];
      }
      unsigned h0, l0, h1, l1;
      split2n(v[0], v[1], h0, l0);
      split2n(v[2], v[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + loff[p]) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + QPL + loff[p]) = u32x2{l0, l1};
    }
  };
  // ---- fragment address of this lane (gemm_bf16x3_tn_big.hip): 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * QP + fcol + wm * 96 * 2, boff = frow * QP + fcol + (QM + wn * 160) * 2;

  f32x16 acc[3][5];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto main_loop = [&](auto x_tag) __attribute__((always_inline)) {
    constexpr bool XS = decltype(x_tag)::value;      // this workgroup accumulates the column sums
    // ---- prologue: stages 0 and 1 -> LDS, stage 2 -> registers, first fragments of stage 0
    gload(0);
    stage_all(lds, XS);
    gload(1);
    stage_all(lds + QSTAGE, XS);
    gload(2);
    __syncthreads();
    bf16x8 al[3], bh[5], ah[3], bl[5], aln[3], bhn[5];
#pragma unroll
    for (int i = 0; i < 3; ++i) al[i] = trq(lds + QPL + aoff + i * 64);
#pragma unroll
    for (int j = 0; j < 5; ++j) bh[j] = trq(lds + boff + j * 64);
    int c0 = 0, c1 = 1, c2 = 2;          // ring positions of stages s, s + 1, s + 2
    for (int s = 0; s < nst; ++s) {
      const char* cur = lds + c0 * QSTAGE;
      const char* nx1 = lds + c1 * QSTAGE;
      char* nx2 = lds + c2 * QSTAGE;
      const unsigned tm = load_mask(s + 3);
      const int soa = (int)((int64_t)(s + 3) * QBK * lda * 4), sob = (int)((int64_t)(s + 3) * QBK * ldb * 4);
      unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
      f32x4 pv = {0.f, 0.f, 0.f, 0.f};
#define SLOT __builtin_amdgcn_sched_barrier(0)
#define MM(x, y, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[i][j], 0, 0, 0)
#define FA(dst, i, lo) dst[i] = trq(cur + (lo) * QPL + aoff + (i) * 64)
#define FB(dst, j, lo) dst[j] = trq(cur + (lo) * QPL + boff + (j) * 64)
#define NA(dst, i, lo) dst[i] = trq(nx1 + (lo) * QPL + aoff + (i) * 64)
#define NB(dst, j, lo) dst[j] = trq(nx1 + (lo) * QPL + boff + (j) * 64)
      // a staged piece (0-2: A, 3-7: B) in three slots: mask + column sums + split the first pair | split the second
      // pair | write both planes + reload for three stages ahead
#define S1(p) { if constexpr ((p) < 3) { pv = rr[p]; if constexpr (XS) { _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) xacc[(p) < 3 ? (p) : 0][c_] += pv[c_]; } } \
                else pv = maskb(rr[p], (p) < 3 ? 0 : (p) - 3); \
                split2n(pv[0], pv[1], sh0, sl0); }
#define S2(p) split2n(pv[2], pv[3], sh1, sl1)
#define S3(p) { *reinterpret_cast<u32x2*>(nx2 + loff[p]) = u32x2{sh0, sh1};            \
                *reinterpret_cast<u32x2*>(nx2 + QPL + loff[p]) = u32x2{sl0, sl1};      \
                rr[p] = (p) < 3 ? bload4(asrd, goff[p] | tm, soa) : bload4(bsrd, goff[p] | tm, sob); }
#include "gemm_bf16x3_tn_p320_body.inc"
#undef S3
#undef S2
#undef S1
#undef NB
#undef NA
#undef FB
#undef FA
#undef MM
#undef SLOT
#pragma unroll
      for (int i = 0; i < 3; ++i) al[i] = aln[i];
#pragma unroll
      for (int j = 0; j < 5; ++j) bh[j] = bhn[j];
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      const int t = c0; c0 = c1; c1 = c2; c2 = t;
    }
  };
  const bool xwg = ONES && nt == tmap.NT - 1;          // workgroup-uniform
  if (nst > 0) {
    if (xwg) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  }
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  if (xwg) {
    // the 16 k rows of a stage hold partial sums of the same 192 column sums: xs[k row][192 m], summed in row order
    float* xs = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int P = (p * 4 + wave) * 64 + lane, k = P / QAP, q = P - k * QAP;
#pragma unroll
      for (int c = 0; c < 4; ++c) xs[k * QM + 4 * q + c] = xacc[p][c];
    }
    __syncthreads();
    if (tid < QM && m0 + tid < M) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < QBK; ++k) v += xs[k * QM + tid];
      float* dst = Cz + (m0 + tid) * ldc + N;
      *dst = accumulate ? *dst + v : v;
    }
  }
  // ---- epilogue: 4-byte stores, 32 lanes x 4 B = 128 contiguous bytes per row (split-K partials: small)
  const int64_t mrow0 = m0 + wm * 96, ncol0 = n0 + wn * 160;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t m = mrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int64_t n = ncol0 + j * 32 + (lane & 31);
        if (m < M && n < N) {
          float* dst = Cz + m * ldc + n;
          *dst = accumulate ? *dst + acc[i][j][e] : acc[i][j][e];
        }
      }
    }
}

}  // namespace

int tssep_gemm_bf16x3_tn_p320_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two,
                                     const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (two || !g->a_kmajor || !g->b_kmajor || sm.remap || g->bias || g->act || g->kperiod > 0 || g->precision == 2) return TSSEP_E_UNSUPPORTED;
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B) || (g->M & 3) || (g->K % QBK)) return TSSEP_E_UNSUPPORTED;
  const int ones = g->b_ones_col ? 1 : 0;
  const int64_t nreal = g->N - ones;
  if (g->M > g->lda || nreal < 1 || ((nreal + 3) & ~(int64_t)3) > g->ldb || g->M < 4 * QM) return TSSEP_E_UNSUPPORTED;
  const int64_t ktiles = g->K / QBK, per = (ktiles + splitk - 1) / splitk;
  if (per < 3) return TSSEP_E_UNSUPPORTED;
  // 32-bit buffer offsets: one split's rows must stay below 2 GB
  if ((per + 4) * QBK * (g->lda > g->ldb ? g->lda : g->ldb) * 4 >= (int64_t)1 << 31) return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + QM - 1) / QM, (nreal + QN - 1) / QN, splitk);
#define QLAUNCH(O_) hipLaunchKernelGGL((gemm_bf16x3_tn_p320_kernel<O_>), dim3((unsigned)tile_map_blocks(tm)), dim3(QNT), 0, \
      (hipStream_t)stream, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->accumulate, sm.ldc, splitk, g->c_split_stride, tm)
  if (ones) QLAUNCH(true); else QLAUNCH(false);
#undef QLAUNCH
  return tssep_launch_status();
}
