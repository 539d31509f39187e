// Big-tile weight-gradient GEMM: C[M,N] (split-K partials) = A^T B with BOTH operands k-major (A = dY [K rows][M],
// B = X [K rows][N], rows = time steps): the unshifted dW_ih / projection / head weight gradients of the step
// (tssep/train/rnnp.py:88-96,161 and tssep/train/net.py:663-666, backward).
//
// Same idea as gemm_bf16x3_big.hip (the split-bf16 kernels are bound by the bytes that return to the vector
// registers per MFMA; a 128 x 128 wave tile halves them against 128 x 64 and takes a third of 64 x 64), applied to
// the transpose-read staging of gemm_bf16x3_tn_kernel:
//  * 512 (m) x 128 (n) output tile, FOUR waves (one per SIMD) stacked along m, each 128 x 128 = 4 x 4 MFMA tiles,
//    256 accumulator registers in AGPRs.  M = 2400 pads to 2560 (6.7 %) like the 256-wide tiles; N is tiled by 128;
//  * K staged 16 rows at a time: [k][m] bf16 rows (hi and lo planes) written UNTRANSPOSED, fragments by
//    ds_read_b64_tr_b16 (two reads per 8-k fragment) exactly as in the tn kernels; THREE LDS stages (45 KB each):
//    the registers of stage s + 2 are staged while stage s computes, so stage s + 1 is complete before stage s starts
//    and its first fragments (a_lo, b_hi) are requested during the last third of stage s -- no fragment-read bubble
//    behind the barrier, which with one wave per SIMD nothing else would hide;
//  * a stage is written slot by slot (generated: tools/gen/gen_tn_big_body.py): 48 MFMAs, one transpose read or one
//    third of a staged piece (split pair / split pair / write both planes + reload for three stages ahead) per slot;
//  * masks: columns beyond M read as zero through out-of-range buffer offsets (M % 4 == 0), the two B pieces of a
//    thread are masked by value (column tail, the virtual ones column that makes the bias gradient a by-product);
//    K % 16 == 0 and no time shift (the dW_hh GEMMs keep the 256 x 128 tn kernel);
//  * same k order and MFMA sequence per output element as the tn kernels.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int WM = 512, WN = 128, WBK = 16, WNT = 256;
constexpr int WPA = WM * 2 + 64;                // bytes per k row of an A plane (512 m x 2 B + 64): 1088
constexpr int WPB = WN * 2 + 64;                // B plane: 320
constexpr int WARR_A = WBK * WPA, WARR_B = WBK * WPB;          // 17 408, 5 120
constexpr int WSTAGE = 2 * WARR_A + 2 * WARR_B;                // A hi, A lo, B hi, B lo = 45 056 B
constexpr unsigned WOOR = 0x80000000u;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int PITCH>
__device__ __forceinline__ bf16x8 trf(const char* p) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// TWO (opt-in, TSSEP_WGRAD_PRODUCTS=2): the a_lo x b_hi product is dropped -- dY enters as plain bf16, X keeps hi + lo:
// 32 instead of 48 MFMAs per stage, no lo plane of A staged or read (as in the tn kernels' TWO variant).
// XC = 10 XR + XO: N = 128 q + XR + XO -- the q column tiles go through the MFMAs and the last columns (XR <= 1 real
// columns of X, then XO <= 1 virtual ones column: dW_ih of birnn0 has N = 513 + 1, of birnn2 1280 + 1) are accumulated on the VALU
// from the raw dY values the workgroups of the LAST column tile stage anyway (exact fp32 FMA chains, 32 XC per thread
// and stage) instead of a whole extra column tile for one or two columns: a fifth / an eleventh of the GEMM.
template <bool TWO, int XC>
__global__ __launch_bounds__(WNT, 1) void gemm_bf16x3_tn_big_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N,
    int64_t K, int64_t lda, int64_t ldb, int accumulate, int64_t ldc, int splitk, int64_t c_split_stride,
    TileMap tmap, int b_ones_col, int slow_first) {
  __shared__ __attribute__((aligned(16))) char lds[3 * WSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  if (XC > 0 && slow_first) {
    // the workgroups of the last column tile carry the VALU columns and run ~10 % longer: they take the FIRST ids, so
    // that a launch of more than one round of workgroups ends on the short ones (split index fastest, as in the map)
    const int tile = mt * tmap.NT + nt;
    if (tile < tmap.MT) { mt = tile; nt = tmap.NT - 1; }
    else { mt = (tile - tmap.MT) / (tmap.NT - 1); nt = (tile - tmap.MT) % (tmap.NT - 1); }
  }
  const int64_t m0 = (int64_t)mt * WM, n0 = (int64_t)nt * WN;
  const int64_t ktiles = K / WBK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int nst = kt_end > kt_begin ? (int)(kt_end - kt_begin) : 0;        // stages of this split

  // ---- loads: A thread <-> (k row tid / 128 + 2 i, columns 4 (tid % 128) ..), i < 8;
  //             B thread <-> (k row tid / 32 + 8 i, columns 4 (tid % 32) ..), i < 2
  const int krA = tid >> 7, cqA = (tid & 127) << 2;
  const int krB = tid >> 5, cqB = (tid & 31) << 2;
  const int64_t Nreal = N - (b_ones_col ? 1 : 0);
  const srd_t asrd = make_srd(A + kt_begin * WBK * lda + m0), bsrd = make_srd(B + kt_begin * WBK * ldb + n0);
  // columns beyond M (multiple of 4) / beyond the row of B: out of range = zero
  const unsigned avo = (m0 + cqA < M) ? (unsigned)((krA * lda + cqA) * 4) : WOOR;
  const int64_t Np4 = (Nreal + 3) & ~(int64_t)3;
  const unsigned bvo = (n0 + cqB < Np4) ? (unsigned)((krB * ldb + cqB) * 4) : WOOR;
  bool bm[4], bone[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bm[e] = n0 + cqB + e < Nreal;
    bone[e] = b_ones_col && n0 + cqB + e == N - 1;
  }
  f32x4 ra[8], rb[2];
  auto load_mask = [&](int st) __attribute__((always_inline)) -> unsigned { return st >= nst ? WOOR : 0u; };
  auto gload = [&](int st) __attribute__((always_inline)) {
    const unsigned tm = load_mask(st);
    const int soa = (int)((int64_t)st * WBK * lda * 4), sob = (int)((int64_t)st * WBK * ldb * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = bload4(asrd, (avo + (unsigned)(i * 2 * lda * 4)) | tm, soa);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = bload4(bsrd, (bvo + (unsigned)(i * 8 * ldb * 4)) | tm, sob);
  };
  // ---- staging: 4 consecutive columns of one k row = 8 bytes of bf16
  const int soffA = krA * WPA + (tid & 127) * 8, soffB = 2 * WARR_A + krB * WPB + (tid & 31) * 8;
  auto maskb = [&](f32x4 b, bool valid_rows) __attribute__((always_inline)) -> f32x4 {
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = bm[e] ? b[e] : ((bone[e] && valid_rows) ? 1.f : 0.f);
    return b;
  };
  auto stage_all = [&](char* st, bool valid_rows) __attribute__((always_inline)) {      // prologue only
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned h0, l0, h1, l1;
      split2n(ra[i][0], ra[i][1], h0, l0);
      split2n(ra[i][2], ra[i][3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffA + i * 2 * WPA) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + WARR_A + soffA + i * 2 * WPA) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 b = maskb(rb[i], valid_rows);
      unsigned h0, l0, h1, l1;
      split2n(b[0], b[1], h0, l0);
      split2n(b[2], b[3], h1, l1);
      *reinterpret_cast<u32x2*>(st + soffB + i * 8 * WPB) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(st + WARR_B + soffB + i * 8 * WPB) = u32x2{l0, l1};
    }
  };
  // ---- fragment address of this lane: 16-lane group g2 covers 16 m, lane ii = 4 (k row) + m quad
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int fcol = (16 * g2 + 4 * (ii & 3)) * 2, frow = 8 * hk + (ii >> 2);
  const int aoff = frow * WPA + fcol + wave * 128 * 2, boff = 2 * WARR_A + frow * WPB + fcol;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // extra columns (XC > 0, last column tile only): column ce = nfull + e is real (< Nreal: read from B) or the ones column
  constexpr int XR = XC / 10, XO = XC % 10, XN = XR + XO;
  const int64_t nfull = (int64_t)tmap.NT * WN;
  float xacc[XN > 0 ? XN : 1][4];
  float rx[XR > 0 ? XR : 1][8];
#pragma unroll
  for (int e = 0; e < (XN > 0 ? XN : 1); ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) xacc[e][c] = 0.f;
  auto main_loop = [&](auto xc_tag) __attribute__((always_inline)) {
    constexpr int XCV = decltype(xc_tag)::value;      // 0 (no extra columns in this workgroup) or XC
    constexpr int XRV = XCV / 10, XOV = XCV % 10;
    srd_t xsrd[XRV > 0 ? XRV : 1];
#pragma unroll
    for (int e = 0; e < XRV; ++e) xsrd[e] = make_srd(B + kt_begin * WBK * ldb + nfull + e);
    auto xload = [&](int st) __attribute__((always_inline)) {          // x[k row][nfull + e] of stage st, this thread's 8 rows
      const unsigned tmx = load_mask(st);
      const int sox = (int)((int64_t)st * WBK * ldb * 4);
#pragma unroll
      for (int e = 0; e < XRV; ++e)
#pragma unroll
        for (int i = 0; i < 8; ++i) rx[e][i] = bload1(xsrd[e], (unsigned)((krA + 2 * i) * ldb * 4) | tmx, sox);
    };
    auto xfma = [&](int i) __attribute__((always_inline)) {             // piece i of the stage held in registers
#pragma unroll
      for (int e = 0; e < XRV; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) xacc[e][c] = fmaf(ra[i][c], rx[e][i], xacc[e][c]);
      if constexpr (XOV > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xacc[XRV][c] += ra[i][c];
      }
    };
    // ---- prologue: stages 0 and 1 -> LDS, stage 2 -> registers, first fragments of stage 0
    gload(0);
    if constexpr (XCV > 0) { xload(0); for (int i = 0; i < 8; ++i) xfma(i); }
    stage_all(lds, true);
    gload(1);
    if constexpr (XCV > 0) { xload(1); for (int i = 0; i < 8; ++i) xfma(i); }
    stage_all(lds + WSTAGE, 1 < nst);
    gload(2);
    if constexpr (XCV > 0) xload(2);
    __syncthreads();
    bf16x8 al[4], bh[4], ah[4], bl[4], aln[4], bhn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!TWO) al[i] = trf<WPA>(lds + WARR_A + aoff + i * 64);
      bh[i] = trf<WPB>(lds + boff + i * 64);
    }
    int c0 = 0, c1 = 1, c2 = 2;          // ring positions of stages s, s + 1, s + 2
    for (int s = 0; s < nst; ++s) {
      const char* cur = lds + c0 * WSTAGE;
      const char* nx1 = lds + c1 * WSTAGE;
      char* nx2 = lds + c2 * WSTAGE;
      const bool rows2 = s + 2 < nst;      // stage s + 2 exists (its rows are real: the ones column reads 1)
      const unsigned tm = load_mask(s + 3);
      const int soa = (int)((int64_t)(s + 3) * WBK * lda * 4), sob = (int)((int64_t)(s + 3) * WBK * ldb * 4);
      unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
      f32x4 bmk = {0.f, 0.f, 0.f, 0.f};
      const unsigned tmx = tm;
      const int sox = sob;
#define SLOT __builtin_amdgcn_sched_barrier(0)
#define MM(x, y, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[i][j], 0, 0, 0)
#define MM1(x, y, i, j) if constexpr (!TWO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[i][j], 0, 0, 0)
#define FA(dst, i, lo) dst[i] = trf<WPA>(cur + (lo) * WARR_A + aoff + (i) * 64)
#define FB(dst, i, lo) dst[i] = trf<WPB>(cur + (lo) * WARR_B + boff + (i) * 64)
#define NA(dst, i, lo) if constexpr (!TWO || (lo) == 0) dst[i] = trf<WPA>(nx1 + (lo) * WARR_A + aoff + (i) * 64)
#define NB(dst, i, lo) dst[i] = trf<WPB>(nx1 + (lo) * WARR_B + boff + (i) * 64)
#define SA1(i) if constexpr (XCV > 0) xfma(i); split2n(ra[i][0], ra[i][1], sh0, sl0)
#define SA2(i) split2n(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) *reinterpret_cast<u32x2*>(nx2 + soffA + (i) * 2 * WPA) = u32x2{sh0, sh1};            \
               if constexpr (!TWO) *reinterpret_cast<u32x2*>(nx2 + WARR_A + soffA + (i) * 2 * WPA) = u32x2{sl0, sl1};   \
               ra[i] = bload4(asrd, (avo + (unsigned)((i) * 2 * lda * 4)) | tm, soa);                                   \
               if constexpr (XRV > 0) { _Pragma("unroll") for (int e_ = 0; e_ < XRV; ++e_)                             \
                   rx[e_][i] = bload1(xsrd[e_], (unsigned)((krA + 2 * (i)) * ldb * 4) | tmx, sox); }
#define SB1(i) bmk = maskb(rb[i], rows2); split2n(bmk[0], bmk[1], sh0, sl0)
#define SB2(i) split2n(bmk[2], bmk[3], sh1, sl1)
#define SB3(i) *reinterpret_cast<u32x2*>(nx2 + soffB + (i) * 8 * WPB) = u32x2{sh0, sh1};            \
               *reinterpret_cast<u32x2*>(nx2 + WARR_B + soffB + (i) * 8 * WPB) = u32x2{sl0, sl1};   \
               rb[i] = bload4(bsrd, (bvo + (unsigned)((i) * 8 * ldb * 4)) | tm, sob)
      // GENERATED-BODY-BEGIN (tools/gen/gen_tn_big_body.py)
    MM1(al, bh, 0, 0); FA(ah, 0, 0); SLOT;
    MM1(al, bh, 0, 1); SA1(0); SLOT;
    MM1(al, bh, 0, 2); FB(bl, 0, 1); SLOT;
    MM1(al, bh, 0, 3); SA2(0); SLOT;
    MM1(al, bh, 1, 0); FB(bl, 1, 1); SLOT;
    MM1(al, bh, 1, 1); SA3(0); SLOT;
    MM1(al, bh, 1, 2); FB(bl, 2, 1); SLOT;
    MM1(al, bh, 1, 3); SA1(1); SLOT;
    MM1(al, bh, 2, 0); FB(bl, 3, 1); SLOT;
    MM1(al, bh, 2, 1); SA2(1); SLOT;
    MM1(al, bh, 2, 2); FA(ah, 1, 0); SLOT;
    MM1(al, bh, 2, 3); SA3(1); SLOT;
    MM1(al, bh, 3, 0); FA(ah, 2, 0); SLOT;
    MM1(al, bh, 3, 1); SA1(2); SLOT;
    MM1(al, bh, 3, 2); FA(ah, 3, 0); SLOT;
    MM1(al, bh, 3, 3); SA2(2); SLOT;
    MM(ah, bl, 0, 0); SA3(2); SLOT;
    MM(ah, bl, 0, 1); SA1(3); SLOT;
    MM(ah, bl, 0, 2); SA2(3); SLOT;
    MM(ah, bl, 0, 3); SA3(3); SLOT;
    MM(ah, bl, 1, 0); SA1(4); SLOT;
    MM(ah, bl, 1, 1); SA2(4); SLOT;
    MM(ah, bl, 1, 2); SA3(4); SLOT;
    MM(ah, bl, 1, 3); SLOT;
    MM(ah, bl, 2, 0); SA1(5); SLOT;
    MM(ah, bl, 2, 1); SA2(5); SLOT;
    MM(ah, bl, 2, 2); SA3(5); SLOT;
    MM(ah, bl, 2, 3); SA1(6); SLOT;
    MM(ah, bl, 3, 0); SA2(6); SLOT;
    MM(ah, bl, 3, 1); SA3(6); SLOT;
    MM(ah, bl, 3, 2); SA1(7); SLOT;
    MM(ah, bl, 3, 3); SA2(7); SLOT;
    MM(ah, bh, 0, 0); NA(aln, 0, 1); SLOT;
    MM(ah, bh, 0, 1); SA3(7); SLOT;
    MM(ah, bh, 0, 2); NB(bhn, 0, 0); SLOT;
    MM(ah, bh, 0, 3); SB1(0); SLOT;
    MM(ah, bh, 1, 0); NB(bhn, 1, 0); SLOT;
    MM(ah, bh, 1, 1); SB2(0); SLOT;
    MM(ah, bh, 1, 2); NB(bhn, 2, 0); SLOT;
    MM(ah, bh, 1, 3); SB3(0); SLOT;
    MM(ah, bh, 2, 0); NB(bhn, 3, 0); SLOT;
    MM(ah, bh, 2, 1); SB1(1); SLOT;
    MM(ah, bh, 2, 2); NA(aln, 1, 1); SLOT;
    MM(ah, bh, 2, 3); SB2(1); SLOT;
    MM(ah, bh, 3, 0); NA(aln, 2, 1); SLOT;
    MM(ah, bh, 3, 1); SB3(1); SLOT;
    MM(ah, bh, 3, 2); NA(aln, 3, 1); SLOT;
    MM(ah, bh, 3, 3); SLOT;
    // GENERATED-BODY-END
#undef SB3
#undef SB2
#undef SB1
#undef SA3
#undef SA2
#undef SA1
#undef NB
#undef NA
#undef FB
#undef FA
#undef MM1
#undef MM
#undef SLOT
#pragma unroll
      for (int i = 0; i < 4; ++i) { if constexpr (!TWO) al[i] = aln[i]; bh[i] = bhn[i]; }
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      const int t = c0; c0 = c1; c1 = c2; c2 = t;
    }
  };
  const bool xwg = XC > 0 && nt == tmap.NT - 1;          // workgroup-uniform
  if (nst > 0) {
    if (xwg) main_loop(std::integral_constant<int, XC>{});
    else main_loop(std::integral_constant<int, 0>{});
  }
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  if (xwg) {
    // the two k-row halves of the workgroup hold partial sums of the same 512 x XC outputs
    float* xs = reinterpret_cast<float*>(lds + 4 * 64 * EPITCH * 4);
#pragma unroll
    for (int e = 0; e < XN; ++e)
#pragma unroll
      for (int c = 0; c < 4; ++c) xs[((krA * 128 + (tid & 127)) * (XN > 0 ? XN : 1) + e) * 4 + c] = xacc[e][c];
    __syncthreads();
    if (krA == 0) {
#pragma unroll
      for (int e = 0; e < XN; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t m = m0 + cqA + c;
          if (m < M) {
            const float v = xacc[e][c] + xs[((128 + (tid & 127)) * (XN > 0 ? XN : 1) + e) * 4 + c];
            float* dst = Cz + m * ldc + nfull + e;
            *dst = accumulate ? *dst + v : v;
          }
        }
    }
    __syncthreads();
  }

  // ---- epilogue: four 64 x 64 blocks per wave through a private 17-KB scratch in the (now free) stage memory
  static_assert(4 * 64 * EPITCH * 4 + 2 * 128 * 2 * 4 * 4 <= 3 * WSTAGE, "epilogue scratch must fit in the stages");
  float* stage = reinterpret_cast<float*>(lds) + wave * 64 * EPITCH;
#pragma unroll
  for (int ih = 0; ih < 2; ++ih)
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      f32x16 a2[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) a2[i][j] = acc[2 * ih + i][2 * jh + j];
      gemm_epilogue_rows(a2, stage, Cz, M, XC > 0 ? nfull : N, m0 + (int64_t)wave * 128 + ih * 64, n0 + jh * 64, lane, nullptr, 0,
                         accumulate, ldc, splitk == 1);
    }
}

}  // namespace

int tssep_gemm_bf16x3_tn_big_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two,
                                    const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (!g->a_kmajor || !g->b_kmajor || sm.remap || g->bias || g->act || g->kperiod > 0) return TSSEP_E_UNSUPPORTED;
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B) || (g->M & 3) || (g->K % WBK)) return TSSEP_E_UNSUPPORTED;
  const int64_t nreal = g->N - (g->b_ones_col ? 1 : 0);
  if (g->M > g->lda || nreal < 1 || ((nreal + 3) & ~(int64_t)3) > g->ldb || g->M < WM) return TSSEP_E_UNSUPPORTED;
  const int64_t ktiles = g->K / WBK, per = (ktiles + splitk - 1) / splitk;
  // 32-bit buffer offsets: one split's rows must stay below 2 GB
  if ((per + 4) * WBK * (g->lda > g->ldb ? g->lda : g->ldb) * 4 >= (int64_t)1 << 31) return TSSEP_E_UNSUPPORTED;
  // N = 128 q + 1 or + 2 (q >= 1): the last one / two columns on the VALU instead of a column tile of their own
  // (one real column at most; the ones column of b_ones_col is the last column)
  if (call.dry) return TSSEP_OK;
  const int rem = (int)(g->N % WN), ones = g->b_ones_col ? 1 : 0;
  int xc = 0;
  if (gemm_switches().tn_xc && g->N > WN && rem >= 1 && rem <= 2 && rem - ones <= 1)
    xc = 10 * (rem - ones) + ones;
  const TileMap tm = make_tile_map((g->M + WM - 1) / WM, xc ? g->N / WN : (g->N + WN - 1) / WN, splitk);
  const int slow_first = splitk > 1 && tm.NT > 1 && gemm_switches().hack != 7;
#define TNB_LAUNCH(TW, XC_) hipLaunchKernelGGL((gemm_bf16x3_tn_big_kernel<TW, XC_>), dim3((unsigned)tile_map_blocks(tm)), dim3(WNT), 0, \
      (hipStream_t)stream, g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->accumulate, sm.ldc, splitk, g->c_split_stride, tm, g->b_ones_col, slow_first)
#define TNB_XC(TW) do { if (xc == 11) TNB_LAUNCH(TW, 11); else if (xc == 10) TNB_LAUNCH(TW, 10); else if (xc == 1) TNB_LAUNCH(TW, 1); \
                        else TNB_LAUNCH(TW, 0); } while (0)
  if (two) TNB_XC(true); else TNB_XC(false);
#undef TNB_XC
#undef TNB_LAUNCH
  return tssep_launch_status();
}
