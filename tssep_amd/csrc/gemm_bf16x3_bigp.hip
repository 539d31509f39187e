// Persistent variant of the big-tile split-bf16 GEMM (gemm_bf16x3_big.hip) for row-major x row-major operands whose
// store is a plain row-major C: the LSTM input projections (N = 2400, K = 320 / 513 / 553 / 1280) and the other wide
// nn.Linear GEMMs (tssep/train/rnnp.py:88-96,146-161).
//
// Why: the 256 x 256 tile's life in gemm_bf16x3_big.hip has a FIXED part of ~22 us next to its K stages (2.3 us per
// 32 k): fitted over K = 553 / 1280 at N = 2400, and the "no epilogue" probe of profiles/r3_gemm_big_probes.jsonl
// alone is 16 us per tile -- half of a K = 320 tile, a third of a K = 513 one.  Nothing of that is bandwidth (256 KB
// per CU and tile): it is the LDS transposition in 64 four-byte writes per 32 x 32 tile, the wait of `s_endpgm` for
// the last store's acknowledgement with the CU's registers and LDS held, the next workgroup's dispatch and its
// prologue's load latency.  This kernel removes the three:
//  * persistent grid (one workgroup per CU) walking the XCD-aware tile list in ONE software pipeline over (tile,
//    K stage): the operand loads run two stages ahead ACROSS tile boundaries, so a tile ends with the next tile's
//    stage 0 in LDS and its stage 1 in flight -- no drain, no dispatch, no prologue;
//  * the MFMA operands swapped (weights as the A operand, activations as B): a lane's four consecutive accumulator
//    registers are then four consecutive COLUMNS of one row, and the transposition through LDS is 16 `ds_write_b128`
//    + 16 `ds_read_b128` per 32 x 128 block (XOR-swizzled 16-byte chunks, conflict-free both ways) instead of 64
//    `ds_write_b32`; the same dot products in the same k order: bit-identical to the other split-bf16 kernels;
//  * stores as buffer stores (scalar base, one 32-bit lane offset, no 64-bit address arithmetic per row; soffset 0:
//    see tools/scan_store_hazard.py), two rows x 512 B per wave instruction; the bias from LDS (a global load would
//    sit behind the prefetched operand tiles in the in-order memory counter);
//  * a tile's accumulators start from the C operand 0 of its first products (no zeroing pass).
// Everything else -- 128 x 128 wave tiles, one wave per SIMD, the slot schedule of a stage -- is the big-tile kernel's
// (gemm_bf16x3_big_schedule.inc).
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

#ifndef BIGP_PROBE_NOB
#define BIGP_PROBE_NOB 0
#endif
#ifndef BIGP_PROBE_NOA      // round 6, the twin for the ACTIVATION operand: 1 = neither loaded, split nor written; 2 = eight 1-KB LDS-DMA copies
#define BIGP_PROBE_NOA 0   // per wave and stage in its place (timing probes, results wrong; profiles/r6_gemm_noa_probe.jsonl)
#endif
constexpr int GM = 256, GN = 256, GBK = 32, GNT = 256;
constexpr int GROWB = 64;                                   // bytes per LDS row: 32 bf16
constexpr int GARR = GM * GROWB;                            // 16 384 B per plane
constexpr int GSTAGE = 4 * GARR;                            // A hi, A lo, B hi, B lo = 65 536 B
constexpr unsigned GOOR = 0x80000000u;                      // buffer offset beyond the range: loads return 0, stores are dropped
constexpr int PBIAS = 4096;                                 // floats of bias kept in LDS (N beyond that: the tiled kernel)
constexpr int XBIAS = 1024, XROW = 4096;                    // XCOL: bias (N <= 1024), row N - 1 of B (K <= 4096)

// PROBE (experiment builds only, TIMING probes): 24 = no epilogue at all (garbage results), 32 = plain instead of
// non-temporal stores, 64 = sc1 stores (results stay right); profiles/r4_gemm_big_p_probes.jsonl
// Tile list of one workgroup without divisions: position (n-group ng, m-tile rm of this XCD, n-tile rn of the group) of
// id = 8 l + xcd in the map of gemm_common.h (l = (ng MTx + rm) NG + rn), advanced by `step` = gridDim.x / 8 ids of the
// same XCD at a time (dq = step / NG, dr = step % NG from the host).  Scalar registers only.
struct TileWalk { int step, dq, dr; };

// XCOL (N = 256 q + 1: the 513 frequency bins of `dgrad birnn0 dx` and of the pre-net projection), as in the tiled kernel:
// the MFMA tiles cover the first N - 1 columns and column N - 1 is computed on the VALU from the raw fp32 A values every
// thread stages anyway (exact fp32, 32 FMAs per thread and stage against row N - 1 of B, kept in LDS); every workgroup
// computes it -- branch-free -- and those of the last column tile store it.  The twelve registers this takes come from
// the load offsets: XCOL requires M % 256 == 0, so no row of a tile needs clamping and the sixteen per-piece lane
// offsets become one per operand plus a scalar row-group offset.
// REMAP: the remapped store of gemm_common.h (row m = (b K + k) T + t, column n = q cm + r -> b sb + k sk + t st +
// perm_b(q) co + r; groups of any size: the logit layer's 513 bins per speaker): a lane's four columns leave as one
// 16-byte store at a 4-byte-aligned address where they lie in one group, element by element where they straddle two; the
// permutation entries of the (at most two) utterances of a wave tile are loaded before the tile's first store.
// ONE (tssep_gemm_args.precision = 3, the plain-bf16 side line): operands rounded to bf16, ONE product a_hi b_hi per k-step --
// the lo planes are neither staged nor read, a stage has 32 MFMAs per wave.
template <int PROBE, int ACT, bool XCOL, bool REMAP, bool ONE>
__global__ __launch_bounds__(GNT, 1) void gemm_bf16x3_bigp_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t Nfull, int64_t K,
    int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, TileMap tmap, TileWalk walk, StoreMap sm) {
  const int64_t N = XCOL ? Nfull - 1 : Nfull;          // columns of the MFMA tiles
  __shared__ __attribute__((aligned(16))) char lds[2 * GSTAGE];
  // the bias, read by the store from LDS: a global load would sit behind the prefetched operand tiles in the in-order
  // memory counter
  // (XCOL: bias (N <= 1024) | row N - 1 of B, zero beyond K (K <= 4096) | the finished column sums of a tile, 8 per thread)
  __shared__ __attribute__((aligned(16))) float extra_s[XCOL ? XBIAS + XROW + 8 * GNT : PBIAS];
  float* const bias_s = extra_s;
  float* const xrow_s = extra_s + XBIAS;
  float* const xs_s = extra_s + XBIAS + XROW;
  constexpr int NBIAS = XCOL ? XBIAS : PBIAS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int KT = (int)((K + GBK - 1) / GBK);
  const bool ktail = (K % GBK) != 0;
  for (int i = tid; i < NBIAS; i += GNT) bias_s[i] = (bias && i < Nfull) ? bias[i] : 0.f;      // (visible after the prologue's barrier)
  if constexpr (XCOL) {
    for (int i = tid; i < XROW; i += GNT) xrow_s[i] = i < K ? B[N * ldb + i] : 0.f;
  }

  // ---- tile list of this workgroup (see TileWalk); `first`: the position of id = blockIdx.x itself is tried as it is
  const int xcd = (int)(blockIdx.x % NXCD);
  int w_ng, w_rm, w_rn;
  {
    const int l = (int)(blockIdx.x / NXCD), per_group = tmap.MTx * tmap.NG;
    w_ng = l / per_group;
    const int r = l - w_ng * per_group;
    w_rm = r / tmap.NG;
    w_rn = r - w_rm * tmap.NG;
  }
  int mt = 0, nt = 0;
  bool w_first = true;
  auto next_tile = [&]() __attribute__((always_inline)) -> bool {
    for (;;) {
      if (!w_first) {
        if (walk.step == 0) return false;                     // fewer ids than CUs: one tile per workgroup
        w_rn += walk.dr;
        w_rm += walk.dq;
        if (w_rn >= tmap.NG) { w_rn -= tmap.NG; ++w_rm; }
        while (w_rm >= tmap.MTx) { w_rm -= tmap.MTx; ++w_ng; }
      }
      w_first = false;
      if (w_ng >= tmap.NGc) return false;
      mt = w_rm * NXCD + xcd;
      nt = w_ng * tmap.NG + w_rn;
      if (mt < tmap.MT && nt < tmap.NT) return true;
    }
  };
  if (!next_tile()) return;

  // ---- loads: lane <-> (row tid / 8 + 32 i, 16-byte chunk tid % 8 of the row's 128-byte K slice)
  const int lrow = tid >> 3, lch = tid & 7;
  srd_t asrd = make_srd(A), bsrd = make_srd(B);
  unsigned aoffs[XCOL ? 1 : 8], boffs[XCOL ? 1 : 8];
  const unsigned la4 = (unsigned)lda * 4u, lb4 = (unsigned)ldb * 4u;
  const int arow32 = (int)(32u * la4), brow32 = (int)(32u * lb4);      // XCOL: scalar offset of row group i = i x this
  auto set_tile_loads = [&](bool valid) __attribute__((always_inline)) {
    const int64_t m0 = (int64_t)mt * GM, n0 = (int64_t)nt * GN;
    asrd = make_srd(A + m0 * lda);
    bsrd = make_srd(B + n0 * ldb);
    if constexpr (XCOL) {
      aoffs[0] = valid ? (unsigned)lrow * la4 + (unsigned)(lch * 16) : GOOR;
      boffs[0] = valid ? (unsigned)lrow * lb4 + (unsigned)(lch * 16) : GOOR;
    } else {
      // rows beyond M / N repeat the last row (their products land in rows / columns the store drops)
      const int mlim = (int)(M - 1 - m0 < GM - 1 ? M - 1 - m0 : GM - 1), nlim = (int)(N - 1 - n0 < GN - 1 ? N - 1 - n0 : GN - 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ra_ = lrow + 32 * i < mlim ? lrow + 32 * i : mlim, rb_ = lrow + 32 * i < nlim ? lrow + 32 * i : nlim;
        aoffs[i] = valid ? (unsigned)ra_ * la4 + (unsigned)(lch * 16) : GOOR;
        boffs[i] = valid ? (unsigned)rb_ * lb4 + (unsigned)(lch * 16) : GOOR;
      }
    }
  };
  // piece i of the stage at byte offset `so` of the rows, lanes masked by `mask`
  auto load_a = [&](int i, unsigned mask, int so) __attribute__((always_inline)) -> f32x4 {
    if constexpr (XCOL) return bload4(asrd, aoffs[0] | mask, so + i * arow32);
    else return bload4(asrd, aoffs[i] | mask, so);
  };
  auto load_b = [&](int i, unsigned mask, int so) __attribute__((always_inline)) -> f32x4 {
    if constexpr (XCOL) return bload4(bsrd, boffs[0] | mask, so + i * brow32);
    else return bload4(bsrd, boffs[i] | mask, so);
  };
  set_tile_loads(true);
  int64_t c_m0 = (int64_t)mt * GM, c_n0 = (int64_t)nt * GN;      // the tile being computed
  bool c_last = nt == tmap.NT - 1;                                 // ... lies in the last column of tiles (XCOL: it stores column N)
  // chunks at or beyond K in the last, partial K stage read offset GOOR = zero; a chunk that straddles K is fixed up
  // in LDS after it was staged (fix_tail)
  const int ktail_k0 = (KT - 1) * GBK + lch * 4;
  const bool tail_out = ktail && ktail_k0 >= K;
  const int tail_keep = (ktail && ktail_k0 < K && ktail_k0 + 4 > K) ? (int)(K - ktail_k0) : 4;
  auto load_mask = [&](int kt) __attribute__((always_inline)) -> unsigned { return (tail_out && kt == KT - 1) ? GOOR : 0u; };
  f32x4 ra[8], rb[8];
  // XCOL: rx = this lane's chunk of row N of B for the stage held in ra; xacc = this lane's partial sums of column N for
  // rows lrow + 32 i of the tile whose stages are being STAGED (a tile's stage 0 is staged during the previous tile's last
  // stage); a tile's finished sums wait for its store in LDS (xs_s)
  f32x4 rx = {0.f, 0.f, 0.f, 0.f};
  float xacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xacc[i] = 0.f;
#define XDOT(i) xacc[i] = fmaf(ra[i][3], rx[3], fmaf(ra[i][2], rx[2], fmaf(ra[i][1], rx[1], fmaf(ra[i][0], rx[0], xacc[i]))))
  // ---- staging: 4 consecutive k of one row = 8 bytes of bf16, chunk (k / 8) ^ ((row >> 2) & 3) of the row
  const int soff = lrow * GROWB + (((lch >> 1) ^ ((tid >> 5) & 3)) << 4) + ((lch & 1) << 3);
  auto stage_ab = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(ra[i][0], ra[i][1], h0, l0);
    split2n(ra[i][2], ra[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + soff + i * 32 * GROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + GARR + soff + i * 32 * GROWB) = u32x2{l0, l1};
    split2n(rb[i][0], rb[i][1], h0, l0);
    split2n(rb[i][2], rb[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + 2 * GARR + soff + i * 32 * GROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + 3 * GARR + soff + i * 32 * GROWB) = u32x2{l0, l1};
  };
  auto fix_tail = [&](char* st) __attribute__((always_inline)) {
    if (tail_keep < 4) {
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        if (e >= tail_keep) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 8; ++i)
              *reinterpret_cast<unsigned short*>(st + p * GARR + soff + i * 32 * GROWB + 2 * e) = 0;
        }
      }
    }
  };

  // ---- fragments: lane = row (lane & 31), 8 consecutive k = chunk 2 ks + (lane >> 5), swizzled as above
  const int fsw = ((lane >> 5) ^ ((lane >> 2) & 3)) << 4;            // k-step 0; k-step 1 = fsw ^ 32
  const int aoff = (wm * 128 + (lane & 31)) * GROWB, boff = 2 * GARR + (wn * 128 + (lane & 31)) * GROWB;
  // acc[i][j]: SWAPPED operands (B rows as the MFMA's A operand): lane holds row m = 32 i + lane % 32 of the wave tile
  // and columns n = 32 j + 8 (e / 4) + 4 (lane / 32) + e % 4
  f32x16 acc[4][4];

  // One stage = two k-steps x three products x 16 accumulator tiles = 96 MFMAs per wave, in the slot schedule of the tiled
  // kernel (gemm_bf16x3_big_schedule.inc).  FIRST: the stage starts the tile's accumulators (C operand 0 in its first
  // product).  (Two generated schedules -- one staging sub-step per MFMA gap; the same with a stage's last product moved
  // behind the next stage's barrier -- measured 1.3 % slower: profiles/r4_ab_gemm_big_p_schedules_rejected.jsonl.)
  auto body = [&](auto first_tag, const char* cur, char* nxt, int kt_load) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    bf16x8 al[4], bh[4], ah[4], bl[4], al1[4], bh1[4], ah1[4], bl1[4];
    const int fo0 = fsw, fo1 = fsw ^ 32;
    const int so = kt_load * GBK * 4;
    const unsigned tmask = load_mask(kt_load);
    if constexpr (XCOL) rx = *reinterpret_cast<const f32x4*>(xrow_s + ((kt_load == 0 ? KT : kt_load) - 1) * GBK + lch * 4);
#define SB __builtin_amdgcn_sched_barrier(0)
#define FRAG_RD(dst, base, i, fo) dst[i] = *reinterpret_cast<const bf16x8*>(cur + (base) + (i) * 32 * GROWB + (fo))
#define FRAG(dst, base, i, fo) FRAG_##dst(dst, base, i, fo)
#define FRAG_ah(d, b, i, f) FRAG_RD(d, b, i, f)
#define FRAG_bh(d, b, i, f) FRAG_RD(d, b, i, f)
#define FRAG_ah1(d, b, i, f) FRAG_RD(d, b, i, f)
#define FRAG_bh1(d, b, i, f) FRAG_RD(d, b, i, f)
#define FRAG_al(d, b, i, f) if constexpr (!ONE) FRAG_RD(d, b, i, f)
#define FRAG_bl(d, b, i, f) if constexpr (!ONE) FRAG_RD(d, b, i, f)
#define FRAG_al1(d, b, i, f) if constexpr (!ONE) FRAG_RD(d, b, i, f)
#define FRAG_bl1(d, b, i, f) if constexpr (!ONE) FRAG_RD(d, b, i, f)
#define MMA(x, y, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[j], x[i], acc[i][j], 0, 0, 0)
#define MMA0(x, y, i, j) { const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
                           acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y[j], x[i], z16, 0, 0, 0); }
    // the six products of a stage by operand names; the first one executed starts the tile's accumulators in a FIRST stage
#define MM(x, y, i, j) MM_##x##_##y(x, y, i, j)
#define MMZ(x, y, i, j) if constexpr (!ONE) { if constexpr (FIRST) MMA0(x, y, i, j) else MMA(x, y, i, j); }
#define MM_ah_bl(x, y, i, j) if constexpr (!ONE) MMA(x, y, i, j)
#define MM_ah_bh(x, y, i, j) if constexpr (ONE && FIRST) MMA0(x, y, i, j) else MMA(x, y, i, j)
#define MM_al1_bh1(x, y, i, j) if constexpr (!ONE) MMA(x, y, i, j)
#define MM_ah1_bl1(x, y, i, j) if constexpr (!ONE) MMA(x, y, i, j)
#define MM_ah1_bh1(x, y, i, j) MMA(x, y, i, j)
    // a staged piece in three slots: split the first pair, split the second pair, write both planes + reload
    unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
#define SPLIT(a_, b_, h_, l_) if constexpr (ONE) h_ = bf16_pair(a_, b_); else split2n(a_, b_, h_, l_)
#if BIGP_PROBE_NOB == 3   // round 6: what pre-split weights by LDS-DMA would deliver WITH the waits a correct kernel needs: the eight copies of
// a wave in the EARLY slots of the stage (they must land before the barrier that ends it), the activation operand's split /
// write / reload in the late ones (same distance to its use), and `vmcnt(8)` -- everything but the eight activation loads
// issued behind the copies -- in front of the barrier.  Timing probe, results wrong (the copies read somewhere valid).
#define SA1(i) (void)0
#define SA2(i) (void)0
#define SA3(i) { const int la_ = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) void*)(nxt + 2 * GARR + ((tid >> 6) * 8 + (i)) * 1024)); \
                 asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((unsigned)(tid * 16 + (i) * 4096)), "s"(B), "s"(la_) : "memory"); }
#define SB1(i) if constexpr (XCOL) XDOT(i); SPLIT(ra[i][0], ra[i][1], sh0, sl0)
#define SB2(i) SPLIT(ra[i][2], ra[i][3], sh1, sl1)
#define SB3(i) { *reinterpret_cast<u32x2*>(nxt + soff + i * 32 * GROWB) = u32x2{sh0, sh1};        \
               if constexpr (!ONE) *reinterpret_cast<u32x2*>(nxt + GARR + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               ra[i] = load_a(i, tmask, so)
#elif BIGP_PROBE_NOA == 2
#define SA1(i) (void)0
#define SA2(i) (void)0
#define SA3(i) { const int la_ = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) void*)(nxt + ((tid >> 6) * 8 + (i)) * 1024)); \
                 asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((unsigned)(tid * 16 + (i) * 4096)), "s"(A), "s"(la_) : "memory"); }
#elif BIGP_PROBE_NOA
#define SA1(i) (void)0
#define SA2(i) (void)0
#define SA3(i) (void)0
#else
#define SA1(i) if constexpr (XCOL) XDOT(i); SPLIT(ra[i][0], ra[i][1], sh0, sl0)
#define SA2(i) SPLIT(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) { *reinterpret_cast<u32x2*>(nxt + soff + i * 32 * GROWB) = u32x2{sh0, sh1};        \
               if constexpr (!ONE) *reinterpret_cast<u32x2*>(nxt + GARR + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               ra[i] = load_a(i, tmask, so)
#endif
#if BIGP_PROBE_NOB == 3
// (defined together with the activation operand, above)
#elif BIGP_PROBE_NOB == 2 // timing probe (wrong results): operand B as eight 1-KB LDS-DMA copies per wave and stage from somewhere valid
#define SB1(i) (void)0
#define SB2(i) (void)0
#define SB3(i) { const int la_ = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) void*)(nxt + 2 * GARR + ((tid >> 6) * 8 + (i)) * 1024)); \
                 asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((unsigned)(tid * 16 + (i) * 4096)), "s"(B), "s"(la_) : "memory"); }
#elif BIGP_PROBE_NOB    // timing probe (wrong results): operand B is neither loaded, split nor written -- what a pre-split, LDS-DMA'd weight operand could save at most
#define SB1(i) (void)0
#define SB2(i) (void)0
#define SB3(i) (void)0
#else
#define SB1(i) SPLIT(rb[i][0], rb[i][1], sh0, sl0)
#define SB2(i) SPLIT(rb[i][2], rb[i][3], sh1, sl1)
#define SB3(i) { *reinterpret_cast<u32x2*>(nxt + 2 * GARR + soff + i * 32 * GROWB) = u32x2{sh0, sh1}; \
               if constexpr (!ONE) *reinterpret_cast<u32x2*>(nxt + 3 * GARR + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               rb[i] = load_b(i, tmask, so)
#endif
#include "gemm_bf16x3_big_schedule.inc"
#undef SB3
#undef SB2
#undef SB1
#undef SA3
#undef SA2
#undef SA1
#undef SPLIT
#undef MM_ah1_bh1
#undef MM_ah1_bl1
#undef MM_al1_bh1
#undef MM_ah_bh
#undef MM_ah_bl
#undef MMZ
#undef MM
#undef MMA0
#undef MMA
#undef FRAG_bl1
#undef FRAG_al1
#undef FRAG_bl
#undef FRAG_al
#undef FRAG_bh1
#undef FRAG_ah1
#undef FRAG_bh
#undef FRAG_ah
#undef FRAG
#undef FRAG_RD
#undef SB
  };

  // ---- epilogue of the tile at (m0, n0): per wave four blocks of 32 rows x 128 columns through a 16-KB scratch in
  // the stage that was consumed last.  Scratch row = 512 B = 32 chunks of 16 B, chunk c of row r at c ^ (r & 15).
  auto epilogue = [&](int64_t m0, int64_t n0, char* scr_base, bool live, bool last_col) __attribute__((always_inline)) {
    if (PROBE == 24) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][7];
      if (sacc == 123.456f) C[tid] = sacc;
      return;
    }
    if constexpr (XCOL) {
      // column N of the full matrix: the 8 lanes of a row hold its eight 16-byte k chunks (same summation tree for every
      // row; exact fp32 products, not bit-comparable with the MFMA columns' split arithmetic -- like the tiled kernels' column)
      const float bx = bias_s[N < NBIAS ? N : 0];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = xs_s[i * GNT + tid];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += bx;
        if constexpr (ACT == 1) v = gemm_tanh(v);
        if (live && last_col && lch == 0) C[(m0 + lrow + 32 * i) * ldc + N] = v;
      }
    }
    char* scr = scr_base + wave * 16384;
    // (from an opaque copy of the lane index: as loop invariants of the tile loop the sixteen swizzled write addresses and
    // the store offsets would be hoisted out of it -- registers the stage body does not have)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int h = lv >> 5, ml = lv & 31;
    const int64_t mrow0 = m0 + (int64_t)wm * 128, ncol0 = n0 + (int64_t)wn * 128;
    const int64_t n = ncol0 + 4 * ml;                         // this lane's four columns in the read-back
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (n < NBIAS - 3 ? n : 0));
    const srd_t csrd = make_srd(C + mrow0 * ldc + ncol0);
    const bool ncol_ok = live && n + 3 < N;
    const int mleft = (int)(M - mrow0 < 128 ? M - mrow0 : 128);      // valid rows of the wave tile
    // byte offset of (row h, columns 4 ml ..) from the wave tile's first element; + 2 rows per store
    unsigned voff = ((unsigned)h * (unsigned)ldc + 4u * (unsigned)ml) * 4u;
    const unsigned vstep = (unsigned)ldc * 8u;
    // REMAP (the whole remapped tensor lies below 2 GB -- the launcher checks --, so float offsets fit 32 bits and the
    // store is a buffer store on C that out-of-range offsets switch off): this lane's column part for the two utterances
    // the wave tile's rows can lie in (K T >= 128), its row part for row h of the tile, advanced by two rows per store with
    // one carry per level.  The four permutation entries are loaded together, before the tile's first store.
    int cof[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, roff = 0, rt = 0, rk = 0;
    bool rsecond = false, one = false;
    const srd_t rsrd = make_srd(C);
    if constexpr (REMAP) {
      const unsigned cm = (unsigned)sm.cm, cq0 = (unsigned)n / cm, cr0 = (unsigned)n - cq0 * cm;
      one = n + 3 < N && cr0 + 3 < cm;
      const int64_t q0 = mrow0 / sm.T;
      const int t0 = (int)(mrow0 - q0 * sm.T);
      const int64_t b0 = q0 / sm.K;
      const int k0 = (int)(q0 - b0 * sm.K);
      const int64_t mlast = mrow0 + 127 < M ? mrow0 + 127 : M - 1;
      const int64_t b1 = mlast / sm.T / sm.K;
      const unsigned qlast = (unsigned)((N - 1) / sm.cm);          // (lanes beyond N read a valid entry and store nothing)
      const unsigned qa = cq0 < qlast ? cq0 : qlast, qb = cq0 + 1 < qlast ? cq0 + 1 : qlast;
      int pg[2][2] = {{(int)qa, (int)qb}, {(int)qa, (int)qb}};      // [utterance][group of the first column, the next group]
      if (sm.perm) {
        pg[0][0] = sm.perm[b0 * sm.perm_ld + qa]; pg[0][1] = sm.perm[b0 * sm.perm_ld + qb];
        pg[1][0] = sm.perm[b1 * sm.perm_ld + qa]; pg[1][1] = sm.perm[b1 * sm.perm_ld + qb];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool wrap = cr0 + q >= cm;
        const int cr = (int)(cr0 + q) - (wrap ? (int)cm : 0);
        cof[0][q] = pg[0][wrap ? 1 : 0] * (int)sm.co + cr;
        cof[1][q] = pg[1][wrap ? 1 : 0] * (int)sm.co + cr;
      }
      rt = t0 + h;
      rk = k0;
      roff = (int)(b0 * sm.sb + (int64_t)k0 * sm.sk + (int64_t)rt * sm.st);
      if (rt >= sm.T) { rt -= (int)sm.T; ++rk; roff += (int)(sm.sk - sm.T * sm.st); }
      if (rk >= sm.K) { rk -= (int)sm.K; rsecond = true; roff += (int)(sm.sb - sm.K * sm.sk); }
    }
    char* wp = scr + ml * 512;
    const int wsw = ml & 15;
    // ACT 2 (the folded Tanh backward, C = acc (1 - y^2)): y of the block after next is requested before a block's stores
    // -- one in-order counter: a load issued behind stores is only seen complete after them -- into two register sets
    f32x4 ya[ACT == 2 ? 2 : 1][ACT == 2 ? 16 : 1];
    const srd_t ysrd = make_srd(ACT == 2 ? sm.aux + mrow0 * sm.ldaux + ncol0 : nullptr);
    const unsigned yoff = ((unsigned)h * (unsigned)sm.ldaux + 4u * (unsigned)ml) * 4u, ystep = (unsigned)sm.ldaux * 8u;
    auto load_aux = [&](int i, f32x4 (&dst)[ACT == 2 ? 16 : 1]) __attribute__((always_inline)) {
      if constexpr (ACT == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = live && n + 3 < N && i * 32 + 2 * r + h < mleft;
          dst[r] = bload4(ysrd, ok ? yoff + (unsigned)(i * 16 + r) * ystep : GOOR, 0);
        }
      }
    };
    if constexpr (ACT == 2) { load_aux(0, ya[0]); load_aux(1, ya[1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = j * 8 + 2 * q + h;
          *reinterpret_cast<f32x4*>(wp + ((c ^ wsw) << 4)) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
      // (one wave writes and reads its own scratch: LDS operations of a wave complete in order)
      const char* rp = scr + h * 512;
      int row = i * 32 + h;
      auto store_rows = [&](int r) __attribute__((always_inline)) {
        f32x4 v = *reinterpret_cast<const f32x4*>(rp + r * 1024 + ((ml ^ ((2 * r + h) & 15)) << 4));
        v += bv;
        if constexpr (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gemm_tanh(v[e]);
        }
        if constexpr (ACT == 2) {
          const f32x4 y = ya[i & 1][r];
          v *= 1.f - y * y;
        }
        if constexpr (!REMAP) {
          const bool ok = ncol_ok && row < mleft;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), csrd, (int)(ok ? voff : GOOR), 0, (PROBE & 32) ? 0 : (PROBE & 64) ? 16 : 2);
          voff += vstep;
        } else {
          const bool ok = live && row < mleft;
          const int base = roff + (rsecond ? cof[1][0] : cof[0][0]);
          // (unconditional, like the plain store: the straight-line count of stores is what the next stage's wait relies on)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrd, (int)(ok && one ? (unsigned)base * 4u : GOOR), 0, 2);
          if (!one && ok) {      // the lane's four columns straddle two groups or the matrix edge (3 lanes in 513): one by one
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (n + q < N) C[roff + (rsecond ? cof[1][q] : cof[0][q])] = v[q];
          }
          rt += 2;
          roff += 2 * (int)sm.st;
          if (rt >= sm.T) { rt -= (int)sm.T; ++rk; roff += (int)(sm.sk - sm.T * sm.st); }
          if (rk >= sm.K) { rk -= (int)sm.K; rsecond = true; roff += (int)(sm.sb - sm.K * sm.sk); }
        }
        row += 2;
      };
      // all 64 stores of a tile as straight-line code -- the compiler's count of what is in flight when the next stage first
      // touches the prefetched operands must reach the counter's limit (`vmcnt(63)`: the sixteen loads and ONE store done);
      // with the stores in a loop it assumed a single trip and waited for half of them
#pragma unroll
      for (int r = 0; r < 16; ++r) store_rows(r);
      if constexpr (ACT == 2) {
        if (i + 2 < 4) load_aux(i + 2, ya[i & 1]);
      }
    }
  };

  // ---- prologue: stage 0 of the first tile -> LDS, its stage 1 -> registers
  {
    const unsigned t0 = load_mask(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = load_a(i, t0, 0); rb[i] = load_b(i, t0, 0); }
    if constexpr (XCOL) {
      __syncthreads();                       // xrow_s
      rx = *reinterpret_cast<const f32x4*>(xrow_s + lch * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) XDOT(i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_ab(lds, i);
    const unsigned t1 = load_mask(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = load_a(i, t1, GBK * 4); rb[i] = load_b(i, t1, GBK * 4); }
  }
  __syncthreads();
  int par = 0;
  bool more = true, live = false, n_last = false, p_last = false;
  int64_t n_m0 = 0, n_n0 = 0, p_m0 = 0, p_n0 = 0;
  // from stage KT - 2 of a tile on, the loads (two stages ahead) belong to the next tile of the list
  auto loader_stage = [&](int kt) __attribute__((always_inline)) -> int {
    int ktl = kt + 2;
    if (ktl == KT) {
      more = next_tile();
      set_tile_loads(more);
      n_m0 = (int64_t)mt * GM;
      n_n0 = (int64_t)nt * GN;
      n_last = nt == tmap.NT - 1;
    }
    return ktl >= KT ? ktl - KT : ktl;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (;;) {
    // The store of the tile finished in the previous round (p_m0, p_n0), through the stage consumed last.  The FIRST round
    // runs it too, with every store out of range: the stage body below then has ONE predecessor, and the compiler's count of
    // the memory operations in flight at its first use of the prefetched operands is "16 loads, then 64 stores" on every
    // path -- with a separate entry from the prologue it waited for `vmcnt(15)` there, i.e. for 49 of the 64 stores just
    // issued (one in-order counter for loads and stores on this architecture).
    epilogue(p_m0, p_n0, lds + (par ^ 1) * GSTAGE, live, p_last);
    if (live && !more) break;
    __syncthreads();                       // every wave's scratch is read before the next stage is written over it
    {
      const int ktl = loader_stage(0);
      char* nxt = lds + (par ^ 1) * GSTAGE;
      // stores stage 1 (registers), loads stage 2
      body(std::true_type{}, lds + par * GSTAGE, nxt, ktl);
      if (ktail && KT == 2) fix_tail(nxt);
      if (BIGP_PROBE_NOB == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
    }
    for (int kt = 1; kt < KT; ++kt) {
      const int ktl = loader_stage(kt);
      char* nxt = lds + (par ^ 1) * GSTAGE;
      if (XCOL && kt == KT - 1) {            // this stage body stages the NEXT tile's stage 0: the column sums of this tile are complete
#pragma unroll
        for (int i = 0; i < 8; ++i) { xs_s[i * GNT + tid] = xacc[i]; xacc[i] = 0.f; }      // (read back by the same thread, in the store)
      }
      body(std::false_type{}, lds + par * GSTAGE, nxt, ktl);
      if (ktail && kt + 1 == KT - 1) fix_tail(nxt);
      if (BIGP_PROBE_NOB == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
    }
    p_m0 = c_m0;
    p_n0 = c_n0;
    p_last = c_last;
    c_m0 = n_m0;
    c_n0 = n_n0;
    c_last = n_last;
    live = true;
  }
#undef XDOT
}

}  // namespace

int tssep_gemm_bf16x3_bigp_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (g->a_kmajor || g->b_kmajor || g->splitk > 1 || g->kperiod > 0 || g->b_ones_col) return TSSEP_E_UNSUPPORTED;
  // the remapped store: groups of >= 4 columns, >= 64 frames, a wave tile's 128 rows within two utterances
  const bool remap = sm.remap != 0;
  if (remap) {
    if (sm.remap != 1 || sm.T < 64 || sm.cm < 4 || sm.K * sm.T < 128) return TSSEP_E_UNSUPPORTED;
    // 32-bit float offsets into the remapped tensor (buffer stores): its last element below 2 GB
    const int64_t last = ((g->M - 1) / (sm.T * sm.K)) * sm.sb + (sm.K - 1) * sm.sk + (sm.T - 1) * sm.st + ((g->N - 1) / sm.cm) * sm.co + sm.cm;
    if (last >= ((int64_t)1 << 29) || sm.sb < 0 || sm.sk < 0 || sm.st < 0 || sm.co < 0) return TSSEP_E_UNSUPPORTED;
  }
  // bias and the Tanh only: a store that reads (accumulate, the folded Tanh backward's aux operand) keeps the tiled kernel
  // N = 256 q + 1: q tiles + one VALU column, when no row of a tile needs clamping (see the kernel)
  const bool xcol = !remap && g->act != 2 && g->N > 256 && g->N % 256 == 1;
  if (xcol && ((g->K & 3) || g->K > XROW || g->N > XBIAS || g->M % GM)) return TSSEP_E_UNSUPPORTED;
  if (g->accumulate || g->N > PBIAS) return TSSEP_E_UNSUPPORTED;
  // the folded Tanh backward: 16-byte rows of y, a wave tile's 128 rows below 2 GB
  if (g->act == 2 && (!sm.aux || (sm.ldaux & 3) || !aligned16(sm.aux) || (g->N & 3) || (int64_t)130 * sm.ldaux * 4 >= (int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
  if (!remap && ((!xcol && (g->N & 3)) || (sm.ldc & 3) || !aligned16(g->C))) return TSSEP_E_UNSUPPORTED;
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B)) return TSSEP_E_UNSUPPORTED;
  if (g->M < 4 * GM || g->K < 2 * GBK) return TSSEP_E_UNSUPPORTED;      // (two K stages: the loader's lead)
  // 32-bit buffer offsets: one tile's rows and the whole K extent must stay below 2 GB
  if ((int64_t)GM * g->lda * 4 + g->K * 4 >= (int64_t)1 << 31 || (int64_t)GN * g->ldb * 4 + g->K * 4 >= (int64_t)1 << 31 ||
      (!remap && (int64_t)(GM + 2) * sm.ldc * 4 >= (int64_t)1 << 31))
    return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + GM - 1) / GM, xcol ? (g->N - 1) / GN : (g->N + GN - 1) / GN, 1);
  const int64_t nids = tile_map_blocks(tm);
  const int ncu = current_device_cus();      // (per device, gemm_common.h)
  // persistent: a multiple of 8 workgroups (a workgroup's ids stay on one XCD) unless the whole list fits the CUs once
  const int ncu8 = ncu / NXCD * NXCD;
  const int64_t grid = nids <= ncu8 || ncu8 == 0 ? nids : ncu8;
  TileWalk walk;
  walk.step = grid == nids ? 0 : (int)(grid / NXCD);
  walk.dq = walk.step / tm.NG;
  walk.dr = walk.step % tm.NG;
  const bool one = g->precision == 3;      // the plain-bf16 side line
#define PLAUNCH2(P_, ACT_, X_, R_, O_) hipLaunchKernelGGL((gemm_bf16x3_bigp_kernel<P_, ACT_, X_, R_, O_>), dim3((unsigned)grid), dim3(GNT), 0, (hipStream_t)stream, \
                     g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, sm.ldc, g->bias, tm, walk, sm)
#define PLAUNCH1(P_, ACT_, X_, R_) do { if (one) PLAUNCH2(P_, ACT_, X_, R_, true); else PLAUNCH2(P_, ACT_, X_, R_, false); } while (0)
#define PLAUNCH(P_) do { if (remap) { if (g->act == 2) PLAUNCH1(P_, 2, false, true); else if (g->act == 1) PLAUNCH1(P_, 1, false, true); else PLAUNCH1(P_, 0, false, true); }      \
                         else if (xcol) { if (g->act == 1) PLAUNCH1(P_, 1, true, false); else PLAUNCH1(P_, 0, true, false); }      \
                         else { if (g->act == 2) PLAUNCH1(P_, 2, false, false); else if (g->act == 1) PLAUNCH1(P_, 1, false, false); else PLAUNCH1(P_, 0, false, false); } } while (0)
#ifdef TSSEP_GEMM_EXP
  {
    const char* pe = getenv("TSSEP_BIGP_PROBE");
    switch (pe ? atoi(pe) : 0) {
      case 24: PLAUNCH(24); return tssep_launch_status();
      case 32: PLAUNCH(32); return tssep_launch_status();
      case 64: PLAUNCH(64); return tssep_launch_status();
      default: break;
    }
  }
#endif
  PLAUNCH(0);
#undef PLAUNCH
#undef PLAUNCH1
#undef PLAUNCH2
  return tssep_launch_status();
}
