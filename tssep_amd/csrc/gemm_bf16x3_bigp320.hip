// 192 x 320 variant of the persistent big-tile split-bf16 GEMM (gemm_bf16x3_bigp.hip) for row-major x row-major
// operands with N = 320 q (or close below): the Tanh projections of birnn0 / birnn1 (N = 320, with and without the
// speaker combination) and the d(input) GEMM of birnn1 with the folded Tanh backward (tssep/train/rnnp.py:88-96,146-161,
// tssep/train/net.py:608-625).
//
// Why: N = 320 pads a 256-wide tile by 60 %; the 256 x 160 tile those GEMMs ran on (gemm_bf16x3_nt_w160.hip: wave tiles
// of 64 x 160, one tile per workgroup) pays 35 % more fragment reads, staging and barriers per MFMA than a 128 x 128
// wave tile, stages every A tile twice (once per column tile) and has the per-tile drain / dispatch / prologue the
// persistent kernel removed.  Here a workgroup's tile is 192 rows x 320 columns: four waves (one per SIMD) as 2 x 2,
// wave tile 96 x 160 = 3 x 5 MFMA tiles = 240 accumulator registers, 90 MFMAs per wave and K stage; per stage 192 + 320
// rows of 32 k are staged -- 64 KB of LDS like the 256 x 256 tile -- and at N = 320 every A tile is staged ONCE.
// Everything else is the persistent kernel's: one software pipeline over (tile, K stage) per CU, swapped MFMA operands
// with the 16-byte transposition through LDS (per 32-row block: columns 0-127 as there, then columns 128-159 through the
// same scratch), buffer stores, the dummy first store section, bit-identical results.  Stores: plain / bias / Tanh / the
// folded Tanh backward; remapped rows with ONE column group and no permutation (the speaker combination).  Slot
// schedule: generated (tools/gen/gen_bigp320_schedule.py).
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int GM = 192, GN = 320, GBK = 32, GNT = 256;
constexpr int NPA = GM / 32, NPB = GN / 32;                 // staged pieces per thread and stage: 6 of A, 10 of B
constexpr int GROWB = 64;                                   // bytes per LDS row: 32 bf16
constexpr int AR_A = GM * GROWB, AR_B = GN * GROWB;         // 12 288 / 20 480 B per plane
constexpr int GSTAGE = 2 * AR_A + 2 * AR_B;                 // A hi, A lo, B hi, B lo = 65 536 B
constexpr int WTM = 96, WTN = 160;                          // wave tile
constexpr unsigned GOOR = 0x80000000u;                      // buffer offset beyond the range: loads return 0, stores are dropped
constexpr int PBIAS = 4096;                                 // floats of bias kept in LDS (N beyond that: the other kernels)

// (see gemm_bf16x3_bigp.hip)
struct TileWalk { int step, dq, dr; };

// ACT: 0 plain / bias, 1 Tanh, 2 the folded Tanh backward (C = acc (1 - y^2), y = sm.aux).  REMAP: row m = (b K + k) T + t
// -> b sb + k sk + t st, columns unchanged (one group, no permutation), the whole tensor below 2 GB.
// ONE (tssep_gemm_args.precision = 3, the plain-bf16 side line): operands rounded to bf16, ONE product a_hi b_hi per k-step --
// the lo planes are neither staged nor read, a stage has 30 MFMAs per wave.
template <int ACT, bool REMAP, bool ONE>
__global__ __launch_bounds__(GNT, 1) void gemm_bf16x3_bigp320_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N, int64_t K,
    int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, TileMap tmap, TileWalk walk, StoreMap sm) {
  __shared__ __attribute__((aligned(16))) char lds[2 * GSTAGE];
  // the bias, read by the store from LDS: a global load would sit behind the prefetched operand tiles in the in-order
  // memory counter
  __shared__ __attribute__((aligned(16))) float bias_s[PBIAS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int KT = (int)((K + GBK - 1) / GBK);
  const bool ktail = (K % GBK) != 0;
  for (int i = tid; i < PBIAS; i += GNT) bias_s[i] = (bias && i < N) ? bias[i] : 0.f;      // (visible after the prologue's barrier)

  // ---- tile list of this workgroup (TileWalk)
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
  unsigned aoffs[NPA], boffs[NPB];
  const unsigned la4 = (unsigned)lda * 4u, lb4 = (unsigned)ldb * 4u;
  auto set_tile_loads = [&](bool valid) __attribute__((always_inline)) {
    const int64_t m0 = (int64_t)mt * GM, n0 = (int64_t)nt * GN;
    asrd = make_srd(A + m0 * lda);
    bsrd = make_srd(B + n0 * ldb);
    // rows beyond M / N repeat the last row (their products land in rows / columns the store drops)
    const int mlim = (int)(M - 1 - m0 < GM - 1 ? M - 1 - m0 : GM - 1), nlim = (int)(N - 1 - n0 < GN - 1 ? N - 1 - n0 : GN - 1);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int r_ = lrow + 32 * i < mlim ? lrow + 32 * i : mlim;
      aoffs[i] = valid ? (unsigned)r_ * la4 + (unsigned)(lch * 16) : GOOR;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int r_ = lrow + 32 * i < nlim ? lrow + 32 * i : nlim;
      boffs[i] = valid ? (unsigned)r_ * lb4 + (unsigned)(lch * 16) : GOOR;
    }
  };
  set_tile_loads(true);
  int64_t c_m0 = (int64_t)mt * GM, c_n0 = (int64_t)nt * GN;      // the tile being computed
  // chunks at or beyond K in the last, partial K stage read offset GOOR = zero; a chunk that straddles K is fixed up
  // in LDS after it was staged (fix_tail)
  const int ktail_k0 = (KT - 1) * GBK + lch * 4;
  const bool tail_out = ktail && ktail_k0 >= K;
  const int tail_keep = (ktail && ktail_k0 < K && ktail_k0 + 4 > K) ? (int)(K - ktail_k0) : 4;
  auto load_mask = [&](int kt) __attribute__((always_inline)) -> unsigned { return (tail_out && kt == KT - 1) ? GOOR : 0u; };
  f32x4 ra[NPA], rb[NPB];

  // ---- staging: 4 consecutive k of one row = 8 bytes of bf16, chunk (k / 8) ^ ((row >> 2) & 3) of the row
  const int soff = lrow * GROWB + (((lch >> 1) ^ ((tid >> 5) & 3)) << 4) + ((lch & 1) << 3);
  auto stage_a = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(ra[i][0], ra[i][1], h0, l0);
    split2n(ra[i][2], ra[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + soff + i * 32 * GROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + AR_A + soff + i * 32 * GROWB) = u32x2{l0, l1};
  };
  auto stage_b = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(rb[i][0], rb[i][1], h0, l0);
    split2n(rb[i][2], rb[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + 2 * AR_A + soff + i * 32 * GROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + 2 * AR_A + AR_B + soff + i * 32 * GROWB) = u32x2{l0, l1};
  };
  auto fix_tail = [&](char* st) __attribute__((always_inline)) {
    if (tail_keep < 4) {
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        if (e >= tail_keep) {
#pragma unroll
          for (int i = 0; i < NPA; ++i) {
            *reinterpret_cast<unsigned short*>(st + soff + i * 32 * GROWB + 2 * e) = 0;
            *reinterpret_cast<unsigned short*>(st + AR_A + soff + i * 32 * GROWB + 2 * e) = 0;
          }
#pragma unroll
          for (int i = 0; i < NPB; ++i) {
            *reinterpret_cast<unsigned short*>(st + 2 * AR_A + soff + i * 32 * GROWB + 2 * e) = 0;
            *reinterpret_cast<unsigned short*>(st + 2 * AR_A + AR_B + soff + i * 32 * GROWB + 2 * e) = 0;
          }
        }
      }
    }
  };

  // ---- fragments: lane = row (lane & 31), 8 consecutive k = chunk 2 ks + (lane >> 5), swizzled as above
  const int fsw = ((lane >> 5) ^ ((lane >> 2) & 3)) << 4;            // k-step 0; k-step 1 = fsw ^ 32
  const int aoff = (wm * WTM + (lane & 31)) * GROWB, boff = 2 * AR_A + (wn * WTN + (lane & 31)) * GROWB;
  // acc[i][j]: SWAPPED operands (B rows as the MFMA's A operand): lane holds row m = 32 i + lane % 32 of the wave tile
  // and columns n = 32 j + 8 (e / 4) + 4 (lane / 32) + e % 4
  f32x16 acc[3][5];

  // One stage = two k-steps x three products x 15 accumulator tiles = 90 MFMAs per wave.  FIRST: the stage starts the
  // tile's accumulators (C operand 0 in its first product).
  auto body = [&](auto first_tag, const char* cur, char* nxt, int kt_load) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    bf16x8 al[3], bh[5], ah[3], bl[5], al1[3], bh1[5], ah1[3], bl1[5];
    const int fo0 = fsw, fo1 = fsw ^ 32;
    const int so = kt_load * GBK * 4;
    const unsigned tmask = load_mask(kt_load);
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
#define SA1(i) SPLIT(ra[i][0], ra[i][1], sh0, sl0)
#define SA2(i) SPLIT(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) { *reinterpret_cast<u32x2*>(nxt + soff + i * 32 * GROWB) = u32x2{sh0, sh1};        \
               if constexpr (!ONE) *reinterpret_cast<u32x2*>(nxt + AR_A + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               ra[i] = bload4(asrd, aoffs[i] | tmask, so)
#define SB1(i) SPLIT(rb[i][0], rb[i][1], sh0, sl0)
#define SB2(i) SPLIT(rb[i][2], rb[i][3], sh1, sl1)
#define SB3(i) { *reinterpret_cast<u32x2*>(nxt + 2 * AR_A + soff + i * 32 * GROWB) = u32x2{sh0, sh1}; \
               if constexpr (!ONE) *reinterpret_cast<u32x2*>(nxt + 2 * AR_A + AR_B + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               rb[i] = bload4(bsrd, boffs[i] | tmask, so)
#include "gemm_bf16x3_bigp320_schedule.inc"
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

  // ---- epilogue of the tile at (m0, n0): per wave three blocks of 32 rows; a block leaves in two passes through the wave's
  // 16-KB scratch in the stage that was consumed last: columns 0-127 (rows of 512 B = 32 chunks of 16 B, chunk c of row r
  // at c ^ (r & 15); read back as 2 rows x 512 B per wave instruction), then columns 128-159 (rows of 144 B -- 36 dwords:
  // conflict-free both ways --; read back as 8 rows x 128 B per wave instruction).
  auto epilogue = [&](int64_t m0, int64_t n0, char* scr_base, bool live) __attribute__((always_inline)) {
    char* scr = scr_base + wave * 16384;
    // (from an opaque copy of the lane index: loop invariants of the tile loop would be hoisted out of it -- registers the
    // stage body does not have)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int h = lv >> 5, ml = lv & 31, l8 = lv & 7, r8 = lv >> 3;
    const int64_t mrow0 = m0 + (int64_t)wm * WTM, ncol0 = n0 + (int64_t)wn * WTN;
    const int64_t n1 = ncol0 + 4 * ml, n2 = ncol0 + 128 + 4 * l8;      // this lane's four columns in the two read-backs
    const f32x4 bv1 = *reinterpret_cast<const f32x4*>(bias_s + (n1 < PBIAS - 3 ? n1 : 0));
    const f32x4 bv2 = *reinterpret_cast<const f32x4*>(bias_s + (n2 < PBIAS - 3 ? n2 : 0));
    const bool ok1 = live && n1 + 3 < N, ok2 = live && n2 + 3 < N;
    const int mleft = (int)(M - mrow0 < WTM ? M - mrow0 : WTM);      // valid rows of the wave tile
    // plain store: byte offsets from the wave tile's first element; REMAP: float offsets into the whole tensor (< 2 GB)
    const srd_t csrd = make_srd(REMAP ? C : C + mrow0 * ldc + ncol0);
    // REMAP: the row part of row r of the wave tile = roff0 + r st, + one carry per level
    int t0 = 0, k0 = 0, roff0 = 0;
    if constexpr (REMAP) {
      const int64_t q0 = mrow0 / sm.T;
      t0 = (int)(mrow0 - q0 * sm.T);
      const int64_t b0 = q0 / sm.K;
      k0 = (int)(q0 - b0 * sm.K);
      roff0 = (int)(b0 * sm.sb + (int64_t)k0 * sm.sk + (int64_t)t0 * sm.st);
    }
    // byte offset of (row r of the wave tile, this lane's columns of pass P)
    auto row_off = [&](int r, int64_t n) __attribute__((always_inline)) -> unsigned {
      if constexpr (REMAP) {
        int t = t0 + r, k = k0, off = roff0 + r * (int)sm.st;
        if (t >= sm.T) { t -= (int)sm.T; ++k; off += (int)(sm.sk - sm.T * sm.st); }
        if (k >= sm.K) off += (int)(sm.sb - sm.K * sm.sk);
        return (unsigned)(off + (int)n) * 4u;
      } else {
        return ((unsigned)r * (unsigned)ldc + (unsigned)(n - ncol0)) * 4u;
      }
    };
    // ACT 2: y of a block (16 + 4 pieces of 16 bytes per lane), requested in front of the block's transposition
    f32x4 ya[ACT == 2 ? 20 : 1];
    const srd_t ysrd = make_srd(ACT == 2 ? sm.aux + mrow0 * sm.ldaux + ncol0 : nullptr);
    auto finish = [&](f32x4 v, const f32x4& bv, const f32x4& y) __attribute__((always_inline)) -> f32x4 {
      v += bv;
      if constexpr (ACT == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gemm_tanh(v[e]);
      }
      if constexpr (ACT == 2) v *= 1.f - y * y;
      return v;
    };
    char* wp = scr + ml * 512;
    const int wsw = ml & 15;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if constexpr (ACT == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + 2 * r + h;
          ya[r] = bload4(ysrd, ok1 && row < mleft ? ((unsigned)row * (unsigned)sm.ldaux + 4u * (unsigned)ml) * 4u : GOOR, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i * 32 + 8 * r + r8;
          ya[16 + r] = bload4(ysrd, ok2 && row < mleft ? ((unsigned)row * (unsigned)sm.ldaux + 128u + 4u * (unsigned)l8) * 4u : GOOR, 0);
        }
      }
      // ---- pass 1: columns 0-127
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = j * 8 + 2 * q + h;
          *reinterpret_cast<f32x4*>(wp + ((c ^ wsw) << 4)) =
              f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
      // (one wave writes and reads its own scratch: LDS operations of a wave complete in order)
      auto store1 = [&](int r) __attribute__((always_inline)) {
        const int row = i * 32 + 2 * r + h;
        f32x4 v = *reinterpret_cast<const f32x4*>(scr + (2 * r + h) * 512 + ((ml ^ ((2 * r + h) & 15)) << 4));
        v = finish(v, bv1, ya[ACT == 2 ? r : 0]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), csrd, (int)(ok1 && row < mleft ? row_off(row, n1) : GOOR), 0, 2);
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) store1(r);
      // ---- pass 2: columns 128-159 (MFMA tile j = 4): lane (row lane % 32, half h) holds columns 8 q + 4 h .. + 3
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(scr + ml * 144 + (2 * q + h) * 16) =
            f32x4{acc[i][4][4 * q], acc[i][4][4 * q + 1], acc[i][4][4 * q + 2], acc[i][4][4 * q + 3]};
      auto store2 = [&](int r) __attribute__((always_inline)) {
        const int row = i * 32 + 8 * r + r8;
        f32x4 v = *reinterpret_cast<const f32x4*>(scr + (8 * r + r8) * 144 + l8 * 16);
        v = finish(v, bv2, ya[ACT == 2 ? 16 + r : 0]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), csrd, (int)(ok2 && row < mleft ? row_off(row, n2) : GOOR), 0, 2);
      };
#pragma unroll
      for (int r = 0; r < 4; ++r) store2(r);
    }
  };

  // ---- prologue: stage 0 of the first tile -> LDS, its stage 1 -> registers
  {
    const unsigned t0 = load_mask(0);
#pragma unroll
    for (int i = 0; i < NPA; ++i) ra[i] = bload4(asrd, aoffs[i] | t0, 0);
#pragma unroll
    for (int i = 0; i < NPB; ++i) rb[i] = bload4(bsrd, boffs[i] | t0, 0);
#pragma unroll
    for (int i = 0; i < NPA; ++i) stage_a(lds, i);
#pragma unroll
    for (int i = 0; i < NPB; ++i) stage_b(lds, i);
    const unsigned t1 = load_mask(1);
#pragma unroll
    for (int i = 0; i < NPA; ++i) ra[i] = bload4(asrd, aoffs[i] | t1, GBK * 4);
#pragma unroll
    for (int i = 0; i < NPB; ++i) rb[i] = bload4(bsrd, boffs[i] | t1, GBK * 4);
  }
  __syncthreads();
  int par = 0;
  bool more = true, live = false;
  int64_t n_m0 = 0, n_n0 = 0, p_m0 = 0, p_n0 = 0;
  // from stage KT - 2 of a tile on, the loads (two stages ahead) belong to the next tile of the list
  auto loader_stage = [&](int kt) __attribute__((always_inline)) -> int {
    int ktl = kt + 2;
    if (ktl == KT) {
      more = next_tile();
      set_tile_loads(more);
      n_m0 = (int64_t)mt * GM;
      n_n0 = (int64_t)nt * GN;
    }
    return ktl >= KT ? ktl - KT : ktl;
  };
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (;;) {
    // the store of the tile finished in the previous round; the first round runs it with every store out of range (one
    // predecessor of the stage body: gemm_bf16x3_bigp.hip)
    epilogue(p_m0, p_n0, lds + (par ^ 1) * GSTAGE, live);
    if (live && !more) break;
    __syncthreads();                       // every wave's scratch is read before the next stage is written over it
    {
      const int ktl = loader_stage(0);
      char* nxt = lds + (par ^ 1) * GSTAGE;
      body(std::true_type{}, lds + par * GSTAGE, nxt, ktl);        // stores stage 1 (registers), loads stage 2
      if (ktail && KT == 2) fix_tail(nxt);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
    }
    for (int kt = 1; kt < KT; ++kt) {
      const int ktl = loader_stage(kt);
      char* nxt = lds + (par ^ 1) * GSTAGE;
      body(std::false_type{}, lds + par * GSTAGE, nxt, ktl);
      if (ktail && kt + 1 == KT - 1) fix_tail(nxt);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
    }
    p_m0 = c_m0;
    p_n0 = c_n0;
    c_m0 = n_m0;
    c_n0 = n_n0;
    live = true;
  }
}

}  // namespace

int tssep_gemm_bf16x3_bigp320_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (g->a_kmajor || g->b_kmajor || g->splitk > 1 || g->kperiod > 0 || g->b_ones_col) return TSSEP_E_UNSUPPORTED;
  const bool remap = sm.remap != 0;
  if (remap) {
    // rows only: one column group, no permutation; >= 96 frames (a wave tile's rows cross at most one boundary per level);
    // 32-bit float offsets into the remapped tensor: its last element below 2 GB
    if (sm.remap != 1 || sm.perm || sm.cm < g->N || sm.T < WTM) return TSSEP_E_UNSUPPORTED;
    const int64_t last = ((g->M - 1) / (sm.T * sm.K)) * sm.sb + (sm.K - 1) * sm.sk + (sm.T - 1) * sm.st + g->N;
    if (last >= ((int64_t)1 << 29) || sm.sb < 0 || sm.sk < 0 || sm.st < 0) return TSSEP_E_UNSUPPORTED;
  }
  if (g->accumulate || g->N > PBIAS || (g->N & 3)) return TSSEP_E_UNSUPPORTED;
  if (!remap && ((sm.ldc & 3) || !aligned16(g->C) || (int64_t)(GM + 2) * sm.ldc * 4 >= (int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
  if (remap && !aligned16(g->C)) return TSSEP_E_UNSUPPORTED;
  if (g->act == 2 && (!sm.aux || (sm.ldaux & 3) || !aligned16(sm.aux) || (int64_t)(WTM + 2) * sm.ldaux * 4 >= (int64_t)1 << 31)) return TSSEP_E_UNSUPPORTED;
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B)) return TSSEP_E_UNSUPPORTED;
  if (g->M < 4 * GM || g->K < 2 * GBK) return TSSEP_E_UNSUPPORTED;      // (two K stages: the loader's lead)
  // 32-bit buffer offsets: one tile's rows and the whole K extent must stay below 2 GB
  if ((int64_t)GM * g->lda * 4 + g->K * 4 >= (int64_t)1 << 31 || (int64_t)GN * g->ldb * 4 + g->K * 4 >= (int64_t)1 << 31) return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + GM - 1) / GM, (g->N + GN - 1) / GN, 1);
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
#define QLAUNCH2(ACT_, R_, O_) hipLaunchKernelGGL((gemm_bf16x3_bigp320_kernel<ACT_, R_, O_>), dim3((unsigned)grid), dim3(GNT), 0, (hipStream_t)stream, \
                     g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, sm.ldc, g->bias, tm, walk, sm)
#define QLAUNCH(ACT_, R_) do { if (one) QLAUNCH2(ACT_, R_, true); else QLAUNCH2(ACT_, R_, false); } while (0)
  if (remap) { if (g->act == 2) QLAUNCH(2, true); else if (g->act == 1) QLAUNCH(1, true); else QLAUNCH(0, true); }
  else { if (g->act == 2) QLAUNCH(2, false); else if (g->act == 1) QLAUNCH(1, false); else QLAUNCH(0, false); }
#undef QLAUNCH2
#undef QLAUNCH
  return tssep_launch_status();
}
