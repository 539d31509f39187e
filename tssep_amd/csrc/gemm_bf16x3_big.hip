// "Big tile" variant of the split-bf16 GEMM for row-major x row-major operands (both k-contiguous) with wide N:
// the LSTM input projections (N = 2400) and the d(input) GEMM of the combination layer (N = 1280)
// (tssep/train/rnnp.py:88-96,146-153).
//
// Why: the split-bf16 kernels are bound by the bytes that return to the vector registers per MFMA -- fragment
// reads from LDS plus the operand tiles from global memory -- not by the matrix pipe: a CU takes back ~90 B per
// clock (probes: 128 KB of fragment reads alone cost 1 200-1 600 cycles per stage, profiles/r3_gemm_stream_probes.jsonl)
// while its four SIMDs can retire 4 096 MFMA flops per clock, so a kernel needs >= 46 flops per returned byte.  A
// 64 x 64 wave tile (the streaming kernel) gets 35, the 128 x 64 one of the 256 x 256 / 8-wave kernel 48; this kernel
// gives every wave a 128 x 128 tile = 96 flops per fragment byte (75 with the global loads):
//  * 256 x 256 output tile, FOUR waves (one per SIMD) of 4 x 4 MFMA tiles: 256 accumulator registers per lane, which
//    is what one wave per SIMD can hold (512 registers; the 8-wave kernels have 256 per wave);
//  * with a single wave per SIMD nothing hides a stall, so a stage is written slot by slot (sched_barrier pins
//    the order): 96 MFMAs per stage (K staged 32 at a time), the fragments of the next k-step requested in two halves
//    while the current one computes (<= 96 fragment registers), and every 16-byte piece of the next stage split,
//    written to the other LDS stage and RELOADED (for the stage after next) in one slot: a whole stage of latency
//    budget per load;
//  * LDS: bf16 hi / lo rows of 64 B, 16-byte chunks XOR-swizzled with (row >> 2) & 3 (conflict-free fragment reads
//    and staging writes), two stages = 128 KB; the epilogue reuses them for the row-transposed stores of
//    gemm_common.h (bias, Tanh and its folded backward, accumulate, the store remaps) in four 64 x 64 blocks per
//    wave;
//  * same k order per output element and same epilogue arithmetic as the other split-bf16 kernels: bit-identical.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

// (experiment builds) truncating split: hi = upper 16 bits, lo = upper 16 bits of x - hi; v_and / v_perm / v_pk_add
// only, no v_cvt_pk_bf16_f32
__device__ __forceinline__ void split2n_trunc(float a, float b, unsigned& hi, unsigned& lo) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  const float la = a - __uint_as_float(ua & 0xffff0000u), lb = b - __uint_as_float(ub & 0xffff0000u);
  lo = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

constexpr int GM = 256, GN = 256, GBK = 32, GNT = 256;
constexpr int GROWB = 64;                                   // bytes per LDS row: 32 bf16
constexpr int GARR = GM * GROWB;                            // 16 384 B per plane
constexpr int GSTAGE = 4 * GARR;                            // A hi, A lo, B hi, B lo = 65 536 B
constexpr unsigned GOOR = 0x80000000u;                      // buffer offset beyond the range: the load returns 0

// PROBE (experiment builds only, -DTSSEP_GEMM_EXP; garbage results, TIMING probes): 1 = no barriers, 2 = no global
// loads, 4 = no staging, 8 = no epilogue stores, 16 = no MFMAs
// XCOL (N = 256 q + 1: the 513 frequency bins of `dgrad birnn0 dx` and of the pre-net projection): the MFMA tiles
// cover the first N - 1 columns and column N - 1 is computed on the VALU from the raw fp32 A values every thread
// stages anyway (32 FMAs per thread and stage against row N - 1 of B, exact fp32; every workgroup computes it --
// branch-free -- and those of the last column tile store it), instead of a third 256-wide tile for ONE column.
template <int PROBE, bool XCOL>
__global__ __launch_bounds__(GNT, 1) void gemm_bf16x3_big_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t Nfull,
    int64_t K, int64_t lda, int64_t ldb, const float* __restrict__ bias, int act, int accumulate, StoreMap sm,
    TileMap tmap) {
  const int64_t N = XCOL ? Nfull - 1 : Nfull;          // columns of the MFMA tiles
  __shared__ __attribute__((aligned(16))) char lds[2 * GSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * GM, n0 = (int64_t)nt * GN;
  const int KT = (int)((K + GBK - 1) / GBK);
  const bool ktail = (K % GBK) != 0;

  // ---- loads: lane <-> (row tid / 8 + 32 i, 16-byte chunk tid % 8 of the row's 128-byte K slice)
  const int lrow = tid >> 3, lch = tid & 7;
  const srd_t asrd = make_srd(A + m0 * lda), bsrd = make_srd(B + n0 * ldb);
  unsigned aoffs[8], boffs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t r = m0 + lrow + 32 * i;
    r = r > M - 1 ? M - 1 : r;
    aoffs[i] = (unsigned)(((r - m0) * lda + lch * 4) * 4);
    r = n0 + lrow + 32 * i;
    r = r > N - 1 ? N - 1 : r;
    boffs[i] = (unsigned)(((r - n0) * ldb + lch * 4) * 4);
  }
  // chunks at or beyond K in the last, partial K stage read offset GOOR = zero; a chunk that straddles K is fixed up
  // in LDS after it was staged (fix_tail); loads behind the last stage are sent out of range as a whole
  const int ktail_k0 = (KT - 1) * GBK + lch * 4;
  const bool tail_out = ktail && ktail_k0 >= K;
  const int tail_keep = (ktail && ktail_k0 < K && ktail_k0 + 4 > K) ? (int)(K - ktail_k0) : 4;
  auto load_mask = [&](int kt) __attribute__((always_inline)) -> unsigned {
    return (kt >= KT || (tail_out && kt == KT - 1)) ? GOOR : 0u;
  };
  f32x4 ra[8], rb[8];
  // XCOL: row N - 1 of B, this lane's chunk: rx belongs to the stage held in ra, rxn to the one being loaded
  const srd_t xsrd = make_srd(B + N * ldb);
  f32x4 rx = {0.f, 0.f, 0.f, 0.f}, rxn = rx;
  float xacc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xacc[i] = 0.f;

  // ---- staging: 4 consecutive k of one row = 8 bytes of bf16, chunk (k / 8) ^ ((row >> 2) & 3) of the row
  const int soff = lrow * GROWB + (((lch >> 1) ^ ((tid >> 5) & 3)) << 4) + ((lch & 1) << 3);
  auto stage_a = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(ra[i][0], ra[i][1], h0, l0);
    split2n(ra[i][2], ra[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + soff + i * 32 * GROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + GARR + soff + i * 32 * GROWB) = u32x2{l0, l1};
  };
  auto stage_b = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
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
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // One stage = two k-steps of 48 MFMAs.  Per k-step: P1 = a_lo x b_hi, P2 = a_hi x b_lo, P3 = a_hi x b_hi over the
  // 16 accumulators.  Fragment schedule: the k-step's a_lo / b_hi are in registers when it starts; a_hi / b_lo are
  // requested during P1, the NEXT k-step's a_lo / b_hi during P2 (into the registers a_lo frees and 16 new ones), its
  // a_hi / b_lo during P3.  Staging pieces (16 per stage) sit in every sixth slot.
  auto body = [&](const char* cur, char* nxt, int kt_load) __attribute__((always_inline)) {
    bf16x8 al[4], bh[4], ah[4], bl[4], al1[4], bh1[4], ah1[4], bl1[4];
    const int fo0 = fsw, fo1 = fsw ^ 32;
    const int so = kt_load * GBK * 4;
    const unsigned tmask = load_mask(kt_load);
    if constexpr (XCOL) rxn = bload4(xsrd, (unsigned)(lch * 16) | tmask, so);
#define SB __builtin_amdgcn_sched_barrier(0)
#define FRAG(dst, base, i, fo) dst[i] = *reinterpret_cast<const bf16x8*>(cur + (base) + (i) * 32 * GROWB + (fo))
#define MM(x, y, i, j) if (PROBE & 16) acc[i][j][0] += (float)x[i][0] + (float)y[j][1]; else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[i][j], 0, 0, 0)
    // a staged piece in three slots (with one wave per SIMD a 16-instruction block between two MFMAs leaves the
    // matrix pipe idle): split the first pair, split the second pair, write both planes + reload
    unsigned sh0 = 0, sl0 = 0, sh1 = 0, sl1 = 0;
#define SPL(a_, b_, h_, l_) if (PROBE & 32) split2n_trunc(a_, b_, h_, l_); else split2n(a_, b_, h_, l_)
#define SA1(i) if constexpr (XCOL) xacc[i] = fmaf(ra[i][3], rx[3], fmaf(ra[i][2], rx[2], fmaf(ra[i][1], rx[1], fmaf(ra[i][0], rx[0], xacc[i])))); \
               if (!(PROBE & 4)) SPL(ra[i][0], ra[i][1], sh0, sl0)
#define SA2(i) if (!(PROBE & 4)) SPL(ra[i][2], ra[i][3], sh1, sl1)
#define SA3(i) if (!(PROBE & 4)) { *reinterpret_cast<u32x2*>(nxt + soff + i * 32 * GROWB) = u32x2{sh0, sh1};        \
               *reinterpret_cast<u32x2*>(nxt + GARR + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               if (!(PROBE & 2)) ra[i] = bload4(asrd, aoffs[i] | tmask, so)
#define SB1(i) if (!(PROBE & 4)) SPL(rb[i][0], rb[i][1], sh0, sl0)
#define SB2(i) if (!(PROBE & 4)) SPL(rb[i][2], rb[i][3], sh1, sl1)
#define SB3(i) if (!(PROBE & 4)) { *reinterpret_cast<u32x2*>(nxt + 2 * GARR + soff + i * 32 * GROWB) = u32x2{sh0, sh1}; \
               *reinterpret_cast<u32x2*>(nxt + 3 * GARR + soff + i * 32 * GROWB) = u32x2{sl0, sl1}; } \
               if (!(PROBE & 2)) rb[i] = bload4(bsrd, boffs[i] | tmask, so)
#define MMZ(x, y, i, j) MM(x, y, i, j)
#include "gemm_bf16x3_big_schedule.inc"
#undef MMZ
    if constexpr (XCOL) rx = rxn;
#undef SPL
#undef SB3
#undef SB2
#undef SB1
#undef SA3
#undef SA2
#undef SA1
#undef MM
#undef FRAG
#undef SB
  };

  // ---- prologue: stage 0 -> LDS, stage 1 -> registers
  {
    const unsigned t0 = load_mask(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = bload4(asrd, aoffs[i] | t0, 0); rb[i] = bload4(bsrd, boffs[i] | t0, 0); }
    if constexpr (XCOL) {
      rx = bload4(xsrd, (unsigned)(lch * 16) | t0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xacc[i] = fmaf(ra[i][3], rx[3], fmaf(ra[i][2], rx[2], fmaf(ra[i][1], rx[1], fmaf(ra[i][0], rx[0], xacc[i]))));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { stage_a(lds, i); stage_b(lds, i); }
    if (ktail && KT == 1) fix_tail(lds);
    const unsigned t1 = load_mask(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { ra[i] = bload4(asrd, aoffs[i] | t1, GBK * 4); rb[i] = bload4(bsrd, boffs[i] | t1, GBK * 4); }
    if constexpr (XCOL) rx = bload4(xsrd, (unsigned)(lch * 16) | t1, GBK * 4);
  }
  __syncthreads();
  int par = 0;
  for (int kt = 0; kt < KT; ++kt) {
    char* nxt = lds + (par ^ 1) * GSTAGE;
    body(lds + par * GSTAGE, nxt, kt + 2);        // stores stage kt + 1 (registers), loads stage kt + 2
    if (ktail && kt + 1 == KT - 1) fix_tail(nxt);
    if (!(PROBE & 1)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    par ^= 1;
  }

  if (XCOL && nt == tmap.NT - 1) {
    // column N of the full matrix: the 8 lanes of a row hold its eight 16-byte k chunks (same summation tree for
    // every row; exact fp32 products, not bit-comparable with the MFMA columns' split arithmetic -- like the 8-wave
    // kernel's extra column)
    const int64_t n = N;
    const float bv = bias ? bias[n] : 0.f;
    const int64_t cq = sm.remap ? n / sm.cm : 0, cr = sm.remap ? n - cq * sm.cm : n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = xacc[i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      const int64_t m = m0 + lrow + 32 * i;
      if (lch == 0 && m < M) {
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
        if (act == 2) { const float y = sm.aux[m * sm.ldaux + n]; v *= 1.f - y * y; }
        if (accumulate) v += C[a];
        C[a] = v;
      }
    }
  }
  // ---- epilogue: four 64 x 64 blocks per wave through a private 17-KB scratch in the (now free) stage memory
  static_assert(4 * 64 * EPITCH * 4 <= 2 * GSTAGE, "epilogue scratch must fit in the stages");
  if ((PROBE & 8) && K >= 0) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][7];
    if (sacc == 123.456f) C[tid] = sacc;
    return;
  }
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
      const int64_t mr = m0 + (int64_t)wm * 128 + ih * 64, nc = n0 + (int64_t)wn * 128 + jh * 64;
      if (!sm.remap)
        gemm_epilogue_rows(a2, stage, C, M, N, mr, nc, lane, bias, act, accumulate, sm.ldc, true, sm.aux, sm.ldaux);
      else if (remap_vec_ok(sm, C))
        gemm_epilogue_rows_remap_vec(a2, stage, C, M, N, mr, nc, lane, bias, act, accumulate, sm);
      else if (remap_wide_ok(sm))
        gemm_epilogue_rows_remap_wide(a2, stage, C, M, N, mr, nc, lane, bias, act, accumulate, sm);
      else
        gemm_epilogue_rows_remap(a2, stage, C, M, N, mr, nc, lane, bias, act, accumulate, sm);
    }
}

}  // namespace

int tssep_gemm_bf16x3_big_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (g->a_kmajor || g->b_kmajor || g->splitk > 1 || g->kperiod > 0 || g->b_ones_col) return TSSEP_E_UNSUPPORTED;
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B)) return TSSEP_E_UNSUPPORTED;
  if (g->M < 4 * GM || g->K < 1) return TSSEP_E_UNSUPPORTED;
  // 32-bit buffer offsets: one tile's rows and the whole K extent must stay below 2 GB
  if ((int64_t)GM * g->lda * 4 + g->K * 4 >= (int64_t)1 << 31 || (int64_t)GN * g->ldb * 4 + g->K * 4 >= (int64_t)1 << 31)
    return TSSEP_E_UNSUPPORTED;
  const TileMap tm = make_tile_map((g->M + GM - 1) / GM, (g->N + GN - 1) / GN, 1);
  if (g->N > 256 && g->N % 256 == 1) {        // 256 q + 1 columns: q tiles + one VALU column
    if (g->K & 3) return TSSEP_E_UNSUPPORTED;
    if (call.dry) return TSSEP_OK;
    const TileMap tmx = make_tile_map((g->M + GM - 1) / GM, (g->N - 1) / GN, 1);
    hipLaunchKernelGGL((gemm_bf16x3_big_kernel<0, true>), dim3((unsigned)tile_map_blocks(tmx)), dim3(GNT), 0, (hipStream_t)stream,
                       g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act, g->accumulate, sm, tmx);
    return tssep_launch_status();
  }
  if (call.dry) return TSSEP_OK;
#define GLAUNCH(P_) hipLaunchKernelGGL((gemm_bf16x3_big_kernel<P_, false>), dim3((unsigned)tile_map_blocks(tm)), dim3(GNT), 0, (hipStream_t)stream, \
                     g->A, g->B, g->C, g->M, g->N, g->K, g->lda, g->ldb, g->bias, g->act, g->accumulate, sm, tm)
#ifdef TSSEP_GEMM_EXP
  {
    const char* pe = getenv("TSSEP_BIG_PROBE");
    switch (pe ? atoi(pe) : 0) {
      case 1: GLAUNCH(1); return tssep_launch_status();
      case 2: GLAUNCH(2); return tssep_launch_status();
      case 4: GLAUNCH(4); return tssep_launch_status();
      case 6: GLAUNCH(6); return tssep_launch_status();
      case 8: GLAUNCH(8); return tssep_launch_status();
      case 14: GLAUNCH(14); return tssep_launch_status();
      case 15: GLAUNCH(15); return tssep_launch_status();
      case 16: GLAUNCH(16); return tssep_launch_status();
      case 30: GLAUNCH(30); return tssep_launch_status();
      case 32: GLAUNCH(32); return tssep_launch_status();
      case 34: GLAUNCH(34); return tssep_launch_status();
      default: break;
    }
  }
#endif
  GLAUNCH(0);
#undef GLAUNCH
  return tssep_launch_status();
}
