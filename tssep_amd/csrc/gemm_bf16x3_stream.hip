// Persistent "streaming" variant of the split-bf16 GEMM for row-major x row-major operands (both k-contiguous):
// every nn.Linear / LSTM input projection forward and its d(input) GEMM (tssep/train/rnnp.py:88-96,146-161,
// tssep/train/net.py:663-666) whose store is a plain row-major C.
//
// Why: timing probes of the 256 x 256 tile (profiles/r3_gemm_tile_life_probes.jsonl) showed that the phases of a
// tile's life ADD instead of overlapping -- at M = 777 216, N = 2400, K = 320: staging 1.39 ms + matrix work 0.7 +
// waiting for loads 1.07 + C stores 1.67 = the 4.84 ms of the whole GEMM.  The stores are the largest piece: a
// workgroup cannot start its next tile before its 128 accumulator registers per lane have left, all 256 CUs hit
// that phase with 64 MB at once, and a CU whose store queue is backed up to HBM also stalls the LOADS of whatever
// else is resident on it (one vector-memory pipeline per CU), so two workgroups per CU do not hide it either.
//
// Design:
//  * persistent grid, one 512-thread workgroup per CU, walking the XCD-aware tile list; ONE continuous software
//    pipeline over (tile, k-stage): the loads of the next tile's first stages are in flight while the current
//    tile finishes (no prologue / drain bubble per tile);
//  * 256 x 128 output tile, 8 waves as 4 x 2, wave tile 64 x 64 = 2 x 2 MFMA tiles: 64 accumulator registers --
//    so that TWO banks fit: a tile accumulates into one bank while the previous tile's bank is DRAINED in four
//    paced pieces (one 32 x 32 sub-tile each, a quarter of the next tile's K loop apart): transposed through a
//    4.5-KB per-wave LDS scratch and stored as 16 B per lane, 8 rows x 128 B per wave instruction.  The write stream
//    to HBM is smooth (7.46 GB over the whole kernel instead of bursts of 64 MB) and never backs up the pipeline;
//  * K staged 32 at a time (whole 128-byte lines per row and load; 24 MFMAs per wave between barriers like the
//    256 x 128 x 16 tile), bf16 hi / lo rows of 64 B in LDS, 16-byte chunks XOR-swizzled with (row >> 2) & 3 instead
//    of padded: the fragment reads (ds_read_b128, lane = row) and the staging writes (ds_write_b64) are
//    conflict-free, two stages = 96 KB;
//  * same k order per output element and same epilogue arithmetic as the other split-bf16 kernels: bit-identical.
#include <cstdlib>
#include <type_traits>
#include "gemm_common.h"

namespace {

using namespace gemm_detail;

constexpr int SM = 256, SN = 128, SBK = 32, SNT = 512;
constexpr int SROWB = 64;                                   // bytes per LDS row: 32 bf16
constexpr int SARR_A = SM * SROWB, SARR_B = SN * SROWB;     // 16 384, 8 192
constexpr int SSTAGE = 2 * SARR_A + 2 * SARR_B;             // A hi, A lo, B hi, B lo = 49 152 B
constexpr int DPF = 36;                                     // floats per row of a wave's 32 x 32 drain scratch
constexpr int DSCR = 32 * DPF * 4;                          // 4 608 B per wave
constexpr int SBIAS = 4096;                                 // floats of bias kept in LDS (N beyond that: no stream kernel)
constexpr unsigned OOR = 0x80000000u;                       // buffer offset beyond the range: the load returns 0

// PROBE (experiment builds only, -DTSSEP_GEMM_EXP; results are garbage, TIMING probes): 1 = no barriers, 2 = no global
// loads, 4 = no staging (split + LDS writes), 8 = no drain (C never stored), 16 = no MFMAs
template <int PROBE>
__global__ __launch_bounds__(SNT, 2) void gemm_bf16x3_stream_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int64_t M, int64_t N,
    int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, TileMap tmap, int64_t nids) {
  __shared__ __attribute__((aligned(16))) char lds[2 * SSTAGE + 8 * DSCR];
  // the bias vector, read by the drain from LDS: a GLOBAL load there would sit in the in-order memory counter behind
  // the prefetched operand tiles -- and the compiler then waits for that counter at the top of EVERY stage
  __shared__ __attribute__((aligned(16))) float bias_s[SBIAS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (wave-uniform by construction; through readfirstlane so that everything derived from them -- the drain's buffer
  // resource above all -- lives in SGPRs: a resource in VGPRs costs a waterfall loop around every store)
  const int wm = __builtin_amdgcn_readfirstlane(wave >> 1), wn = __builtin_amdgcn_readfirstlane(wave & 1);
  const int KT = (int)((K + SBK - 1) / SBK);
  const bool ktail = (K % SBK) != 0;
  const int64_t G = gridDim.x;
  for (int i = tid; i < SBIAS; i += SNT) bias_s[i] = (bias && i < N) ? bias[i] : 0.f;      // (visible after the prologue's barrier)

  // ---- tile iterators: the loader runs two stages ahead of the MFMAs over the same list
  auto next_tile = [&](int64_t& id, int& mt, int& nt) __attribute__((always_inline)) -> bool {
    for (;;) {
      id += G;
      if (id >= nids) return false;
      int z;
      if (tile_map_decode(tmap, id, mt, nt, z)) return true;
    }
  };
  int64_t l_id = (int64_t)blockIdx.x - G, c_id = l_id;      // (both iterators walk ids blockIdx.x, + G, + 2 G, ...)
  int l_mt = 0, l_nt = 0, c_mt = 0, c_nt = 0, l_kt = 0;
  bool l_valid = next_tile(l_id, l_mt, l_nt);
  bool c_valid = next_tile(c_id, c_mt, c_nt);
  if (!c_valid) return;

  // ---- loads: lane <-> (row tid / 8 + 64 i, 16-byte chunk tid % 8 of the row's 128-byte K slice).
  // aoffs / boffs: the tile's row offsets; la / lb: what the NEXT load uses (chunks at or beyond K in the last,
  // partial K stage of a tile read offset OOR = zero; a chunk that straddles K is fixed up in LDS, see below)
  const int lrow = tid >> 3, lch = tid & 7;
  unsigned aoffs[4], boffs[2], tmask = 0u;     // tmask: 0x80000000 on lanes whose chunk of the NEXT load lies beyond K
  srd_t asrd = make_srd(A), bsrd = make_srd(B);
  const int ktail_k0 = (KT - 1) * SBK + lch * 4;             // this lane's first k in the last stage
  const bool tail_out = ktail && ktail_k0 >= K;              // its chunk lies beyond K there
  const int tail_keep = (ktail && ktail_k0 < K && ktail_k0 + 4 > K) ? (int)(K - ktail_k0) : 4;   // 1..3: straddles K
  auto set_next_load = [&]() __attribute__((always_inline)) {
    tmask = (tail_out && l_kt == KT - 1) ? OOR : 0u;      // (offsets stay below 2 GB: OR-ing the bit puts them out of range)
  };
  auto set_tile_loads = [&]() __attribute__((always_inline)) {
    const int64_t m0 = (int64_t)l_mt * SM, n0 = (int64_t)l_nt * SN;
    asrd = make_srd(A + m0 * lda);
    bsrd = make_srd(B + n0 * ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t r = m0 + lrow + 64 * i;
      r = r > M - 1 ? M - 1 : r;
      aoffs[i] = l_valid ? (unsigned)(((r - m0) * lda + lch * 4) * 4) : OOR;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t r = n0 + lrow + 64 * i;
      r = r > N - 1 ? N - 1 : r;
      boffs[i] = l_valid ? (unsigned)(((r - n0) * ldb + lch * 4) * 4) : OOR;
    }
  };
  set_tile_loads();
  set_next_load();
  f32x4 ra[4], rb[2];
  bool ld_tail = false, st_tail = false;       // the stage being loaded / held in registers is a partial K stage
  auto gload = [&]() __attribute__((always_inline)) {          // (prologue; the stage body issues its loads itself)
    const int so = l_kt * SBK * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = bload4(asrd, aoffs[i] | tmask, so);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = bload4(bsrd, boffs[i] | tmask, so);
    ld_tail = ktail && l_valid && l_kt == KT - 1;
  };
  auto advance_loader = [&]() __attribute__((always_inline)) {
    if (l_valid) {
      if (++l_kt == KT) {
        l_kt = 0;
        l_valid = next_tile(l_id, l_mt, l_nt);
        set_tile_loads();
      }
    }
    set_next_load();
  };

  // ---- staging: 4 consecutive k of one row = 8 bytes of bf16, chunk (k / 8) ^ ((row >> 2) & 3) of the row
  const int soff = lrow * SROWB + (((lch >> 1) ^ ((tid >> 5) & 3)) << 4) + ((lch & 1) << 3);
  auto stage_a = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(ra[i][0], ra[i][1], h0, l0);
    split2n(ra[i][2], ra[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + soff + i * 64 * SROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + SARR_A + soff + i * 64 * SROWB) = u32x2{l0, l1};
  };
  auto stage_b = [&](char* st, int i) __attribute__((always_inline)) {
    unsigned h0, l0, h1, l1;
    split2n(rb[i][0], rb[i][1], h0, l0);
    split2n(rb[i][2], rb[i][3], h1, l1);
    *reinterpret_cast<u32x2*>(st + 2 * SARR_A + soff + i * 64 * SROWB) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(st + 2 * SARR_A + SARR_B + soff + i * 64 * SROWB) = u32x2{l0, l1};
  };
  auto sstore = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_a(st, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) stage_b(st, i);
  };
  // a chunk that straddles K (K % 4 != 0) was staged with whatever lies in the row's padding: zero those bf16
  auto fix_tail = [&](char* st) __attribute__((always_inline)) {
    if (tail_keep < 4) {
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        if (e >= tail_keep) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<unsigned short*>(st + soff + i * 64 * SROWB + 2 * e) = 0;
            *reinterpret_cast<unsigned short*>(st + SARR_A + soff + i * 64 * SROWB + 2 * e) = 0;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<unsigned short*>(st + 2 * SARR_A + soff + i * 64 * SROWB + 2 * e) = 0;
            *reinterpret_cast<unsigned short*>(st + 2 * SARR_A + SARR_B + soff + i * 64 * SROWB + 2 * e) = 0;
          }
        }
      }
    }
  };

  // ---- fragments: lane = row (lane & 31), 8 consecutive k = chunk 2 ks + (lane >> 5), swizzled as above
  const int fsw = ((lane >> 5) ^ ((lane >> 2) & 3)) << 4;            // k-step 0; k-step 1 = fsw ^ 32
  const int aoff = (wm * 64 + (lane & 31)) * SROWB, boff = 2 * SARR_A + (wn * 64 + (lane & 31)) * SROWB;
  f32x16 acc[2][2][2];                                               // [bank][i][j]

  // One stage, written slot by slot (sched_barrier pins the order: the scheduler's own choice put MFMAs on one
  // accumulator back to back and the loads right in front of their use): 24 MFMAs out of `cur`; the fragments of
  // k-step 1 requested while k-step 0 computes (<= 48 fragment registers live beside the 128 accumulators); from
  // the 9th MFMA on, one staged 16-byte piece every two to three MFMAs = split, write to `nxt`, and the load of the
  // same piece of the stage after next into the registers just freed -- ~22 MFMA slots before it is needed.
  float* scr = reinterpret_cast<float*>(lds + 2 * SSTAGE + wave * DSCR);      // this wave's 32 x 32 drain scratch
  int64_t p_m0 = 0, p_n0 = 0;       // origin of the tile whose bank is being drained
  int p_left = 0;                   // sub-tiles of it still in registers
  // DQ >= 0: the stage also drains sub-tile DQ = 2 i + j of the OTHER bank (the previous tile) between its MFMAs:
  // 16 ds_write_b32 into the wave's scratch in the first eight slots, then four times {read a row group back as
  // 16 B per lane, add the bias, buffer_store} -- as a separate block behind the barrier the same instructions ran
  // with the matrix pipe idle in all 8 waves at once and cost as much as the epilogue they replace (measured).
  auto body = [&](auto bank_tag, auto dq_tag, bool pending, const char* cur, char* nxt) __attribute__((always_inline)) {
    constexpr int BANK = decltype(bank_tag)::value;
    constexpr int DQ = decltype(dq_tag)::value;
    constexpr int DI = DQ >= 0 ? (DQ >> 1) : 0, DJ = DQ >= 0 ? (DQ & 1) : 0;
    bf16x8 ah[2], al[2], bh[2], bl[2], ah1[2], al1[2], bh1[2], bl1[2];
    const int fo0 = fsw, fo1 = fsw ^ 32;
    const int so = l_kt * SBK * 4;
    // drain: C sub-tile origin as a buffer resource (scalar), per-lane offset (row r0, columns c4 .. c4 + 3)
    f32x4 dv[2], dbv;
    srd_t csrd = asrd;
    unsigned dvo = 0u;
    int64_t dm = 0;
    if constexpr (DQ >= 0) {
      const int64_t mb = p_m0 + wm * 64 + DI * 32, nb = p_n0 + wn * 64 + DJ * 32;
      csrd = make_srd(C + mb * ldc + nb);
      dm = mb + (lane >> 3);
      dvo = (pending && !(PROBE & 8) && nb + (lane & 7) * 4 < N) ? (unsigned)(((lane >> 3) * ldc + (lane & 7) * 4) * 4) : OOR;
    }
#define DW(e) if constexpr (DQ >= 0) scr[(((e) & 3) + 8 * ((e) >> 2) + 4 * (lane >> 5)) * DPF + (lane & 31)] = acc[1 - BANK][DI][DJ][e]
#define DBIAS if constexpr (DQ >= 0) { const int64_t n_ = p_n0 + wn * 64 + DJ * 32 + (lane & 7) * 4; dbv = *reinterpret_cast<const f32x4*>(bias_s + (n_ < SBIAS - 3 ? n_ : 0)); }
#define DR(it) if constexpr (DQ >= 0) dv[(it) & 1] = *reinterpret_cast<const f32x4*>(scr + ((it) * 8 + (lane >> 3)) * DPF + (lane & 7) * 4)
// (the row-group offset goes into the per-lane offset, soffset stays the immediate 0: for a store of more than 64 bits
// whose soffset is an SGPR the compiler inserts NO wait state in front of a VALU write of the data registers -- LLVM's
// hazard recognizer exempts that form -- and on gfx950 the store was seen to read data the next instruction had already
// overwritten: the integer offset of the next operand load reached C as a denormal in lanes 12-15 of every 16, ~100
// elements of 50 M, only where a workgroup walks several tiles -- round 4, tools/scan_store_hazard.py guards the build)
#define DS(it) if constexpr (DQ >= 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dv[(it) & 1] + dbv), csrd, \
                   (int)((dm + (it) * 8 < M) ? dvo + (unsigned)((it) * 8 * ldc * 4) : OOR), 0, 2)
#define SB __builtin_amdgcn_sched_barrier(0)
#define FRAG(dst, base, i, fo) dst[i] = *reinterpret_cast<const bf16x8*>(cur + (base) + (i) * 32 * SROWB + (fo))
#define MM(x, y, i, j) if (PROBE & 16) acc[BANK][i][j][0] += (float)x[i][0] + (float)y[j][1]; else acc[BANK][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[BANK][i][j], 0, 0, 0)
#define STA(i) if (!(PROBE & 4)) stage_a(nxt, i); if (!(PROBE & 2)) ra[i] = bload4(asrd, aoffs[i] | tmask, so)
#define STB(i) if (!(PROBE & 4)) stage_b(nxt, i); if (!(PROBE & 2)) rb[i] = bload4(bsrd, boffs[i] | tmask, so)
    FRAG(al, SARR_A + aoff, 0, fo0); FRAG(bh, boff, 0, fo0); FRAG(bh, boff, 1, fo0); FRAG(al, SARR_A + aoff, 1, fo0);
    FRAG(ah, aoff, 0, fo0); FRAG(bl, SARR_B + boff, 0, fo0); FRAG(bl, SARR_B + boff, 1, fo0); FRAG(ah, aoff, 1, fo0); SB;
    MM(al, bh, 0, 0); FRAG(al1, SARR_A + aoff, 0, fo1); DW(0); DW(1); SB;
    MM(al, bh, 0, 1); FRAG(bh1, boff, 0, fo1); DW(2); DW(3); SB;
    MM(al, bh, 1, 0); FRAG(bh1, boff, 1, fo1); DW(4); DW(5); SB;
    MM(al, bh, 1, 1); FRAG(al1, SARR_A + aoff, 1, fo1); DW(6); DW(7); SB;
    MM(ah, bl, 0, 0); DW(8); DW(9); SB;
    MM(ah, bl, 0, 1); DW(10); DW(11); SB;
    MM(ah, bl, 1, 0); STA(0); DW(12); DW(13); SB;
    MM(ah, bl, 1, 1); DW(14); DW(15); SB;
    MM(ah, bh, 0, 0); FRAG(ah1, aoff, 0, fo1); DBIAS; SB;
    MM(ah, bh, 0, 1); FRAG(bl1, SARR_B + boff, 0, fo1); SB;
    MM(ah, bh, 1, 0); FRAG(bl1, SARR_B + boff, 1, fo1); STA(1); SB;
    MM(ah, bh, 1, 1); FRAG(ah1, aoff, 1, fo1); DR(0); SB;
    MM(al1, bh1, 0, 0); STA(2); SB;
    MM(al1, bh1, 0, 1); DR(1); DS(0); SB;
    MM(al1, bh1, 1, 0); SB;
    MM(al1, bh1, 1, 1); STA(3); SB;
    MM(ah1, bl1, 0, 0); DS(1); DR(2); SB;
    MM(ah1, bl1, 0, 1); STB(0); SB;
    MM(ah1, bl1, 1, 0); DS(2); DR(3); SB;
    MM(ah1, bl1, 1, 1); SB;
    MM(ah1, bh1, 0, 0); STB(1); SB;
    MM(ah1, bh1, 0, 1); DS(3); SB;
    MM(ah1, bh1, 1, 0); SB;
    MM(ah1, bh1, 1, 1); SB;
#undef DS
#undef DR
#undef DBIAS
#undef DW
#undef STB
#undef STA
#undef MM
#undef FRAG
#undef SB
  };

  // ---- drain of one 32 x 32 sub-tile (i, j) of a finished bank: D layout -> rows through the wave's scratch
  // the same drain as a block of its own: what is left when a tile has fewer than four stages, and the last tile
  auto drain_sub = [&](const f32x16& t, int i, int j) __attribute__((always_inline)) {
    if (PROBE & 8) { if (t[0] == 1.25f) C[lane] = t[3]; return; }
    const int col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * half) * DPF + col] = t[e];
    // (same wave writes and reads: LDS operations of one wave complete in order)
    const int c4 = (lane & 7) * 4, r0 = lane >> 3;
    const int64_t mb = p_m0 + wm * 64 + i * 32, nb = p_n0 + wn * 64 + j * 32;
    const int64_t n = nb + c4;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (n < SBIAS - 3 ? n : 0));      // (zeros beyond N)
    const srd_t csrd = make_srd(C + mb * ldc + nb);
    const unsigned vo = n < N ? (unsigned)((r0 * ldc + c4) * 4) : OOR;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {       // (not unrolled: the loop around it is at the register ceiling)
      const f32x4 v = *reinterpret_cast<const f32x4*>(scr + (it * 8 + r0) * DPF + c4) + bv;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), csrd,
                                             (int)((mb + it * 8 + r0 < M) ? vo + (unsigned)(it * 8 * ldc * 4) : OOR), 0, 2);
    }
  };
  auto drain_one = [&](auto bank_tag) __attribute__((always_inline)) {         // the next pending sub-tile of bank BANK (register indices static)
    constexpr int BANK = decltype(bank_tag)::value;
    switch (4 - p_left) {
      case 0: drain_sub(acc[BANK][0][0], 0, 0); break;
      case 1: drain_sub(acc[BANK][0][1], 0, 1); break;
      case 2: drain_sub(acc[BANK][1][0], 1, 0); break;
      default: drain_sub(acc[BANK][1][1], 1, 1); break;
    }
    --p_left;
  };

  // ---- prologue of the stream: stage 0 -> LDS, stage 1 -> registers
  int par = 0;
  gload();
  st_tail = ld_tail;
  advance_loader();
  sstore(lds);
  if (st_tail) fix_tail(lds);
  gload();
  st_tail = ld_tail;
  advance_loader();
  __syncthreads();

  // one tile: KT stages into bank BANK while bank 1 - BANK (the previous tile) leaves in four paced pieces
  auto run_tile = [&](auto bank_tag) __attribute__((always_inline)) {
    constexpr int BANK = decltype(bank_tag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[BANK][i][j][e] = 0.f;
    // Four segments of KT / 4 stages: the first stage of segment q drains sub-tile q of the other bank between its
    // MFMAs (a switch between body variants inside ONE loop made the register allocator spill ~800 values; as
    // straight-line segments each body copy is a loop of its own).  With nothing pending (first tile) the drain's
    // stores are sent out of range.
    const bool pending = p_left > 0;
    auto stage = [&](auto dq_tag) __attribute__((always_inline)) {
      char* nxt = lds + (par ^ 1) * SSTAGE;
      body(bank_tag, dq_tag, pending, lds + par * SSTAGE, nxt);      // stores the stage in registers, loads the one after it
      if (st_tail) fix_tail(nxt);
      if (!(PROBE & 1)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      par ^= 1;
      st_tail = ktail && l_valid && l_kt == KT - 1;
      advance_loader();
    };
    int kt = 0;
    stage(std::integral_constant<int, 0>{});
    for (kt = 1; kt * 4 < KT; ++kt) stage(std::integral_constant<int, -1>{});
    stage(std::integral_constant<int, 1>{});
    for (++kt; kt * 4 < 2 * KT; ++kt) stage(std::integral_constant<int, -1>{});
    stage(std::integral_constant<int, 2>{});
    for (++kt; kt * 4 < 3 * KT; ++kt) stage(std::integral_constant<int, -1>{});
    stage(std::integral_constant<int, 3>{});
    for (++kt; kt < KT; ++kt) stage(std::integral_constant<int, -1>{});
    p_left = 0;
    p_m0 = (int64_t)c_mt * SM;
    p_n0 = (int64_t)c_nt * SN;
    p_left = 4;
    c_valid = next_tile(c_id, c_mt, c_nt);
  };
  for (;;) {
    run_tile(std::integral_constant<int, 0>{});
    if (!c_valid) { while (p_left > 0) drain_one(std::integral_constant<int, 0>{}); break; }
    run_tile(std::integral_constant<int, 1>{});
    if (!c_valid) { while (p_left > 0) drain_one(std::integral_constant<int, 1>{}); break; }
  }
}
}  // namespace

int tssep_gemm_bf16x3_stream_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call) {
  using namespace gemm_detail;
  void* const stream = call.stream;
  if (g->a_kmajor || g->b_kmajor || sm.remap || g->splitk > 1 || g->kperiod > 0 || g->b_ones_col) return TSSEP_E_UNSUPPORTED;
  // plain stores + bias only (the GEMMs with a Tanh / its backward / an accumulate in the store keep the tiled kernels:
  // their extra loads would sit in the drain's way), 16-byte rows of C
  if (g->act != 0 || g->accumulate || g->N > SBIAS || (g->N & 3) || (sm.ldc & 3) || !aligned16(g->C)) return TSSEP_E_UNSUPPORTED;
  if ((int64_t)40 * sm.ldc * 4 >= (int64_t)1 << 31) return TSSEP_E_UNSUPPORTED;      // (tanh stores of the step all go through a remap: not duplicated here)
  if ((g->lda & 3) || (g->ldb & 3) || !aligned16(g->A) || !aligned16(g->B)) return TSSEP_E_UNSUPPORTED;
  if (g->M < 4 * SM || g->K <= 3 * SBK) return TSSEP_E_UNSUPPORTED;      // (>= 4 K stages: one per drained sub-tile)
  // 32-bit buffer offsets: one tile's rows and the whole K extent must stay below 2 GB
  if ((int64_t)SM * g->lda * 4 + g->K * 4 >= (int64_t)1 << 31 || (int64_t)SN * g->ldb * 4 + g->K * 4 >= (int64_t)1 << 31)
    return TSSEP_E_UNSUPPORTED;
  if (call.dry) return TSSEP_OK;
  const TileMap tm = make_tile_map((g->M + SM - 1) / SM, (g->N + SN - 1) / SN, 1);
  const int64_t nids = tile_map_blocks(tm);
  const int ncu = current_device_cus();      // (per device, gemm_common.h)
  const int64_t grid = nids < ncu ? nids : ncu;
#define SLAUNCH(P_) hipLaunchKernelGGL(gemm_bf16x3_stream_kernel<P_>, dim3((unsigned)grid), dim3(SNT), 0, (hipStream_t)stream, g->A, g->B, \
                     g->C, g->M, g->N, g->K, g->lda, g->ldb, sm.ldc, g->bias, tm, nids)
#ifdef TSSEP_GEMM_EXP
  {
    const char* pe = getenv("TSSEP_STREAM_PROBE");
    switch (pe ? atoi(pe) : 0) {
      case 1: SLAUNCH(1); return tssep_launch_status();
      case 2: SLAUNCH(2); return tssep_launch_status();
      case 4: SLAUNCH(4); return tssep_launch_status();
      case 6: SLAUNCH(6); return tssep_launch_status();
      case 8: SLAUNCH(8); return tssep_launch_status();
      case 10: SLAUNCH(10); return tssep_launch_status();
      case 14: SLAUNCH(14); return tssep_launch_status();
      case 15: SLAUNCH(15); return tssep_launch_status();
      case 16: SLAUNCH(16); return tssep_launch_status();
      case 24: SLAUNCH(24); return tssep_launch_status();
      case 30: SLAUNCH(30); return tssep_launch_status();
      default: break;
    }
  }
#endif
  SLAUNCH(0);
#undef SLAUNCH
  return tssep_launch_status();
}
