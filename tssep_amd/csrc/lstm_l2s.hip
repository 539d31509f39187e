// EXPERIMENT BUILD ONLY (`make exp`; not in libtssep_hip.so, not in the ABI).  Measured in round 6 and NOT built into the
// product: profiles/r6_l2s_probe.jsonl, DESIGN.md 4.2 -- forward kernels v2 (four waves) and v3 (eight waves, K split
// between the two waves of a SIMD), both parity-green against the exact-fp32 kernel; compute core 14.6-15.2 us per step,
// with the activation stream 37-40 (W-stationary production kernel: 21.4).
//
// Sequence-parallel BLSTM recurrence on the bf16 matrix cores with W_hh STREAMED from the XCD's L2 every step
// (round 6; VERDICT r5 "Next round" #1).  Replaces the T-sequential part of torch.nn.LSTM as the reference uses it
// (tssep/train/rnnp.py:88-95,146-153: one bidirectional layer, batch_first, zero initial state) for launches with
// enough sequences to fill the chip WITHOUT any exchange between workgroups:
//
//   * one workgroup (4 waves, one per SIMD) owns 32 sequences of ONE direction for the whole launch: no peers, no
//     tags, no spin, no error flag, no co-residency contract -- other kernels (weight gradients, RCCL) may run beside it;
//   * W_hh lives in HBM/L2 as packed split-bf16 MFMA A-fragments (hi | lo, 4 bytes per weight, 1.44 MB per direction
//     at H = 300); every step every wave streams the fragments of ITS row tiles straight into registers (16 B per
//     lane, 1 KB per wave and load, each byte used by exactly one wave: LDS would only add a copy).  The ring of
//     KS fragment pairs is refilled right behind the MFMAs that consumed a slot and keeps running across tiles and
//     time steps (the next step re-reads the same planes: they stay in the L2 -- blocks of one direction share an XCD);
//   * h_{t-1} of the 32 sequences sits in LDS as bf16 hi / lo B-fragments (double-buffered, one barrier per step) and
//     is held in registers for the whole step (KS x 2 x 4 VGPRs);
//   * v_mfma_f32_32x32x16_bf16, D[32 gate rows x 32 sequences]: the tile's rows are ordered so that a lane ends up
//     with the four gate pre-activations of four CONSECUTIVE units of one sequence: the cell update is lane-local and
//     every access to gates / cell / h is one 16-byte access;
//   * product = w_lo h_hi + w_hi h_lo + w_hi h_hi, fp32 accumulate on top of the input projection (x W_ih^T + b,
//     computed beforehand by the GEMMs into `gates`): the arithmetic of the W-stationary kernels (lstm_onchip.hip).
//
// Tensor layouts: those of tssep_blstm_fwd / _bwd (lstm.hip): gates [N T][2 dirs][H][4 gates] (pre-activations in,
// activations out / activations in, d(pre-activations) out), cell [N T][2][H], hout / dhout rows of ldo floats with
// direction d at column d * dstride.
//
// Packed weights (tssep_lstm_pack_l2s), forward: [dir][tile RT][kstep KS][hi, lo][lane 64][8 bf16] with KS = ceil(H/16),
// RT = 2 KS tiles of 8 units; fragment row i (= lane % 32) of tile `tile` is gate i % 4 of unit
// tile * 8 + ((i / 4) % 2) * 4 + i / 8, fragment column 8 (lane / 32) + e is k = 16 kstep + 8 (lane / 32) + e.
#include <type_traits>
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SEQ = 32;      // sequences per workgroup = the N of the MFMA
#ifndef L2S_PROBE
#define L2S_PROBE 0        // timing probes, results wrong by construction: 1 no MFMAs, 2 no weight loads, 4 no pre-activation loads / stores
#endif
#ifndef L2S_GXMODE
#define L2S_GXMODE 0       // v2: 0 = pre-activations by LDS-DMA (asm, two tiles ahead); 1 = by ordinary coalesced loads into 16 registers per
#endif                     // tile in flight (two tiles ahead), written to the LDS image when the tile starts (probe: profiles/r6_l2s_probe.jsonl)
#ifndef L2S_RING
#define L2S_RING 10       // weight-ring depth in fragment pairs (a divisor of 10 * KS)
#endif

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
  lo = cvt_pk_bf16(a - ha, b - hb);
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// the gate non-linearities of the W-stationary kernels (v_exp_f32 / v_rcp_f32, relative error ~2^-22)
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}

__host__ __device__ inline int l2s_ks(int H) { return (H + 15) / 16; }

// ------------------------------------------------------------------------------------------------ pack
// one thread per packed bf16 PAIR (hi and lo of one weight are written by the same thread)
__global__ void l2s_pack_fwd_kernel(const float* __restrict__ w_hh_f, const float* __restrict__ w_hh_r, int H, int KS,
                                    unsigned short* __restrict__ out) {
  const int RT = 2 * KS;
  const int64_t total = (int64_t)2 * RT * KS * 64 * 8;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int el = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int ks = (int)(r % KS); r /= KS;
    const int tile = (int)(r % RT);
    const int dir = (int)(r / RT);
    const int i = lane & 31;
    const int unit = tile * 8 + ((i >> 2) & 1) * 4 + (i >> 3), gate = i & 3;
    const int k = ks * 16 + 8 * (lane >> 5) + el;
    const float* w = dir ? w_hh_r : w_hh_f;
    const float v = (unit < H && k < H) ? w[(int64_t)(gate * H + unit) * H + k] : 0.f;
    unsigned hi, lo;
    split2(v, 0.f, hi, lo);
    const int64_t base = ((((int64_t)dir * RT + tile) * KS + ks) * 2) * 512 + lane * 8 + el;
    out[base] = (unsigned short)(hi & 0xffffu);
    out[base + 512] = (unsigned short)(lo & 0xffffu);
  }
}

// --------------------------------------------------------------------------------------------- forward
// Buffer addressing throughout: one resource per tensor (base = the workgroup's first sequence), ONE 32-bit lane offset
// per tensor for the whole launch, everything that moves (time step, tile, kstep) in the scalar offset -- with plain
// pointers the compiler hoists the ~400 loop-invariant 64-bit fragment addresses out of the time loop and spills.
// Lanes without work (sequence >= N, units >= H) carry an out-of-range offset: loads return zeros, stores are dropped.
typedef __amdgpu_buffer_rsrc_t srd_t;
constexpr unsigned VOOR = 0x80000000u;
#ifndef L2S_STORE_AUX
#define L2S_STORE_AUX 2
#endif
constexpr int AUX_NT = 2;
__device__ __forceinline__ srd_t make_srd(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 bload(srd_t r, unsigned voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
}
__device__ __forceinline__ f32x4 bload_nt(srd_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, AUX_NT));
}
// (the scalar part of a store's address is added to the LANE offset, soffset stays the immediate 0: a > 64-bit buffer
// store with a register soffset is not guarded against a VALU write of its data registers in the next issue slot --
// tools/scan_store_hazard.py, round 4.  An out-of-range lane offset stays out of range: the sums stay below 2^32.)
__device__ __forceinline__ void bstore_nt(f32x4 v, srd_t r, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(voff + (unsigned)soff), 0, L2S_STORE_AUX);
}

// LDS-DMA copy of 64 x 16 bytes: lane l's 16 bytes at `base + voff` land at LDS address `lds + 16 l`.  Inline assembly:
// the compiler must not know that LDS is written (with the builtin it waits `vmcnt(0)` in front of every LDS read that
// follows, i.e. for the copy just requested two tiles ahead AND, in order, for the whole weight ring); the consumer waits
// explicitly.  The LDS reads of the slot's previous content have returned (lgkmcnt(0)) before the copy is issued.
__device__ __forceinline__ void dma16(const void* base, unsigned voff, unsigned lds) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
}

// Forward.  Wave w owns the NTW CONSECUTIVE row tiles w NTW ... (8 units each); per tile and step:
//   1. the tile's pre-activations (32 sequences x 128 B) leave its LDS slot (filled two tiles earlier by four LDS-DMA
//      instructions, each 8 sequences x 128 contiguous bytes) for 16 registers; the slot is refilled for two tiles ahead;
//   2. KS x 3 MFMAs against the weight ring (refilled R fragment pairs ahead, straight from the L2);
//   3. lane-local cell update; activations, c, h go to a staging image in LDS (swizzled: conflict-free both ways) and
//      the bf16 hi / lo of h into the next step's B fragments;
//   4. the staging image leaves as six coalesced 1-KB stores (128-byte runs per sequence).
// A lane of the MFMA layout touches 16 bytes of 32 different sequences (2.4 MB apart): issued as global accesses these
// run at 0.85 TB/s (probe, round 6: 82 us per step, the weights + MFMAs alone 14.8).
template <int KS, int R>
__global__ __launch_bounds__(256, 1) void blstm_l2s_fwd_kernel(float* __restrict__ gates, float* __restrict__ cell,
                                                               float* __restrict__ hout, int64_t ldo, int64_t dstride,
                                                               const u32x4* __restrict__ wpk, int64_t N, int64_t T, int H) {
  constexpr int RT = 2 * KS, NTW = (RT + 3) / 4, NF = NTW * KS;     // NF fragment pairs per wave and step
  static_assert(NTW % 2 == 0, "the two pre-activation slots alternate by tile parity across steps");
  static_assert(NF % R == 0, "the weight ring must close over one time step");
  static_assert(NTW * 4 - RT <= 2, "one spare B-fragment slot takes the tiles that do not exist");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  constexpr int HB = (KS + 1) * 128;                               // u32x4 per h buffer: [KS + 1 (spare)][hi, lo][64 lanes]
  constexpr int GI = 2 * HB, GO = GI + 4 * 2 * 256;                // gin[wave][slot 2][256], gout[wave][256 + 64 + 64]
  u32x4* hfr = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = lane & 31, hl = lane >> 5;
  // blocks of one direction share an XCD (workgroup b runs on XCD b % 8): ONE direction's planes per L2
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int dir = xcd & 1;
  const int64_t grp = (int64_t)slot * 4 + (xcd >> 1);
  if (grp * SEQ >= N) return;
  const int64_t n0 = grp * SEQ;
  const int G4 = H * 4;                    // floats per (row, direction)

  for (int i = tid; i < 2 * HB; i += 256) hfr[i] = u32x4{0u, 0u, 0u, 0u};

  const srd_t rw = make_srd(wpk + (int64_t)dir * RT * KS * 128);
  const char* gb = reinterpret_cast<const char*>(gates + n0 * T * 2 * G4);
  const srd_t rg = make_srd(gb);
  const srd_t rc = make_srd(cell + n0 * T * 2 * H);
  const srd_t rh = make_srd(hout + n0 * T * ldo);
  const unsigned vw = (unsigned)lane * 16u;
  const int sg_t = 2 * G4 * 4, sc_t = 2 * H * 4, sh_t = (int)ldo * 4;      // bytes per time step

  // ---- io lanes (lane-linear images): gates instruction j covers sequences 8 j ... 8 j + 7, 128 bytes each; a
  // sequence's eight 16-byte pieces (one unit's four gates each) are rotated by seq / 2 inside its row of the image
  unsigned gl[4];                           // byte offset of this lane's piece from (first sequence, t = 0, tile 0) | VOOR: no such sequence
  int gpiece[2];                            // its unit within the tile (instructions 0, 2 / 1, 3)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seq = j * 8 + (lane >> 3), piece = ((lane & 7) - (seq >> 1)) & 7;
    gl[j] = (unsigned)((((int64_t)seq * T * 2 + dir) * G4) * 4 + piece * 16) | ((n0 + seq < N) ? 0u : VOOR);
    gpiece[j & 1] = piece;
  }
  const unsigned gsafe = (unsigned)(dir * G4 * 4);                           // a piece that always exists (first sequence, unit 0)
  const int cseq = lane >> 1, cpiece = ((lane & 1) - (cseq >> 3)) & 1;       // c / h image: 32 bytes per sequence, rotated by seq / 8
  const unsigned cmask = (n0 + cseq < N) ? 0u : VOOR;
  const unsigned cl = (unsigned)((((int64_t)cseq * T * 2 + dir) * H) * 4 + cpiece * 16) | cmask;
  const unsigned hlo = (unsigned)(((int64_t)cseq * T * ldo + dir * dstride) * 4 + cpiece * 16) | cmask;
  // ---- MFMA lanes: sequence s, units hl * 4 + q of the tile
  int swz[4];                               // u32x4 index of piece hl * 4 + q in a gates image
#pragma unroll
  for (int q = 0; q < 4; ++q) swz[q] = s * 8 + ((hl * 4 + q + (s >> 1)) & 7);
  const int cswz = s * 2 + ((hl + (s >> 3)) & 1);
  u32x4* gin = smem + GI + wave * 512;
  u32x4* gout = smem + GO + wave * 384;
  const unsigned gin_lds = lds_addr(gin);

  auto tile_of = [&](int i_) { return wave * NTW + i_; };
  // byte offset of the weight fragments of the i-th tile (tiles that do not exist run on the last real tile's weights
  // with every store out of range and h = 0 in the spare fragment slot: straight-line code, and the barrier waits for
  // the waves with NTW real tiles anyway)
  auto tile_off = [&](int i_) { const int t_ = tile_of(i_); return (t_ < RT ? t_ : RT - 1) * KS * 2048; };
  // four LDS-DMA instructions: tile i_ of time step t_ into slot i_ & 1
  auto gx_dma = [&](int i_, int t_) {
    if (L2S_PROBE & (4 | 16)) return;
    const int tile_ = tile_of(i_);
    const char* base = gb + (int64_t)t_ * sg_t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = !(gl[j] & VOOR) && tile_ * 8 + gpiece[j & 1] < H;
      dma16(base, ok ? gl[j] + (unsigned)(tile_ * 128) : gsafe, gin_lds + (unsigned)(((i_ & 1) * 256 + j * 64) * 16));
    }
  };

  f32x4 gxr[2][4];
  auto gx_regs = [&](int slot_, int i_, int t_) {       // L2S_GXMODE 1: four coalesced 1-KB loads into registers
    const int tile_ = tile_of(i_);
    const unsigned so = (unsigned)(t_ * sg_t + tile_ * 128);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gxr[slot_][j] = bload_nt(rg, (tile_ * 8 + gpiece[j & 1] < H) ? gl[j] + so : VOOR, 0);
  };
  // weight ring: R fragment pairs, refilled R fragments ahead right behind the MFMAs that read a slot; it runs across
  // tiles and across time steps (NF % R == 0: slot f % R of step t + 1 is slot f % R of step t)
  u32x4 wh[R], wl[R];
  {
    const int t0 = dir ? (int)T - 1 : 0;
    if (L2S_GXMODE == 1) { gx_regs(0, 0, t0); gx_regs(1, 1, t0); } else { gx_dma(0, t0); gx_dma(1, t0); }
  }
#pragma unroll
  for (int f = 0; f < R; ++f) {
    const int o = tile_off(f / KS) + (f % KS) * 2048;
    wh[f] = bload(rw, vw, o);
    wl[f] = bload(rw, vw, o + 1024);
  }
  float c[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) c[i][q] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
#pragma clang loop unroll(disable)
  for (int step = 0; step < (int)T; ++step) {
    const int t = dir ? (int)T - 1 - step : step;
    const int tn = step + 1 < (int)T ? (dir ? t - 1 : t + 1) : t;      // (the last step prefetches its own row again)
    // (per-tile fragment offsets the compiler cannot see through: it would otherwise hoist the ~400 loop-invariant
    // offsets out of the time loop into SGPRs it then spills to VGPR lanes)
    int tb[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      tb[i] = tile_off(i);
      asm volatile("" : "+s"(tb[i]));
    }
    bf16x8 bh[KS], bl[KS];
    {
      const u32x4* hb = hfr + cur * HB + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bh[ks] = as_bf16x8(hb[ks * 128]);
        bl[ks] = as_bf16x8(hb[ks * 128 + 64]);
      }
    }
    u32x4* hnext = hfr + (cur ^ 1) * HB;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int tile = tile_of(i);
      // 1. this tile's pre-activations: requested two tiles (>= 64 vector-memory instructions) ago
      f32x4 gq[4];
      if (L2S_GXMODE == 1) {
        // the tile's pre-activations arrived in registers (coalesced 1-KB loads, two tiles ago): into the lane-linear image,
        // back in the MFMA layout; the registers are reloaded for two tiles ahead
#pragma unroll
        for (int j = 0; j < 4; ++j) gin[(i & 1) * 256 + j * 64 + lane] = __builtin_bit_cast(u32x4, gxr[i & 1][j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = __builtin_bit_cast(f32x4, gin[(i & 1) * 256 + swz[q]]);
        if (i + 2 < NTW) gx_regs(i & 1, i + 2, t); else gx_regs(i & 1, i + 2 - NTW, tn);
      } else {
        asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = (L2S_PROBE & 4) ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f32x4, gin[(i & 1) * 256 + swz[q]]);
        if (i + 2 < NTW) gx_dma(i + 2, t); else gx_dma(i + 2 - NTW, tn);
      }
      __builtin_amdgcn_sched_barrier(0);
      // 2.
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = i * KS + ks, sl = f % R;
        if (L2S_PROBE & 1) {
          acc[ks & 15] += __uint_as_float(wl[sl][0] ^ wh[sl][1]) + (float)bh[ks][0] + (float)bl[ks][1];
        } else {
          acc = MFMA_BF16(as_bf16x8(wl[sl]), bh[ks], acc);
          acc = MFMA_BF16(as_bf16x8(wh[sl]), bl[ks], acc);
          acc = MFMA_BF16(as_bf16x8(wh[sl]), bh[ks], acc);
        }
        const int fn = (f + R) % NF;
        const int o = tb[fn / KS] + (fn % KS) * 2048;
        if (!(L2S_PROBE & 2)) {
          wh[sl] = bload(rw, vw, o);
          wl[sl] = bload(rw, vw, o + 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // 3.
      f32x4 cv, hv;
      const bool uok = tile * 8 + hl * 4 < H;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ig = fast_sigmoid(acc[4 * q + 0] + gq[q][0]), fg = fast_sigmoid(acc[4 * q + 1] + gq[q][1]);
        const float gg = fast_tanh(acc[4 * q + 2] + gq[q][2]), og = fast_sigmoid(acc[4 * q + 3] + gq[q][3]);
        const float cn = fg * c[i][q] + ig * gg;
        c[i][q] = cn;
        cv[q] = cn;
        hv[q] = uok ? og * fast_tanh(cn) : 0.f;      // (units >= H ran on whatever the copy found: h stays 0, finite)
        gout[swz[q]] = __builtin_bit_cast(u32x4, f32x4{ig, fg, gg, og});
      }
      gout[256 + cswz] = __builtin_bit_cast(u32x4, cv);
      gout[320 + cswz] = __builtin_bit_cast(u32x4, hv);
      // h as B-fragment halves: k = unit, kstep tile / 2, fragment lane s + 32 (tile % 2), elements hl * 4 + q
      unsigned h01, l01, h23, l23;
      split2(hv[0], hv[1], h01, l01);
      split2(hv[2], hv[3], h23, l23);
      u32x2* dst = reinterpret_cast<u32x2*>(hnext + (tile >> 1) * 128 + s + 32 * (tile & 1)) + hl;
      dst[0] = u32x2{h01, h23};
      dst[64 * 2] = u32x2{l01, l23};
      // 4. (lane offsets only: see bstore_nt)
      if (!(L2S_PROBE & (4 | 8))) {
        const unsigned so = (unsigned)(t * sg_t + tile * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned v = (tile * 8 + gpiece[j & 1] < H) ? gl[j] + so : VOOR;      // (gl | VOOR stays out of range)
          bstore_nt(__builtin_bit_cast(f32x4, gout[j * 64 + lane]), rg, v, 0);
        }
        const bool cok = tile * 8 + cpiece * 4 < H && !(L2S_PROBE & 32);      // (32: no c / h stores -- the partial sectors)
        bstore_nt(__builtin_bit_cast(f32x4, gout[256 + lane]), rc, cok ? cl + (unsigned)(t * sc_t + tile * 32) : VOOR, 0);
        bstore_nt(__builtin_bit_cast(f32x4, gout[320 + lane]), rh, cok ? hlo + (unsigned)(t * sh_t + tile * 32) : VOOR, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------- forward, eight waves (v3)
// The same recurrence with TWO waves per SIMD (<= 256 registers each), so that a wave stalled behind an HBM-latency
// access in its in-order vector-memory queue leaves the SIMD's matrix pipe to its partner: the pair of waves (w, w + 4)
// owns the NTW tiles of v2's wave w and splits K -- wave w the k-steps [0, KA), wave w + 4 the rest -- with h fragments
// and weight ring for its half only.  Per tile the wave that does NOT finalise hands its 32 x 32 partial sums over through
// LDS (4 KB, lane-linear, then a counter word; the finaliser polls it), the other adds, runs the cell update and the
// tile's io exactly as in v2; the finaliser alternates with the tile parity (its pre-activation slot is refilled for ITS
// next tile, two tiles ahead).  h is single-buffered (two barriers per step) to make room for the hand-over buffers.
template <int KS, int R>
__global__ __launch_bounds__(512, 1) void blstm_l2s8_fwd_kernel(float* __restrict__ gates, float* __restrict__ cell,
                                                                float* __restrict__ hout, int64_t ldo, int64_t dstride,
                                                                const u32x4* __restrict__ wpk, int64_t N, int64_t T, int H) {
  constexpr int RT = 2 * KS, NTW = (RT + 3) / 4, KA = (KS + 1) / 2, KB = KS - KA;
  static_assert(NTW % 2 == 0, "the finaliser alternates with the tile parity");
  static_assert((NTW * KA) % R == 0 && (NTW * KB) % R == 0, "the weight ring must close over one time step in both halves");
  static_assert(NTW * 4 - RT <= 2, "one spare B-fragment slot takes the tiles that do not exist");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  constexpr int HB = (KS + 1) * 128;                               // h: [KS + 1 (spare)][hi, lo][64 lanes], ONE buffer
  constexpr int GI = HB, GO = GI + 8 * 256, PB = GO + 8 * 384, FL = PB + 8 * 256;      // gin[wave][256], gout[wave][384], pbuf[wave][256], flags
  u32x4* hfr = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = wave & 3, kh = wave >> 2;                          // SIMD pair, K half = finaliser parity
  const int s = lane & 31, hl = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int dir = xcd & 1;
  const int64_t grp = (int64_t)slot * 4 + (xcd >> 1);
  if (grp * SEQ >= N) return;
  const int64_t n0 = grp * SEQ;
  const int G4 = H * 4;

  for (int i = tid; i < HB; i += 512) hfr[i] = u32x4{0u, 0u, 0u, 0u};
  volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(smem + FL);      // [wave]: hand-overs published BY that wave
  if (tid < 8) flags[tid] = 0u;

  const srd_t rw = make_srd(wpk + (int64_t)dir * RT * KS * 128);
  const char* gb = reinterpret_cast<const char*>(gates + n0 * T * 2 * G4);
  const srd_t rg = make_srd(gb);
  const srd_t rc = make_srd(cell + n0 * T * 2 * H);
  const srd_t rh = make_srd(hout + n0 * T * ldo);
  const unsigned vw = (unsigned)lane * 16u;
  const int sg_t = 2 * G4 * 4, sc_t = 2 * H * 4, sh_t = (int)ldo * 4;

  unsigned gl[4];
  int gpiece[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int seq = j * 8 + (lane >> 3), piece = ((lane & 7) - (seq >> 1)) & 7;
    gl[j] = (unsigned)((((int64_t)seq * T * 2 + dir) * G4) * 4 + piece * 16) | ((n0 + seq < N) ? 0u : VOOR);
    gpiece[j & 1] = piece;
  }
  const unsigned gsafe = (unsigned)(dir * G4 * 4);
  const int cseq = lane >> 1, cpiece = ((lane & 1) - (cseq >> 3)) & 1;
  const unsigned cmask = (n0 + cseq < N) ? 0u : VOOR;
  const unsigned cl = (unsigned)((((int64_t)cseq * T * 2 + dir) * H) * 4 + cpiece * 16) | cmask;
  const unsigned hlo = (unsigned)(((int64_t)cseq * T * ldo + dir * dstride) * 4 + cpiece * 16) | cmask;
  int swz[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) swz[q] = s * 8 + ((hl * 4 + q + (s >> 1)) & 7);
  const int cswz = s * 2 + ((hl + (s >> 3)) & 1);
  u32x4* gin = smem + GI + wave * 256;
  u32x4* gout = smem + GO + wave * 384;
  u32x4* pmine = smem + PB + wave * 256;                  // what I hand over
  const u32x4* ptheirs = smem + PB + (wave ^ 4) * 256;    // what my partner hands over
  const unsigned gin_lds = lds_addr(gin);

  auto tile_of = [&](int i_) { return pr * NTW + i_; };
  auto tile_off = [&](int i_) { const int t_ = tile_of(i_); return (t_ < RT ? t_ : RT - 1) * KS * 2048; };
  auto gx_dma = [&](int i_, int t_) {       // tile i_ (one I finalise) of time step t_ into my slot
    if (L2S_PROBE & (4 | 16)) return;
    const int tile_ = tile_of(i_);
    const char* base = gb + (int64_t)t_ * sg_t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = !(gl[j] & VOOR) && tile_ * 8 + gpiece[j & 1] < H;
      dma16(base, ok ? gl[j] + (unsigned)(tile_ * 128) : gsafe, gin_lds + (unsigned)(j * 64 * 16));
    }
  };

  // this wave's K half: k-steps [k0, k0 + KH)
  const int k0 = kh ? KA : 0;
  u32x4 wh[R], wl[R];
  float c[NTW / 2][4];
#pragma unroll
  for (int i = 0; i < NTW / 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) c[i][q] = 0.f;
  gx_dma(kh, dir ? (int)T - 1 : 0);           // my first tile: i = kh

  auto run = [&](auto kh_tag) __attribute__((always_inline)) {
    constexpr int KHC = decltype(kh_tag)::value ? KB : KA;     // k-steps of this half
    constexpr int PAR = decltype(kh_tag)::value;               // tiles with i % 2 == PAR are finalised here
    constexpr int NF = NTW * KHC;
#pragma unroll
    for (int f = 0; f < R; ++f) {
      const int o = tile_off(f / KHC) + (k0 + f % KHC) * 2048;
      wh[f] = bload(rw, vw, o);
      wl[f] = bload(rw, vw, o + 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned cnt = 0;                          // hand-overs so far (mine published / theirs consumed: both advance per tile)
#pragma clang loop unroll(disable)
    for (int step = 0; step < (int)T; ++step) {
      const int t = dir ? (int)T - 1 - step : step;
      const int tn = step + 1 < (int)T ? (dir ? t - 1 : t + 1) : t;
      int tb[NTW];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        tb[i] = tile_off(i) + k0 * 2048;
        asm volatile("" : "+s"(tb[i]));
      }
      bf16x8 bh[KHC], bl[KHC];
      {
        const u32x4* hb = hfr + k0 * 128 + lane;
#pragma unroll
        for (int ks = 0; ks < KHC; ++ks) {
          bh[ks] = as_bf16x8(hb[ks * 128]);
          bl[ks] = as_bf16x8(hb[ks * 128 + 64]);
        }
      }
      __syncthreads();                         // every wave holds its fragments of h_{t-1}: h_t may be written
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int tile = tile_of(i);
        const bool fin = (i & 1) == PAR;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KHC; ++ks) {
          const int f = i * KHC + ks, sl = f % R;
          acc = MFMA_BF16(as_bf16x8(wl[sl]), bh[ks], acc);
          acc = MFMA_BF16(as_bf16x8(wh[sl]), bl[ks], acc);
          acc = MFMA_BF16(as_bf16x8(wh[sl]), bh[ks], acc);
          const int fn = (f + R) % NF;
          const int o = tb[fn / KHC] + (fn % KHC) * 2048;
          if (!(L2S_PROBE & 2)) {
            wh[sl] = bload(rw, vw, o);
            wl[sl] = bload(rw, vw, o + 1024);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        ++cnt;
        if (!fin) {
          // hand my partial sums over: four lane-linear 16-byte writes, then the counter (LDS operations of a wave complete in order)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            pmine[q * 64 + lane] = __builtin_bit_cast(u32x4, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) flags[wave] = cnt;
        } else {
          // my pre-activations of this tile (copied in two tiles -- more than 30 vector-memory instructions -- ago; read
          // behind the K loop: sixteen registers the loop does not have), then the slot is refilled for my next tile
          f32x4 gq[4];
          asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) gq[q] = (L2S_PROBE & 4) ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(f32x4, gin[swz[q]]);
          if (i + 2 < NTW) gx_dma(i + 2, t); else gx_dma(i + 2 - NTW, tn);
          while (flags[wave ^ 4] < cnt) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");
          f32x4 cv, hv;
          const bool uok = tile * 8 + hl * 4 < H;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 pq = __builtin_bit_cast(f32x4, ptheirs[q * 64 + lane]);
            const float ig = fast_sigmoid(acc[4 * q + 0] + pq[0] + gq[q][0]), fg = fast_sigmoid(acc[4 * q + 1] + pq[1] + gq[q][1]);
            const float gg = fast_tanh(acc[4 * q + 2] + pq[2] + gq[q][2]), og = fast_sigmoid(acc[4 * q + 3] + pq[3] + gq[q][3]);
            const float cn = fg * c[i / 2][q] + ig * gg;
            c[i / 2][q] = cn;
            cv[q] = cn;
            hv[q] = uok ? og * fast_tanh(cn) : 0.f;
            gout[swz[q]] = __builtin_bit_cast(u32x4, f32x4{ig, fg, gg, og});
          }
          gout[256 + cswz] = __builtin_bit_cast(u32x4, cv);
          gout[320 + cswz] = __builtin_bit_cast(u32x4, hv);
          unsigned h01, l01, h23, l23;
          split2(hv[0], hv[1], h01, l01);
          split2(hv[2], hv[3], h23, l23);
          u32x2* dst = reinterpret_cast<u32x2*>(hfr + (tile >> 1) * 128 + s + 32 * (tile & 1)) + hl;
          dst[0] = u32x2{h01, h23};
          dst[64 * 2] = u32x2{l01, l23};
          if (!(L2S_PROBE & (4 | 8))) {
            const unsigned so = (unsigned)(t * sg_t + tile * 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned v = (tile * 8 + gpiece[j & 1] < H) ? gl[j] + so : VOOR;
              bstore_nt(__builtin_bit_cast(f32x4, gout[j * 64 + lane]), rg, v, 0);
            }
            const bool cok = tile * 8 + cpiece * 4 < H;
            bstore_nt(__builtin_bit_cast(f32x4, gout[256 + lane]), rc, cok ? cl + (unsigned)(t * sc_t + tile * 32) : VOOR, 0);
            bstore_nt(__builtin_bit_cast(f32x4, gout[320 + lane]), rh, cok ? hlo + (unsigned)(t * sh_t + tile * 32) : VOOR, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();                         // h_t complete
    }
  };
  if (kh) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
}

}  // namespace

extern "C" int tssep_lstm_l2s_supported(int H) { return (H > 0 && (H & 3) == 0 && (l2s_ks(H) == 19 || l2s_ks(H) == 20)) ? 1 : 0; }

extern "C" int64_t tssep_lstm_l2s_pack_floats(int H, int which) {
  if (!tssep_lstm_l2s_supported(H) || which < 0 || which > 1) return 0;
  const int64_t KS = l2s_ks(H);
  return (int64_t)2 * (2 * KS) * KS * 2 * 64 * 4;      // u32x4 per lane = 4 floats
}

extern "C" int tssep_lstm_pack_l2s(const float* w_hh_f, const float* w_hh_r, int H, float* wf, void* stream) {
  if (!w_hh_f || !w_hh_r || !wf) return TSSEP_E_NULL;
  if (!tssep_lstm_l2s_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(wf)) return TSSEP_E_ALIGN;
  const int KS = l2s_ks(H);
  hipLaunchKernelGGL(l2s_pack_fwd_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w_hh_f, w_hh_r, H, KS,
                     reinterpret_cast<unsigned short*>(wf));
  return tssep_launch_status();
}

extern "C" int tssep_blstm_l2s_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride, const float* wf,
                                   int64_t N, int64_t T, int H, void* stream) {
  if (!gates || !cell || !hout || !wf) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!tssep_lstm_l2s_supported(H) || (ldo & 3) || (dstride & 3)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(cell) || !aligned16(hout) || !aligned16(wf)) return TSSEP_E_ALIGN;
  const int KS = l2s_ks(H);
  const int64_t groups = (N + SEQ - 1) / SEQ;
  const unsigned grid = (unsigned)(8 * ((groups + 3) / 4));
  const size_t lds = (size_t)(2 * (KS + 1) * 128 + 4 * 2 * 256 + 4 * 384) * 16;
  hipStream_t s = (hipStream_t)stream;
  static bool attr_done[2] = {false, false};
  if (KS == 19) {
    if (!attr_done[0]) { hipFuncSetAttribute(reinterpret_cast<const void*>(&blstm_l2s_fwd_kernel<19, (L2S_RING == 10 ? 10 : 19)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[0] = true; }
    hipLaunchKernelGGL((blstm_l2s_fwd_kernel<19, (L2S_RING == 10 ? 10 : 19)>), dim3(grid), dim3(256), lds, s, gates, cell, hout, ldo, dstride,
                       reinterpret_cast<const u32x4*>(wf), N, T, H);
  } else {
    if (!attr_done[1]) { hipFuncSetAttribute(reinterpret_cast<const void*>(&blstm_l2s_fwd_kernel<20, (L2S_RING == 10 ? 10 : 20)>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done[1] = true; }
    hipLaunchKernelGGL((blstm_l2s_fwd_kernel<20, (L2S_RING == 10 ? 10 : 20)>), dim3(grid), dim3(256), lds, s, gates, cell, hout, ldo, dstride,
                       reinterpret_cast<const u32x4*>(wf), N, T, H);
  }
  return tssep_launch_status();
}

extern "C" int tssep_blstm_l2s8_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride, const float* wf,
                                    int64_t N, int64_t T, int H, void* stream) {
  if (!gates || !cell || !hout || !wf) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (l2s_ks(H) != 19 || (H & 3) || (ldo & 3) || (dstride & 3)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(cell) || !aligned16(hout) || !aligned16(wf)) return TSSEP_E_ALIGN;
  constexpr int KS = 19;
  const int64_t groups = (N + SEQ - 1) / SEQ;
  const unsigned grid = (unsigned)(8 * ((groups + 3) / 4));
  const size_t lds = (size_t)((KS + 1) * 128 + 8 * 256 + 8 * 384 + 8 * 256 + 1) * 16;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&blstm_l2s8_fwd_kernel<19, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  hipLaunchKernelGGL((blstm_l2s8_fwd_kernel<19, 5>), dim3(grid), dim3(512), lds, (hipStream_t)stream, gates, cell, hout, ldo, dstride,
                     reinterpret_cast<const u32x4*>(wf), N, T, H);
  return tssep_launch_status();
}
