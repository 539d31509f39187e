// On-chip-weights BLSTM recurrence on the bf16 matrix cores ("v3"), forward and backward through
// time.  Replaces the T-sequential part of torch.nn.LSTM (tssep/train/rnnp.py:88-95,146-153).
//
// Why a third recurrence: the streaming kernels (lstm.hip) re-read the 1.44 MB W_hh from L2 on every
// step (18-24 us/step whatever the batch); the fp32 cluster kernels (lstm_cluster.hip) keep W_hh in
// registers but need 10 CUs per 8 sequences.  Here a cluster of G = ceil(H/64) <= 5 workgroups
// (512 threads, one per CU) keeps its 64-unit slice of W_hh ON CHIP for the whole launch as split
// bf16 (hi + lo, 152 VGPRs per lane) and evaluates the recurrent product for 32 sequences per step as
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi     on v_mfma_f32_32x32x16_bf16, fp32 accumulation
// (h is split into bf16 hi+lo when it is staged into LDS).  Measured against fp64 on a 253-step
// H=300 LSTM the split adds < 4e-6 to the fp32 rounding noise of h -- far inside the 1e-3 bar.
// Per step: ~1.7 us of MFMA for 32 sequences + one exchange of h between the G workgroups.
//
// Exchange (MI355X_MICROARCH.md "valid forms", recipe R2 -- the data is the flag): values travel in
// 8-byte granules {16-bit tag = epoch | step+1, two values rounded to 24 bits} ("compact granules"
// below); a lane writes its 4 values as ONE 16-byte (write-through) store = 2 granules, and the
// consumer reads 2 granules per 16-byte sc1 load.  Only the 8-byte halves need to be untorn (each
// carries its own tag), which is what aligned dwordx4 accesses give.  One memory hop per step: no drain, no barrier, no flag
// (a flag-based protocol was measured first: 3 hops, 8 us/step, collapsing under load).
// Two slots alternate by step parity (a slot is rewritten only after every workgroup consumed it:
// publishing step t+2 transitively requires everyone to have gathered step t).  All granules are
// zeroed by a memset node before every launch; spins are bounded and raise err[0].
// Cluster membership is taken by arrival ticket (no residency / dispatch-order assumption).
//
// MFMA tile geometry (forward): wave w of workgroup g owns units 64g + 8w .. +7.  A = W rows
// (32 = 8 units x 4 gates), B = h^T columns (32 sequences).  Row r = 4*rg + gate with unit
// u' = 4*(rg&1) + (rg>>1): the D layout (lane = sequence + 32*half, rows (e&3)+8(e>>2)+4*half)
// then gives every lane the 4 gates of the 4 CONSECUTIVE units 4*half .. 4*half+3 of one sequence,
// so the cell update is lane-local and every global / exchange access is a 16-byte vector.
#include <atomic>
#include <type_traits>
#include "common.h"

namespace {

constexpr int KP = 304;              // padded reduction length (19 k-steps of 16)
constexpr int KS = KP / 16;          // 19
constexpr int UPW = 64;              // hidden units per workgroup
constexpr int SEQS = 32;             // sequences per cluster
constexpr int HPITCH = KP * 2 + 16;  // bytes per LDS row of bf16 h (624: 39 slots of 16 B, 39 odd)
constexpr int SPIN_LIMIT = 1 << 21;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// Gate non-linearities on the hardware exp2 / reciprocal units (v_exp_f32, v_rcp_f32): a handful
// of instructions instead of ~25-40 for the OCML expf / tanhf the other kernels use.  4 cells per
// lane and step make the transcendental work visible here (~1.2 us of a 7 us step).  Relative error
// ~2^-22 -- the same class as the split-bf16 product, two orders inside the 1e-3 parity bar.
// (v_exp_f32 / v_rcp_f32 through the builtins: `__frcp_rn` is the correctly rounded reciprocal = the ten-instruction
// IEEE division sequence, and `__expf` adds a range fix-up -- 20 divisions per lane and step in the forward kernel,
// half of the cell update's vector instructions, on the critical path between the MFMAs and the publish.)
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2/(1 + e^{2x}); saturates correctly for large |x| (e^{2x} -> inf or 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}
#define SC1 16
#define SC0 1

// ---- compact exchange granules ---------------------------------------------------------------------
// One naturally aligned 8-byte granule carries TWO values and a 16-bit tag:
//     word0 = tag16 | a24[15:0] << 16,   word1 = a24[23:16] | b24 << 8,
// a24 / b24 = the fp32 value rounded to its upper 24 bits (sign, exponent, 15 mantissa bits: 2^-16
// relative -- what the split-bf16 MFMA operand keeps of it anyway: hi 8 + lo 8 significant bits).
// tag16 = {launch epoch 1..127} << 9 | (step + 1) mod 512.  (Round 4: rounds 1-3 spent 11 bits on the step and refused
// T > 2046.  A poll only has to tell step t from what the SAME slot held before -- step t - 2, or the zeros of the
// reset kernel -- and a slot is rewritten every second step, so the step may wrap; the epoch never is 0, so a tag never
// equals the zeroed buffer's.  The seven epoch bits exclude a stale granule of the previous 126 launches.)
// Against the {32-bit tag, fp32 value}
// granule of round 1 this halves every byte of the exchange: the forward gather moves 41 instead of
// 82 KB per workgroup and step (both sequence halves now fit into ONE round trip of ten 16-byte
// loads), and the backward's exchange working set per XCD drops from 4.9 MB -- more than the 4-MB L2:
// every publish was written back to HBM and every gather missed (PMC round 1: 14 GB written per
// launch for 4.7 GB of d(gates)) -- to 2.5 MB.
__device__ __forceinline__ unsigned tag16_base(const int* err) {
  const unsigned e = (unsigned)__hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(((e % 127u) + 1u) << 9));
}
__device__ __forceinline__ unsigned mk_tag(unsigned tagbase, int64_t step_plus_1) {
  return tagbase | ((unsigned)step_plus_1 & 0x1ffu);
}
__device__ __forceinline__ u32x2 pack_granule(unsigned tag16, float a, float b) {
  const unsigned ua = __float_as_uint(a) + 0x80u, ub = __float_as_uint(b) + 0x80u;   // round to 24 bits
  return u32x2{tag16 | ((ua & 0x00ffff00u) << 8), (ua >> 24) | (ub & 0xffffff00u)};
}
__device__ __forceinline__ float granule_a(unsigned w0, unsigned w1) {
  return __uint_as_float(((w0 >> 8) & 0x00ffff00u) | (w1 << 24));
}
__device__ __forceinline__ float granule_b(unsigned w1) { return __uint_as_float(w1 & 0xffffff00u); }

// ---- cluster membership and work distribution -------------------------------------------------
// header words (zeroed before every launch): [0] arrivals (global ticket in the cross-XCD mode),
// [1] next work item, [8..15] per-XCD tickets, [32..95] per-cluster claim {round+1, item}.
constexpr int HDR_BYTES = 1024;
constexpr int CL_PER_XCD = 8;          // cluster-id stride per XCD (32 CUs / G >= 5 -> at most 6)
struct Membership { int g, cid, ok; };

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCD = false: clusters are G consecutive arrival tickets (workgroups of a cluster sit on different
//   XCDs; the exchange must use agent-scope sc1 accesses that go through to the memory side).
// XCD = true : clusters are formed from workgroups that run on the SAME XCD (HW_REG_XCC_ID), so the
//   exchange is coherent in that XCD's L2: granules are stored with sc0 (written through the CU's
//   L1 into the L2, where they stay) and gathered with sc1 loads (bypass the reader's L1, hit the
//   L2) -- one L2 round trip per step instead of a trip through the fabric: 5.5 vs 6.8 us/step
//   measured.  (sc0 LOADS do not work: outside tgsplit mode they may hit the reader's own L1 and a
//   poll spins on its stale line; `buffer_inv sc0` does not help either -- both measured.)  No dispatch-order assumption: a cluster starts
//   only once its G members have arrived, leftovers exit when every workgroup has taken a ticket,
//   and work items are claimed dynamically, so any set of complete clusters finishes the job.
//   (The verdict must be the same for all members of a cluster: "complete" is monotone and final
//   once every workgroup has arrived.  An early exit on "all items already claimed" is NOT -- a
//   member that left that way stranded its peers waiting for the leader's claim: removed.)
template <bool XCD>
__device__ __forceinline__ Membership join_cluster(unsigned* xhead, int G, int nitems, int* sh) {
  if (threadIdx.x == 0) {
    if (!XCD) {
      const int t = (int)atomicAdd(xhead, 1u);
      sh[0] = t % G; sh[1] = t / G; sh[2] = 1;
    } else {
      const int xcd = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);      // XCC_ID[3:0]
      const int t = (int)atomicAdd(xhead + 8 + xcd, 1u);
      atomicAdd(xhead, 1u);
      const int cl = t / G;
      int ok = 0;
      if (cl < CL_PER_XCD) {
        // bounded (~ seconds): two W-stationary launches running CONCURRENTLY could each hold CUs
        // with half-formed clusters and starve the other's missing members; this library never
        // does that (one compute stream), a caller who does gets ok = 0 / a peer timeout, not a hang
        for (int spins = 0; spins < (SPIN_LIMIT << 1); ++spins) {
          if ((int)ld_agent(xhead + 8 + xcd) >= (cl + 1) * G) { ok = 1; break; }
          if (ld_agent(xhead) >= gridDim.x) { ok = (int)ld_agent(xhead + 8 + xcd) >= (cl + 1) * G; break; }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      sh[0] = t % G; sh[1] = xcd * CL_PER_XCD + cl; sh[2] = ok;
    }
  }
  __syncthreads();
  Membership m;
  m.g = __builtin_amdgcn_readfirstlane(sh[0]);
  m.cid = __builtin_amdgcn_readfirstlane(sh[1]);
  m.ok = __builtin_amdgcn_readfirstlane(sh[2]);
  return m;
}
// work item of this cluster's `round`-th turn (>= nitems: done).  Every workgroup of the cluster
// calls it; contains a barrier.
template <bool XCD>
__device__ __forceinline__ int64_t next_item(unsigned* xhead, const Membership& m, int round,
                                             int nitems, int nclusters, int* sh) {
  if (!XCD) return (int64_t)m.cid + (int64_t)round * nclusters;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* claim = xhead + 32 + m.cid;
    unsigned item;
    if (m.g == 0) {
      item = atomicAdd(xhead + 1, 1u);
      item = item < (unsigned)nitems ? item : 0xffffu;
      __hip_atomic_store(claim, ((unsigned)(round + 1) << 16) | item, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned v;
      int spins = 0;       // the leader claims within microseconds of finishing its previous item
      while (((v = ld_agent(claim)) >> 16) != (unsigned)(round + 1) && ++spins < (SPIN_LIMIT << 3))
        __builtin_amdgcn_s_sleep(2);
      item = (v >> 16) == (unsigned)(round + 1) ? (v & 0xffffu) : 0xfffeu;     // 0xfffe: give up
    }
    sh[3] = item >= 0xfffeu ? nitems + (item == 0xfffeu) : (int)item;
  }
  __syncthreads();
  return (int64_t)__builtin_amdgcn_readfirstlane(sh[3]);
}

// ---- packed weights ------------------------------------------------------------------------
// fwd: wf[dir][g][wave 8][ks 19][hl 2][lane 64] u32x4 : A fragment of k-step ks,
//      lane (i = lane&31, kg = lane>>5): W_hh[gate*H + unit][16 ks + 8 kg + 0..7] as 8 bf16,
//      row i = 4*rg + gate, unit = 64 g + 8 wave + 4*(rg&1) + (rg>>1)
// bwd: wb[dir][g][wave 10][ks 16][hl 2][lane 64] u32x4 : A = W_hh^T tile, rows = OUTPUT units
//      32*wave + i, k = own gate column 16 ks + 8 kg + j with column c = 4*ul + gate of unit 64g + ul
__device__ __forceinline__ float w_at(const float* w, int H, int row, int col) {
  return (row >= 0 && col < H) ? w[(int64_t)row * H + col] : 0.f;
}
__global__ void pack_onchip_kernel(const float* w_hh_f, const float* w_hh_r, int H, int G,
                                   u32x4* wf, u32x4* wb) {
  const int64_t n_f = (int64_t)2 * G * 8 * KS * 2 * 64;
  const int64_t n_b = (int64_t)2 * G * 10 * 16 * 2 * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_f + n_b;
       e += (int64_t)gridDim.x * blockDim.x) {
    float x[8];
    int hl;
    if (e < n_f) {
      int64_t r = e;
      const int lane = (int)(r & 63); r >>= 6;
      hl = (int)(r & 1); r >>= 1;
      const int ks = (int)(r % KS); r /= KS;
      const int wave = (int)(r & 7); r >>= 3;
      const int g = (int)(r % G);
      const int d = (int)(r / G);
      const int i = lane & 31, kg = lane >> 5;
      const int rg = i >> 2, gate = i & 3;
      const int unit = 64 * g + 8 * wave + 4 * (rg & 1) + (rg >> 1);
      const float* w = d ? w_hh_r : w_hh_f;
      for (int j = 0; j < 8; ++j)
        x[j] = unit < H ? w_at(w, H, gate * H + unit, 16 * ks + 8 * kg + j) : 0.f;
    } else {
      int64_t r = e - n_f;
      const int lane = (int)(r & 63); r >>= 6;
      hl = (int)(r & 1); r >>= 1;
      const int ks = (int)(r & 15); r >>= 4;
      const int wave = (int)(r % 10); r /= 10;
      const int g = (int)(r % G);
      const int d = (int)(r / G);
      const int i = lane & 31, kg = lane >> 5;
      const int uo = 32 * wave + i;
      const float* w = d ? w_hh_r : w_hh_f;
      for (int j = 0; j < 8; ++j) {
        const int c = 16 * ks + 8 * kg + j;          // own gate column: unit-local ul, gate
        const int ui = 64 * g + (c >> 2), gate = c & 3;
        x[j] = (ui < H && uo < H) ? w[(int64_t)(gate * H + ui) * H + uo] : 0.f;
      }
    }
    unsigned h[4], l[4];
    for (int j = 0; j < 4; ++j) split2(x[2 * j], x[2 * j + 1], h[j], l[j]);
    const u32x4 v = hl ? u32x4{l[0], l[1], l[2], l[3]} : u32x4{h[0], h[1], h[2], h[3]};
    if (e < n_f) wf[e] = v; else wb[e - n_f] = v;
  }
}

// exchange buffer layout (bytes), per work item (sequence group, direction):
//   header (HDR_BYTES, whole launch): see join_cluster
//   payload: [item][slot 2][G][SEQS][PW] 8-byte granules {tag, value}, zeroed every launch;
//            PW = 64 (forward: own units) or 320 (backward: partial sums for every unit)
struct XBuf {
  unsigned* flags;
  float* payload;
};

// Row of (sequence n, frame t) in the activation tensors.  layout 0: n*T + t (sequence-major, what
// the GEMMs produce today); layout 1: time-major inside groups of 32 sequences,
// (n/32)*T*32 + t*32 + n%32, so that the 32 sequences a cluster touches at one step are contiguous.
#define ROW(n_, t_) (layout ? (((n_) >> 5) * T * 32 + (t_) * 32 + ((n_) & 31)) : ((n_) * T + (t_)))

// ------------------------------------------------------------------------------- forward
constexpr int PUBPITCH = UPW + 4;        // floats per LDS row of [seq][unit] scalars (h, c)

// Wave roles.  All 8 waves hold W and run the MFMAs and the cell update of their unit slice.  Beyond
// that, waves 0-3 ("exchange") do nothing but the inter-workgroup exchange and waves 4-7 ("io") do
// nothing but HBM traffic, because a wave's vector-memory counter retires IN ORDER: with both kinds
// of access in one wave, every gather of the exchange waited for the activation stores and gate
// loads queued in front of it, and HBM time (19 MB per step at 768 sequences, ~4.3 us at the
// achievable 4.5 TB/s) ADDED to the ~5 us exchange chain instead of overlapping it (measured by
// ablation: 2.5 ms per launch with, 1.27 ms without the HBM traffic; re-ordering or delaying the
// accesses inside one wave changed nothing).  With the roles split: 2.25 ms at 768 sequences,
// 1.83 (was 2.07) at 192; the io arm (store acks + load latency, back to back) is what remains
// exposed -- issuing the loads before the stores needs 32 more live registers and spilled.  Gate pre-activations therefore travel
//   HBM -> io registers -> LDS xg[step&1] (two steps ahead) -> cell update (in place: activations)
//   -> io waves -> HBM,
// in a row-contiguous pattern (16 lanes x 16 B per sequence row), and the exchange waves gather
// and publish for all 32 sequences (two halves of 16).
template <bool XCD>
__global__ __launch_bounds__(512, 2) void blstm_onchip_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ cell, float* __restrict__ hout, int64_t ldo,
    int64_t dstride, const u32x4* __restrict__ wf, unsigned* __restrict__ xhead,
    float* __restrict__ xpayload, int* __restrict__ err, int64_t N,
    int64_t T, int H, int G, int nclusters, int layout) {
  const unsigned tagbase = tag16_base(err);
  __shared__ __attribute__((aligned(16))) char hs_hi[SEQS * HPITCH];
  __shared__ __attribute__((aligned(16))) char hs_lo[SEQS * HPITCH];
  __shared__ __attribute__((aligned(16))) float pub[SEQS * PUBPITCH];      // h_t  [seq][unit]
  __shared__ __attribute__((aligned(16))) float cellb[SEQS * PUBPITCH];    // c_t  [seq][unit]
  // gate tiles, three in rotation (step+2 arriving by LDS-DMA, step+1 waiting, step in use /
  // being flushed).  The image is lane-linear, as global_load_lds writes it: f32x4 index
  // ((s/16)*4 + unit/16)*256 + (s%16)*16 + ((unit%16) ^ (s%16)) -- the XOR is applied to the SOURCE
  // address of the DMA so that the cell update's reads (fixed unit, 32 sequences) spread over the banks
  __shared__ f32x4 xg[3][SEQS * UPW];
  __shared__ int s_fail, s_mem[4];
  constexpr int AUXL = SC1, AUXS = XCD ? SC0 : SC1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform: SGPR)
  if (tid == 0) s_fail = 0;
  const bool stream_nt = (layout & 16) != 0;           // non-temporal activation stream (host: N >= 160)
  layout &= 1;
  const int64_t ngroups = (N + SEQS - 1) / SEQS;
  // (membership is wave-uniform by construction; join_cluster returns it through readfirstlane so
  // that the buffer descriptors below are built from SGPRs, no waterfall loops around the loads)
  const Membership mem = join_cluster<XCD>(xhead, G, (int)(2 * ngroups), s_mem);
  if (!mem.ok) return;
  const int g = __builtin_amdgcn_readfirstlane(mem.g);      // scalar: uniform branches, SGPR addressing
  const int j = lane & 31, half = lane >> 5;
  const int ul0 = 8 * wave + 4 * half;                 // this lane's 4 consecutive local units
  const int unit0 = 64 * g + ul0;
  const int foff = j * HPITCH + half * 16;             // B fragment offset (row = sequence j)
  const bool vec_ok = ((H | ldo | dstride) & 3) == 0 && ((((uintptr_t)cell) | ((uintptr_t)hout)) & 15) == 0;
  const bool io_wave = wave >= 4;
  const int s2 = (tid & 255) >> 4, uq = tid & 15;      // exchange / io thread <-> rows s2, s2+16
  // operand-image columns the gather never writes (k >= 64 G) must stay zero
  for (int i = tid; i < SEQS * HPITCH / 4; i += 512) {
    reinterpret_cast<unsigned*>(hs_hi)[i] = 0u;
    reinterpret_cast<unsigned*>(hs_lo)[i] = 0u;
  }

  for (int round = 0;; ++round) {
    const int64_t work = next_item<XCD>(xhead, mem, round, (int)(2 * ngroups), nclusters, s_mem);
    if (work >= 2 * ngroups) {
      if (XCD && work > 2 * ngroups && tid == 0) atomicExch(err, 5);      // leader never claimed (the static map of the cross-XCD mode simply runs past the end)
      break;
    }
    const int dir = (int)(work & 1);
    const int64_t seq0 = (work >> 1) * SEQS;
    // stationary weights -> registers (A fragments, hi and lo)
    u32x4 wh[KS], wl[KS];
    {
      const u32x4* wp = wf + ((((int64_t)(dir * G + g) * 8 + wave) * KS) * 2) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        wh[ks] = wp[(int64_t)(ks * 2 + 0) * 64];
        wl[ks] = wp[(int64_t)(ks * 2 + 1) * 64];
      }
    }
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    float* pl = xpayload + work * 2 * G * SEQS * UPW;          // 8-byte granules of 2 values: 4 B per value
    const __amdgpu_buffer_rsrc_t prs =
        __builtin_amdgcn_make_buffer_rsrc(pl, 0, 2 * G * SEQS * UPW * 4, 0x00020000);

    // ---- io waves: row-contiguous HBM access, lane <-> (row s2 (+16), unit 16 q + (uq ^ s2))
    const int iow = (tid & 255) >> 6;                    // io wave index 0..3 (rows 4 iow .. + 3)
    const int usw = uq ^ s2;                             // this lane's unit within a 16-unit block
    // asynchronous HBM -> LDS copy of the gate tile of step_ (no registers, returns immediately).
    // The streamed activations (this copy, the flush stores below) are non-temporal from 160 sequences
    // up: every access is a full 1-KB run, touched once -- left to the normal policy they evict the
    // exchange granules from the 4-MB L2 (PMC: the publishes were being written back to HBM, 9.3 GB per
    // launch against 7 GB of activations).  Measured effect on the launch: +5 % between two processes on
    // one box, 0-1 % toggled inside one process (layout bit 32) -- inside the pool's run-to-run spread;
    // below 160 sequences (a few clusters, pure latency) the hint costs 10 % and is off.  The backward's
    // accesses are 16-B pieces at a 64-B stride (four instructions per line): there the same hint makes
    // partial-line HBM transactions, 1.2-1.6x slower.
    // Addressing of the io arm: ONE uniform 64-bit base per tensor and step (SGPRs) + a 32-bit per-lane
    // byte offset fixed for the whole work item (+ an immediate for the unit block q).  The first version
    // rebuilt a 64-bit per-lane address for each of the 8 copies and 12 stores of a step; at the kernel's
    // 256-VGPR ceiling the compiler then spilled a few of those values to scratch, and a scratch reload
    // is a VMEM load: its `s_waitcnt vmcnt(0)` also waits -- the counter retires in order -- for the
    // asynchronous copies and stores issued just before it.  The ISA of round 1 had six such reloads
    // inside the io path of every step: six serialised memory round trips per step.
    // ROW(seq0 + s, t) = ROW(seq0, 0) + s * SN + t * ST in both row layouts (seq0 is a multiple of 32).
    const int64_t SN = layout ? 1 : T, ST = layout ? 32 : 1;
    const int64_t row0 = ROW(seq0, 0);
    unsigned goff[2], coff[2], hoff[2];
    bool rok[2], uok[4], uval[4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const unsigned lr = (unsigned)((s2 + 16 * hf) * SN);              // row distance from row0
      rok[hf] = seq0 + s2 + 16 * hf < N;
      goff[hf] = (lr * 2u * (unsigned)H + (unsigned)usw) * 16u;         // gates: 16 B per unit
      coff[hf] = (lr * 2u * (unsigned)H + 4u * (unsigned)uq) * 4u;      // cell
      hoff[hf] = (lr * (unsigned)ldo + 4u * (unsigned)uq) * 4u;         // hout
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uok[q] = 64 * g + 16 * q + usw < H;
      uval[q] = 64 * g + 4 * uq + q < H;
    }
    const bool full4 = vec_ok && 64 * g + 4 * uq + 4 <= H;
    auto gates_base = [&](int64_t t_) {
      return reinterpret_cast<char*>(gates) + (((row0 + t_ * ST) * 2 + dir) * (int64_t)H + 64 * g) * 16;
    };
    // (Hiding the copies from the compiler -- inline asm + one explicit vmcnt(0) at the start of the next io
    // block, so that no LDS read of the step in between waits for them -- measured 1-3 % SLOWER than the
    // builtin in an alternating A/B: kept simple.)
    auto io_dma = [&](int64_t step_, int b_) {
      const int64_t t_ = dir ? T - 1 - step_ : step_;
      const char* gb = gates_base(t_);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (rok[hf] && uok[q]) {
            const auto* src = (const __attribute__((address_space(1))) void*)(gb + goff[hf] + q * 256);
            auto* dst = (__attribute__((address_space(3))) void*)&xg[b_][((hf * 4 + q) * 4 + iow) * 64];
            if (stream_nt) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 2);      // aux 2 = nt
            else __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
          }
        }
      }
    };
    // flush of step_: cell state + h first (their addresses may be reloaded from scratch: harmless while
    // nothing younger is in flight), then -- after the copies of step_ + 2 have been issued -- the gates
    auto io_flush_ch = [&](int64_t step_) {
      const int64_t t_ = dir ? T - 1 - step_ : step_;
      char* cb = reinterpret_cast<char*>(cell) + (((row0 + t_ * ST) * 2 + dir) * (int64_t)H + 64 * g) * 4;
      char* hb = reinterpret_cast<char*>(hout) + ((row0 + t_ * ST) * ldo + dir * dstride + 64 * g) * 4;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (!rok[hf]) continue;
        const int s = s2 + 16 * hf;
        const f32x4 cq = *reinterpret_cast<const f32x4*>(cellb + s * PUBPITCH + 4 * uq);
        const f32x4 hq = *reinterpret_cast<const f32x4*>(pub + s * PUBPITCH + 4 * uq);
        float* cdst = reinterpret_cast<float*>(cb + coff[hf]);
        float* hdst = reinterpret_cast<float*>(hb + hoff[hf]);
        if (full4) {
          if (stream_nt) {
            __builtin_nontemporal_store(cq, reinterpret_cast<f32x4*>(cdst));
            __builtin_nontemporal_store(hq, reinterpret_cast<f32x4*>(hdst));
          } else {
            *reinterpret_cast<f32x4*>(cdst) = cq;
            *reinterpret_cast<f32x4*>(hdst) = hq;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (uval[q]) {
              cdst[q] = cq[q];
              hdst[q] = hq[q];
            }
        }
      }
    };
    auto io_flush_g = [&](int64_t step_, int b_) {
      const int64_t t_ = dir ? T - 1 - step_ : step_;
      char* gb = gates_base(t_);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (!rok[hf]) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (uok[q]) {
            const f32x4 v = xg[b_][((hf * 4 + q) * 4 + iow) * 64 + lane];
            f32x4* dst = reinterpret_cast<f32x4*>(gb + goff[hf] + q * 256);
            if (stream_nt) __builtin_nontemporal_store(v, dst); else *dst = v;
          }
        }
      }
    };
    // cells outside H / N keep a defined (zero) pre-activation: the DMA skips them
    for (int i = tid; i < 3 * SEQS * UPW; i += 512) (&xg[0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (io_wave) {
      io_dma(0, 0);
      if (T > 1) io_dma(1, 1);
    }
    __syncthreads();          // (a barrier drains the copies in flight: vmcnt(0))
    int buf = 0;              // step % 3

    for (int64_t step = 0; step < T; ++step) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (step > 0) {
        // ---- exchange waves: gather h_{t-1} of every source workgroup for BOTH sequence halves in one
        // round trip: ten 16-byte loads in flight, each two granules = the 4 units 4 uq .. + 3 of one
        // (source workgroup, sequence); then into the bf16 hi+lo operand image
        if (!io_wave) {
          const int slot = (int)((step - 1) & 1);
          const unsigned want = mk_tag(tagbase, step);
          u32x4 v[10];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int gs = 0; gs < 5; ++gs)
              v[5 * hf + gs] = gs < G ? __builtin_amdgcn_raw_buffer_load_b128(
                                            prs, (((slot * G + gs) * SEQS + s2 + 16 * hf) * UPW + 4 * uq) * 4, 0, AUXL)
                                      : u32x4{want, 0u, want, 0u};
          int spins = 0;
          bool fail = false;
          for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 10; ++i) ok = ok && (v[i][0] & 0xffffu) == want && (v[i][2] & 0xffffu) == want;
            if (ok) break;
            if (++spins > SPIN_LIMIT) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int gs = 0; gs < 5; ++gs)
                if (gs < G && !((v[5 * hf + gs][0] & 0xffffu) == want && (v[5 * hf + gs][2] & 0xffffu) == want))
                  v[5 * hf + gs] = __builtin_amdgcn_raw_buffer_load_b128(
                      prs, (((slot * G + gs) * SEQS + s2 + 16 * hf) * UPW + 4 * uq) * 4, 0, AUXL);
          }
          if (fail) s_fail = 1;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int s = s2 + 16 * hf;
#pragma unroll
            for (int gs = 0; gs < 5; ++gs) {
              const int k = 64 * gs + 4 * uq;                      // column of h = unit index
              if (gs < G && k < KP) {
                const u32x4 w = v[5 * hf + gs];
                unsigned h0, l0, h1, l1;
                split2(granule_a(w[0], w[1]), granule_b(w[1]), h0, l0);
                split2(granule_a(w[2], w[3]), granule_b(w[3]), h1, l1);
                *reinterpret_cast<u32x2*>(hs_hi + s * HPITCH + 2 * k) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(hs_lo + s * HPITCH + 2 * k) = u32x2{l0, l1};
              }
            }
          }
        }
        __syncthreads();
        if (s_fail) {
          if (tid == 0) atomicExch(err, 3);
          return;
        }
        // ---- recurrent product for 32 gate rows x 32 sequences (h_{-1} = 0: skipped at step 0)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hs_hi + foff + ks * 32);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hs_lo + foff + ks * 32);
          acc = MFMA_BF16(as_bf16x8(wl[ks]), bh, acc);
          acc = MFMA_BF16(as_bf16x8(wh[ks]), bl, acc);
          acc = MFMA_BF16(as_bf16x8(wh[ks]), bh, acc);
        }
      }
      // ---- lane-local cell update: acc[4q .. 4q+3] = gates (i,f,g,o) of unit unit0 + q; the
      // pre-activations come from xg[buf] and the activations replace them in place
      {
        f32x4 hv, cv;
        f32x4* xrow = &xg[buf][((j >> 4) * 4 + (ul0 >> 4)) * 256 + (j & 15) * 16];
        const int uq0 = ul0 & 15, sx = j & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gx = xrow[(uq0 + q) ^ sx];
          const float a0 = acc[4 * q + 0] + gx[0], a1 = acc[4 * q + 1] + gx[1];
          const float a2 = acc[4 * q + 2] + gx[2], a3 = acc[4 * q + 3] + gx[3];
          const float ig = fast_sigmoid(a0), fg = fast_sigmoid(a1);
          const float gg = fast_tanh(a2), og = fast_sigmoid(a3);
          const float cn = fg * c[q] + ig * gg;
          c[q] = cn;
          cv[q] = cn;
          xrow[(uq0 + q) ^ sx] = f32x4{ig, fg, gg, og};
          hv[q] = (unit0 + q < H) ? og * fast_tanh(cn) : 0.f;
        }
        *reinterpret_cast<f32x4*>(pub + j * PUBPITCH + ul0) = hv;
        *reinterpret_cast<f32x4*>(cellb + j * PUBPITCH + ul0) = cv;
      }
      __syncthreads();
      if (!io_wave) {
        // ---- publish h_t: 2 granules = 4 units per 16-byte store, 256-byte runs per sequence row
        // (write-through; no drain, no flag -- the tag is the flag)
        const unsigned tag = mk_tag(tagbase, step + 1);
        const int slot = (int)(step & 1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int s = s2 + 16 * hf;
          const f32x4 pv = *reinterpret_cast<const f32x4*>(pub + s * PUBPITCH + 4 * uq);
          const int go = (((slot * G + g) * SEQS + s) * UPW + 4 * uq) * 4;
          const u32x2 ga = pack_granule(tag, pv[0], pv[1]), gb = pack_granule(tag, pv[2], pv[3]);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{ga[0], ga[1], gb[0], gb[1]}, prs, go, 0, AUXS);
        }
      } else {
        // ---- io waves: start the copy of step+2's pre-activations (into the tile flushed a step
        // ago) BEFORE the stores of this step's activations / cell / h: the memory counter retires
        // in order, so loads issued after the stores would also wait for their acknowledgements
        int b2 = buf + 2;
        if (b2 >= 3) b2 -= 3;
        io_flush_ch(step);
        if (step + 2 < T) io_dma(step + 2, b2);
        io_flush_g(step, buf);
      }
      buf = buf == 2 ? 0 : buf + 1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ backward
// dh_{t-1} = dgates_t x W_hh as a reduce-scatter: workgroup g owns the 256 gate columns of its 64
// units (the K slice: d(gate) values are produced locally by the lane-local cell backward) and
// multiplies them into partial sums for ALL 320 output units; the owner of a unit adds the G
// partials in a fixed order.  Output tiles (32 units) 0..7 belong to waves 0..7 (16 k-steps
// each); tiles 8 and 9 are split over the 8 waves by k (4 k-steps of one of them each) and reduced
// through LDS, so every wave issues 60 MFMAs per step and holds 160 stationary registers.  The
// partials are transposed through LDS so that each publish instruction writes 1 KB contiguous
// (scattered 32-byte pieces cost 3.7 us/step more, measured).
constexpr int DPITCH = 256 * 2 + 16;     // bytes per LDS row of bf16 d(gates): 33 slots of 16 B
constexpr int PPITCH = 5 * UPW + 4;      // floats per LDS row of partial dh (bank-skewed)

template <bool XCD>
__global__ __launch_bounds__(512, 2) void blstm_onchip_bwd_kernel(
    float* __restrict__ gates, const float* __restrict__ cell, const float* __restrict__ dhout,
    int64_t ldo, int64_t dstride, const u32x4* __restrict__ wb, unsigned* __restrict__ xhead,
    float* __restrict__ xpayload, int* __restrict__ err, int64_t N, int64_t T, int H, int G,
    int nclusters, int layout) {
  const unsigned tagbase = tag16_base(err);
  __shared__ __attribute__((aligned(16))) char dg_hi[SEQS * DPITCH];
  __shared__ __attribute__((aligned(16))) char dg_lo[SEQS * DPITCH];
  __shared__ __attribute__((aligned(16))) float red[8 * 64 * 16];          // tiles 8/9 partials
  __shared__ __attribute__((aligned(16))) float psum[SEQS * PPITCH];       // [seq][unit] partial dh
  __shared__ u32x4 wl_sh[4 * 512];        // lo words of the shared-tile fragments (register relief)
  // the per-thread carries of the cell backward (dc chain, c_{t-1} prefetched a step ahead) rest in LDS
  // across the MFMA phase -- the high-pressure zone, where the compiler otherwise spilled a W fragment to
  // scratch and reloaded it right behind the d(gates) stores: `s_waitcnt vmcnt(0)` for the reload then
  // waited for the store acknowledgements of every step
  __shared__ f32x4 carry[2 * 512];
  __shared__ int s_fail, s_mem[4];
  constexpr int AUXL = SC1, AUXS = XCD ? SC0 : SC1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform: SGPR)
  if (tid == 0) s_fail = 0;
  const int64_t ngroups = (N + SEQS - 1) / SEQS;
  const Membership mem = join_cluster<XCD>(xhead, G, (int)(2 * ngroups), s_mem);
  if (!mem.ok) return;
  const int g = __builtin_amdgcn_readfirstlane(mem.g);      // scalar: uniform branches, SGPR addressing
  const int j = lane & 31, half = lane >> 5;
  const int s = tid >> 4, uq = tid & 15;                   // cell backward: sequence, unit quad
  const int unit0 = 64 * g + 4 * uq;
  const int Hp = G * UPW;
  const int foff = j * DPITCH + half * 16;
  const bool vec_ok = ((H | ldo | dstride) & 3) == 0 &&
                      ((((uintptr_t)cell) | ((uintptr_t)dhout)) & 15) == 0;

  for (int round = 0;; ++round) {
    const int64_t work = next_item<XCD>(xhead, mem, round, (int)(2 * ngroups), nclusters, s_mem);
    if (work >= 2 * ngroups) {
      if (XCD && work > 2 * ngroups && tid == 0) atomicExch(err, 5);      // leader never claimed (the static map of the cross-XCD mode simply runs past the end)
      break;
    }
    const int dir = (int)(work & 1);
    const int64_t seq0 = (work >> 1) * SEQS;
    const int64_t n = seq0 + s;
    const bool nvalid = n < N;
    // addressing as in the forward: one uniform base per tensor and step + 32-bit per-lane byte offsets
    // (ROW(seq0 + s, t) = ROW(seq0, 0) + s * SN + t * ST in both row layouts), instead of 64-bit per-lane
    // addresses rebuilt for every access: fewer live registers, fewer scratch reloads -- each of which
    // drains the in-order memory counter, i.e. waits for the d(gates) stores issued just before it
    const int64_t SN = layout ? 1 : T, ST = layout ? 32 : 1;
    const int64_t row0 = ROW(seq0, 0);
    const unsigned lr = (unsigned)(s * SN);
    const unsigned goffb = (lr * 2u * (unsigned)H + 4u * (unsigned)uq) * 16u;       // gates (16 B per unit)
    const unsigned coffb = (lr * 2u * (unsigned)H + 4u * (unsigned)uq) * 4u;        // cell
    const unsigned hoffb = (lr * (unsigned)ldo + 4u * (unsigned)uq) * 4u;           // dhout
    const bool full = nvalid && unit0 + 4 <= H && vec_ok;
    // stationary W_hh^T fragments: own tile (16 k-steps) + k-steps {2w, 2w+1} of tiles 8 and 9
    u32x4 wh[20], wl[16];
    {
      const u32x4* wbase = wb + ((int64_t)(dir * G + g) * 10) * 16 * 2 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        wh[ks] = wbase[(((int64_t)wave * 16 + ks) * 2 + 0) * 64];
        wl[ks] = wbase[(((int64_t)wave * 16 + ks) * 2 + 1) * 64];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int tile = 8 + (wave & 1), ks = 4 * (wave >> 1) + x;
        wh[16 + x] = wbase[(((int64_t)tile * 16 + ks) * 2 + 0) * 64];
        wl_sh[x * 512 + tid] = wbase[(((int64_t)tile * 16 + ks) * 2 + 1) * 64];
      }
    }
    carry[tid] = f32x4{0.f, 0.f, 0.f, 0.f};            // dcc
    carry[512 + tid] = f32x4{0.f, 0.f, 0.f, 0.f};      // c_{t-1} loaded by the previous step
    float* pl = xpayload + work * 2 * G * SEQS * Hp;            // 8-byte granules of 2 values: 4 B per value
    const __amdgpu_buffer_rsrc_t prs =
        __builtin_amdgcn_make_buffer_rsrc(pl, 0, 2 * G * SEQS * Hp * 4, 0x00020000);
    __syncthreads();

    // ---- (1) this thread's saved activations, software-pipelined: the loads of step t + 1 are issued at
    // the END of step t (behind its publish), not at the start of step t + 1 in front of the gather -- a
    // wave's memory counter retires in order, so a gather issued behind HBM loads returns no earlier than
    // they do, even when its granules have long been sitting in the L2.  The values are not live across
    // the MFMA phase (same register pressure as before).
    f32x4 g4[4], ct0 = {0.f, 0.f, 0.f, 0.f}, cp = ct0, dh = ct0;
    char* gb = nullptr;
    auto load_act = [&](int64_t step_) {
      const int64_t t_ = dir ? step_ : T - 1 - step_;
      const int64_t tp_ = dir ? t_ + 1 : t_ - 1;
      const bool prev_ = step_ + 1 < T;
      cp = f32x4{0.f, 0.f, 0.f, 0.f};
      dh = cp;
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      gb = reinterpret_cast<char*>(gates) + (((row0 + t_ * ST) * 2 + dir) * (int64_t)H + 64 * g) * 16 + goffb;
      const char* cb = reinterpret_cast<const char*>(cell) + (((row0 + t_ * ST) * 2 + dir) * (int64_t)H + 64 * g) * 4 + coffb;
      const char* cpb = reinterpret_cast<const char*>(cell) + (((row0 + tp_ * ST) * 2 + dir) * (int64_t)H + 64 * g) * 4 + coffb;
      const char* hb = reinterpret_cast<const char*>(dhout) + ((row0 + t_ * ST) * ldo + dir * dstride + 64 * g) * 4 + hoffb;
      if (full) {
#pragma unroll
        for (int q = 0; q < 4; ++q) g4[q] = *reinterpret_cast<const f32x4*>(gb + 16 * q);
        if (step_ == 0) ct0 = *reinterpret_cast<const f32x4*>(cb);
        if (prev_) cp = *reinterpret_cast<const f32x4*>(cpb);
        dh = *reinterpret_cast<const f32x4*>(hb);
      } else if (nvalid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (unit0 + q < H) {
            g4[q] = *reinterpret_cast<const f32x4*>(gb + 16 * q);
            if (step_ == 0) ct0[q] = reinterpret_cast<const float*>(cb)[q];
            if (prev_) cp[q] = reinterpret_cast<const float*>(cpb)[q];
            dh[q] = reinterpret_cast<const float*>(hb)[q];
          }
      }
    };
    load_act(0);
    carry[512 + tid] = ct0;                 // the cell state of step 0 (later steps: what the previous one loaded)

    for (int64_t step = 0; step < T; ++step) {
      const bool has_prev = step + 1 < T;
      const f32x4 ct = carry[512 + tid];
      f32x4 dcc = carry[tid];
      // ---- (2) reduce-scatter: add the G partial dh published with tag = step (fixed order)
      // (issuing the gather BEFORE the activation loads -- so that it does not retire behind their HBM
      // latency -- was tried: the compiler then waits for everything at the first tag check anyway and
      // spills 16 registers)
      if (step > 0) {
        const int slot = (int)((step - 1) & 1);
        const unsigned want = mk_tag(tagbase, step);
        u32x4 v[5];
        // this workgroup's own partial never leaves the CU: it is still in psum (rewritten only
        // after this step's barrier) -- 1/G less exchange traffic, same summation order.  One 16-byte
        // load = 2 granules = this thread's 4 units from one source workgroup.
        const f32x4 own = *reinterpret_cast<const f32x4*>(psum + s * PPITCH + unit0);
#pragma unroll
        for (int gs = 0; gs < 5; ++gs)
          v[gs] = (gs < G && gs != g) ? __builtin_amdgcn_raw_buffer_load_b128(
                                            prs, (((slot * G + gs) * SEQS + s) * Hp + unit0) * 4, 0, AUXL)
                                      : u32x4{want, 0u, want, 0u};
        int spins = 0;
        bool fail = false;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 5; ++i) ok = ok && (v[i][0] & 0xffffu) == want && (v[i][2] & 0xffffu) == want;
          if (ok) break;
          if (++spins > SPIN_LIMIT) { fail = true; break; }
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int gs = 0; gs < 5; ++gs)
            if (gs < G && gs != g && !((v[gs][0] & 0xffffu) == want && (v[gs][2] & 0xffffu) == want))
              v[gs] = __builtin_amdgcn_raw_buffer_load_b128(
                  prs, (((slot * G + gs) * SEQS + s) * Hp + unit0) * 4, 0, AUXL);
        }
        if (fail) s_fail = 1;
#pragma unroll
        for (int gs = 0; gs < 5; ++gs) {             // fixed order gs = 0..G-1 (own partial in its place)
          if (gs == g) {
            dh[0] += own[0]; dh[1] += own[1]; dh[2] += own[2]; dh[3] += own[3];
          } else if (gs < G) {
            dh[0] += granule_a(v[gs][0], v[gs][1]);
            dh[1] += granule_b(v[gs][1]);
            dh[2] += granule_a(v[gs][2], v[gs][3]);
            dh[3] += granule_b(v[gs][3]);
          }
        }
      }
      // ---- (3) cell backward; d(gates) -> global (in place) and LDS (bf16 hi+lo, MFMA B operand)
      unsigned hi[8], lo[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (nvalid && unit0 + q < H) {
          const float tc = fast_tanh(ct[q]);
          const float d_o = dh[q] * tc;
          const float dc = dh[q] * g4[q][3] * (1.f - tc * tc) + dcc[q];
          dcc[q] = dc * g4[q][1];
          d[0] = dc * g4[q][2] * g4[q][0] * (1.f - g4[q][0]);
          d[1] = dc * cp[q] * g4[q][1] * (1.f - g4[q][1]);
          d[2] = dc * g4[q][0] * (1.f - g4[q][2] * g4[q][2]);
          d[3] = d_o * g4[q][3] * (1.f - g4[q][3]);
          *reinterpret_cast<f32x4*>(gb + 16 * q) = d;
        }
        split2(d[0], d[1], hi[2 * q], lo[2 * q]);
        split2(d[2], d[3], hi[2 * q + 1], lo[2 * q + 1]);
      }
      {
        const int o = s * DPITCH + 32 * uq;                 // 16 gate columns = 32 bytes of bf16
        *reinterpret_cast<u32x4*>(dg_hi + o) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4*>(dg_hi + o + 16) = u32x4{hi[4], hi[5], hi[6], hi[7]};
        *reinterpret_cast<u32x4*>(dg_lo + o) = u32x4{lo[0], lo[1], lo[2], lo[3]};
        *reinterpret_cast<u32x4*>(dg_lo + o + 16) = u32x4{lo[4], lo[5], lo[6], lo[7]};
      }
      carry[tid] = dcc;                 // same thread reads them back at the next step
      carry[512 + tid] = cp;
      __syncthreads();
      if (s_fail) {
        if (tid == 0) atomicExch(err, 4);
        return;
      }
      if (has_prev) {
        // ---- (4) partial dh_prev: own output tile over all 16 k-steps; wave w also covers k-steps
        //      4 (w>>1) .. +3 of shared tile 8 + (w&1)
        // three accumulator chains round-robin (own tile even / odd k-steps, shared tile): MFMAs
        // into one accumulator back to back wait for each other (the loop measured 2x issue-bound)
        f32x16 acc, acc1, accs;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; accs[e] = 0.f; }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int kx = 4 * (wave >> 1) + x;
          const bf16x8 sh = *reinterpret_cast<const bf16x8*>(dg_hi + foff + kx * 32);
          const bf16x8 sl = *reinterpret_cast<const bf16x8*>(dg_lo + foff + kx * 32);
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            const int ks = 4 * x + 2 * y;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(dg_hi + foff + ks * 32);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(dg_lo + foff + ks * 32);
            const bf16x8 ch = *reinterpret_cast<const bf16x8*>(dg_hi + foff + ks * 32 + 32);
            const bf16x8 cl = *reinterpret_cast<const bf16x8*>(dg_lo + foff + ks * 32 + 32);
            acc = MFMA_BF16(as_bf16x8(wl[ks]), bh, acc);
            acc1 = MFMA_BF16(as_bf16x8(wl[ks + 1]), ch, acc1);
            if (y == 0) accs = MFMA_BF16(as_bf16x8(wl_sh[x * 512 + tid]), sh, accs);
            acc = MFMA_BF16(as_bf16x8(wh[ks]), bl, acc);
            acc1 = MFMA_BF16(as_bf16x8(wh[ks + 1]), cl, acc1);
            if (y == 0) accs = MFMA_BF16(as_bf16x8(wh[16 + x]), sl, accs);
            acc = MFMA_BF16(as_bf16x8(wh[ks]), bh, acc);
            acc1 = MFMA_BF16(as_bf16x8(wh[ks + 1]), ch, acc1);
            if (y == 1) accs = MFMA_BF16(as_bf16x8(wh[16 + x]), sh, accs);
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc1[e];
        // own tile -> psum[seq][unit] (D rows (e&3) + 8 (e>>2) + 4 half); shared-tile partials -> red
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f32x4*>(psum + j * PPITCH + 32 * wave + 8 * q + 4 * half) =
              f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          *reinterpret_cast<f32x4*>(red + ((wave * 4 + q) * 64 + lane) * 4) =
              f32x4{accs[4 * q], accs[4 * q + 1], accs[4 * q + 2], accs[4 * q + 3]};
        }
        __syncthreads();
        {   // wave w sums slice e4 = w & 3 of tile 8 + (w >> 2) over the 4 waves that hold it
          const int tt = wave >> 2, e4 = wave & 3;
          f32x4 sum = *reinterpret_cast<const f32x4*>(red + ((tt * 4 + e4) * 64 + lane) * 4);
#pragma unroll
          for (int ww = 1; ww < 4; ++ww)
            sum += *reinterpret_cast<const f32x4*>(red + (((tt + 2 * ww) * 4 + e4) * 64 + lane) * 4);
          *reinterpret_cast<f32x4*>(psum + j * PPITCH + 32 * (8 + tt) + 8 * e4 + 4 * half) = sum;
        }
        __syncthreads();
        // ---- (5) publish: 2 granules per lane and store, 1 KB contiguous per wave instruction
        const unsigned tag = mk_tag(tagbase, step + 1);
        const int slot = (int)(step & 1);
        // quad index qd = tid + 512 i walks the [seq][Hp/4] array linearly: a wave instruction writes
        // 1 KB contiguous (4 x 256-byte runs per instruction measured 3x slower under load); one
        // 16-byte store = 2 granules = 4 units; (seq, quad) advance incrementally -- no divisions
        {
          const int hp4 = Hp >> 2;
          int sq = tid / hp4, uq4 = tid - sq * hp4;
          const int dsq = 512 / hp4, duq = 512 - dsq * hp4;
          for (int qd = tid; qd < SEQS * hp4; qd += 512) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(psum + sq * PPITCH + 4 * uq4);
            if ((uq4 >> 4) != g) {                           // (own 64 units stay in psum)
              const u32x2 ga = pack_granule(tag, v[0], v[1]), gb = pack_granule(tag, v[2], v[3]);
              __builtin_amdgcn_raw_buffer_store_b128(u32x4{ga[0], ga[1], gb[0], gb[1]}, prs,
                                                     (((slot * G + g) * SEQS + sq) * Hp + 4 * uq4) * 4, 0, AUXS);
            }
            sq += dsq;
            uq4 += duq;
            if (uq4 >= hp4) { uq4 -= hp4; ++sq; }
          }
        }
      }
      if (has_prev) load_act(step + 1);     // prefetch: in flight while the peers finish their step
      // no barrier here: dg_* is rewritten after the next step's gather, psum/red after its barrier
    }
  }
}


// =====================================================================================================
// Forward recurrence, interleaved variant (round 3): NGA groups of 16 sequences per cluster instead of one
// group of 32.  A step of the kernel above is a serial chain -- gather h (two L2 round trips, 3-4 us under
// load) -> MFMAs (1.7 us) -> cell update (0.9 us) -> publish -- in which the matrix cores and the vector
// ALUs idle through the exchange and the exchange idles through the arithmetic.  Here the SAME stationary
// W serves NGA independent sequence groups in rotation: while group p's h travels, the other groups compute,
// so a group's gather is issued a phase ahead of its use and has NGA - 1 phases to arrive.
//  * group = 16 sequences: `v_mfma_f32_16x16x32_bf16` (A = 16 gate rows x 32 k, B = 32 k x 16 sequences;
//    a wave owns 2 row blocks = 8 units x 4 gates, 2 x 10 x 3 = 60 MFMAs per phase), K padded to 320;
//    row r of a block = 4 u' + gate, so lane (u' = lane / 16, sequence = lane % 16) receives the four gates
//    of unit 8 wave + 4 rb + u' in its four accumulator registers: the cell update stays lane-local (two
//    cells per lane and phase);
//  * LDS: the h operand image of every group (NGA x 21 KB), ONE h / c staging tile, and a ring of four
//    16-KB gate tiles in SEPARATE arrays indexed by the phase (static: the compiler's alias analysis then
//    knows which LDS-DMA a read can depend on): the tile of phase n + 2 is requested after the flush of
//    phase n -- two phases of latency budget for the asynchronous copy, four tiles live;
//  * barriers are `s_waitcnt lgkmcnt(0); s_barrier` in inline assembly: `__syncthreads()` waits for every
//    outstanding LDS-DMA (vmcnt(0)), i.e. it would put an HBM round trip into every phase; the io waves
//    wait for the copy they need with an explicit `vmcnt(10)` (the six stores and four copies queued behind it
//    may still be in flight: the counter retires in order, and all vector-memory instructions of the io arm
//    are unconditional -- lanes outside N / H store to out-of-range buffer offsets / copy a dummy address);
//  * work: bundles of NGA consecutive sequence groups of ONE direction (W is per direction); the launcher picks
//    NGA so that it divides the number of groups (every bundle is full).
// NGA = 4: four phases per step; NGA = 2: the four phases are two steps; NGA = 1: four steps (T-tail masked).
#ifndef ONCHIP16_ABL
#define ONCHIP16_ABL 0   // experiment builds (results wrong): 1 no MFMAs, 2 no cell arithmetic, 4 no io arm, 8 no exchange, 16 no tag polling,
                         // 32 half of the forward's MFMAs (one of a wave's two row blocks: what a cluster of ten workgroups would leave)
#endif
constexpr int SQ = 16;
constexpr int KP2 = 320, KS2 = 10;
constexpr int HP2 = KP2 * 2 + 16;            // 656 B: 16 rows x 16 B conflict-free (164 dwords = 36 mod 64)
constexpr int TILE4 = SQ * UPW;              // f32x4 per gate tile (16 KB)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void pack_onchip16_kernel(const float* w_hh_f, const float* w_hh_r, int H, int G, int NW, u32x4* wf) {
  // wf[dir][g][wave NW][rb 2][ks 10][hl 2][lane 64]: A fragment (16 rows x 32 k) of row block rb, k-step ks:
  // lane (i = lane % 16, kg = lane / 16) holds W_hh[gate * H + unit][32 ks + 8 kg + 0..7], i = 4 u' + gate,
  // unit = 8 NW g + 8 wave + 4 rb + u'   (NW = 8: five workgroups of 64 units; NW = 4: ten of 32)
  const int UW = 8 * NW;
  const int64_t n = (int64_t)2 * G * NW * 2 * KS2 * 2 * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int lane = (int)(r & 63); r >>= 6;
    const int hl = (int)(r & 1); r >>= 1;
    const int ks = (int)(r % KS2); r /= KS2;
    const int rb = (int)(r & 1); r >>= 1;
    const int wave = (int)(r % NW); r /= NW;
    const int g = (int)(r % G);
    const int d = (int)(r / G);
    const int i = lane & 15, kg = lane >> 4;
    const int unit = UW * g + 8 * wave + 4 * rb + (i >> 2), gate = i & 3;
    const float* w = d ? w_hh_r : w_hh_f;
    // the k axis of workgroup g starts at its OWN slice: local k <-> hidden unit (k + UW g) mod UW G (the operand image in
    // LDS is rotated the same way) -- the k-steps of the own slice are the first ones for every workgroup, a compile-time index
    float x[8];
    for (int j = 0; j < 8; ++j) {
      const int kl = 32 * ks + 8 * kg + j;
      const int kc = kl < UW * G ? (kl + UW * g) % (UW * G) : H;
      x[j] = unit < H ? w_at(w, H, gate * H + unit, kc) : 0.f;
    }
    unsigned h[4], l[4];
    for (int j = 0; j < 4; ++j) split2(x[2 * j], x[2 * j + 1], h[j], l[j]);
    wf[e] = hl ? u32x4{l[0], l[1], l[2], l[3]} : u32x4{h[0], h[1], h[2], h[3]};
  }
}

#define MFMA16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// Where the exchange waves request the next phase's h (two groups per cluster only; see ONCHIP16_BWD_GATHER below):
// 0 = behind this phase's publish (the loads' round trip through L2 sits in front of the next phase's barrier),
// 1 = behind the first barrier, k >= 2 = after k - 1 MFMA k-steps.  768 sequences 1.45 -> 1.36 ms, 1 536: 2.70 -> 2.50,
// 3 072: 5.73 -> 5.43 (positions 1, 4, 7, 11 within noise of each other; profiles/r3_onchip16_fwd_gather.jsonl) --
// two groups now beat four (2.63 / 5.65 ms, whose requests were two phases ahead already) at every size.
#ifndef ONCHIP16_FWD_EARLY_DMA
#define ONCHIP16_FWD_EARLY_DMA 1
#endif
// one group per cluster: s_sleep units (64 cycles) between the publish and the first request of the peers' values
#ifndef ONCHIP16_FWD_G1_DELAY
#define ONCHIP16_FWD_G1_DELAY 14
#endif
#ifndef ONCHIP16_FWD_GATHER
#define ONCHIP16_FWD_GATHER 4
#endif
// one group per cluster: the own slice's products of the NEXT step behind the publish (see `accn` in the kernel), and the
// pause that is left between them and the first request of the peers' values.  The twelve MFMAs are the pause: 0.692 ->
// 0.620 ms at 8 sequences, 0.682 -> 0.617 at 32 with none (3: 0.638 / 0.632, 6: 0.657 / 0.654, 10: 0.686 / 0.682;
// profiles/r4_onchip16_fwd_own_early.jsonl)
#ifndef ONCHIP16_FWD_OWN_EARLY
#define ONCHIP16_FWD_OWN_EARLY 1
#endif
#ifndef ONCHIP16_FWD_G1_DELAY_OWN
#define ONCHIP16_FWD_G1_DELAY_OWN 0
#endif
// Two groups and more: the gather loads of the NEXT phase are requested beside this phase's MFMAs, and the exchange wave
// issues its publish store(s) AFTER them.  The vector-memory counter retires in order, so at the next phase's start the
// loads are long complete and only the publish stores are outstanding -- `vmcnt(<stores issued since>)` is the correct
// wait.  The compiler cannot know that across the loop (its scoreboard restarts at the loop header with the loads as the
// NEWEST operations) and emits vmcnt(3) .. vmcnt(0): the exchange waves wait for the ACKNOWLEDGEMENT of the publish stores
// in front of every phase (trace, round 5: 1 100 cycles in the forward, 2 500 in the backward at 3 072 sequences).
// 1 = the loads are issued through inline assembly (invisible to the compiler's wait insertion; their registers are
// zero-filled first, so a copy the compiler might make before the data lands carries tag 0 and falls to the re-poll
// path) and waited for with an explicit count.  Measured (round 5, alternating builds on one box, 3 072 sequences): the
// exchange waves then reach the first barrier 1 100-1 200 cycles earlier -- and wait there for the io waves, whose tail
// (six LDS copies, ~290 cycles each to issue under load) is as long: backward 6.19 -> 6.23 ms, forward 5.37 -> 5.55 (the
// zero fills and wait states inside the MFMA section).  Kept as an option, default 0
// (profiles/r5_onchip16_tails.jsonl).
#ifndef ONCHIP16_ASM_GATHER
#define ONCHIP16_ASM_GATHER 0
#endif
__device__ __forceinline__ void asm_gather_load(u32x4& v, const void* base, unsigned byte_off) {
  v = u32x4{0u, 0u, 0u, 0u};
  // (s_nop 4: the base may have been written by a VALU instruction -- v_readlane of a spilled SGPR -- immediately before;
  // a vector-memory instruction that reads an SGPR needs five wait states behind such a write, and the compiler's
  // hazard recognizer does not look into inline assembly: without it the load ran with a stale high half of the base)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "+v"(v) : "v"(byte_off), "s"(base) : "memory");
}
// two groups per cluster: 1 = decode the next phase's h in front of this phase's second barrier when it has arrived (see
// `early_decoded` in the kernel); 0 = always at the start of the next phase (rounds 3-4).  Measured (round 5, alternating on
// one box, ms per launch, 0 -> 1): 768 sequences 1.395 -> 1.47, 1 536: 2.62 -> 2.72, 3 072: 5.26 -> 5.41 -- SLOWER: the trace
// shows why: with many sequences the first barrier of a phase is gated by the io waves (the phase's gate tile lands ~1 700
// cycles into it, behind the store acknowledgements queued in front of it), so a decode at the phase start was hidden
// already, while in front of the second barrier it delays the publish (profiles/r5_onchip16_early_decode_rejected.jsonl)
#ifndef ONCHIP16_FWD_EARLY_DECODE
#define ONCHIP16_FWD_EARLY_DECODE 0
#endif
// Experiment builds only (-DONCHIP16_TRACE=1): wave 0 (exchange) and the first io wave of workgroup 0 of the first cluster
// leave `s_memtime` stamps of steps 128 .. 131 of their first work item in the spare words of the exchange header
// (words 96 ..: [wave role 2][step 4][stamp 8] 64-bit ticks + re-poll counts at [64 .. 71] of that block)
#ifndef ONCHIP16_TRACE
#define ONCHIP16_TRACE 0
#endif
// NW = waves per workgroup (round 5).  8: five workgroups of 64 units per cluster, one per CU -- the two waves of a SIMD
// belong to ONE workgroup and stand in the same section of the same phase at all times (both in the MFMAs: they share the
// pipe; both in the cell update; both parked at the barriers).  4: TEN workgroups of 32 units per cluster, TWO per CU from
// different clusters (different sequence groups): the two waves of a SIMD then belong to two INDEPENDENT phase chains, and
// the hardware interleaves one chain's matrix section with the other's exchange / cell / io sections without any
// choreography (MI355X_MICROARCH.md "two waves per SIMD": matrix beside memory is the pairing that nets).  A wave holds the
// same 160 weight registers and runs the same 60 MFMAs per phase either way; the exchange arm is waves 0 .. NW/2 - 1
// (nine peers instead of four per gather, half as many threads), the io arm waves NW/2 .. NW - 1 (four copies + six stores
// per wave and phase as before).
template <int NGA, bool NT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void blstm_onchip16_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ cell, float* __restrict__ hout, int64_t ldo,
    int64_t dstride, const u32x4* __restrict__ wf, unsigned* __restrict__ xhead,
    float* __restrict__ xpayload, int* __restrict__ err, int64_t N, int64_t T, int H, int G, int nclusters,
    int layout) {
  constexpr bool OWN_EARLY = NGA == 1 && ONCHIP16_FWD_OWN_EARLY && !(ONCHIP16_ABL & 12);
  constexpr int UW = 8 * NW;                   // hidden units per workgroup (64 | 32)
  constexpr int NTH = 64 * NW;                 // threads
  constexpr int NEX = NW / 2, NIO = NW / 2;    // exchange waves 0 .. NEX - 1, io waves NEX .. NW - 1
  constexpr int QW = UW / 4;                   // unit quads of the own slice (16 | 8)
  constexpr int QB = UW / 16;                  // 16-unit blocks of the own slice (4 | 2)
  constexpr int GMAX = 40 / NW;                // workgroups per cluster at most (K = 320): 5 | 10
  constexpr int PUBP = UW + 4;                 // floats per LDS row of [seq][unit] scalars (h, c)
  constexpr int TILEW = SQ * UW;               // f32x4 per gate tile (16 | 8 KB)
  constexpr int KOWN = UW / 32;                // k-steps of the own slice (2 | 1)
  const unsigned tagbase = tag16_base(err);
  __shared__ __attribute__((aligned(16))) char hs[NGA * 2 * SQ * HP2];       // [group][hi | lo][seq][k] bf16
  __shared__ __attribute__((aligned(16))) float pub[SQ * PUBP];              // h_t [seq][unit] of the phase
  __shared__ __attribute__((aligned(16))) float cellb[SQ * PUBP];            // c_t
  __shared__ f32x4 ring0[TILEW], ring1[TILEW], ring2[TILEW], ring3[TILEW];    // gate tiles of phases 0..3
  __shared__ int s_fail, s_mem[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_fail = 0;
  layout &= 1;
  const int64_t ng16 = (N + SQ - 1) / SQ;                    // sequence groups
  const int64_t nb_dir = (ng16 + NGA - 1) / NGA;             // bundles per direction
  const Membership mem = join_cluster<true>(xhead, G, (int)(2 * nb_dir), s_mem);
  if (!mem.ok) return;
  const int g = __builtin_amdgcn_readfirstlane(mem.g);
  const int j = lane & 15, up = lane >> 4;                   // MFMA: sequence, unit within the row block
  const int foff = j * HP2 + up * 16;                        // B fragment offset of this lane
  const bool io_wave = wave >= NEX;
  const int s2 = (tid & (64 * NEX - 1)) / QW, uq = tid % QW; // exchange / io thread <-> (sequence s2, unit quad uq)
  const int iow = wave % NIO;                                // (scalar: LDS addresses of the copies stay in SGPRs)
  constexpr unsigned OOR = 0x80000000u;
  int trace_round = -1;
  int trace_spins = 0;
  auto stamp = [&](int role, int64_t st, int idx) __attribute__((always_inline)) {
    if constexpr (ONCHIP16_TRACE) {
      if (trace_round == 0 && st >= 128 && st < 132 && lane == 0 && (wave == 0 || wave == NEX) && g == 0) {
        unsigned long long* tb = reinterpret_cast<unsigned long long*>(xhead + 96);
        tb[(role * 4 + (int)(st - 128)) * 8 + idx] = __builtin_readcyclecounter();
        if (idx == 1 && role == 0) tb[64 + (int)(st - 128)] = (unsigned long long)trace_spins;
      }
    }
  };
  // operand-image columns the gather never writes (k >= 64 G, when H <= 256) must be zero, not stale LDS: 0 x NaN
  for (int i = tid; i < NGA * 2 * SQ * HP2 / 4; i += NTH) reinterpret_cast<unsigned*>(hs)[i] = 0u;

  for (int round = 0;; ++round) {
    const int64_t bundle = next_item<true>(xhead, mem, round, (int)(2 * nb_dir), nclusters, s_mem);
    if (bundle >= 2 * nb_dir) {
      if (bundle > 2 * nb_dir && tid == 0) atomicExch(err, 5);
      break;
    }
    const int dir = (int)(bundle & 1);
    const int64_t sg0 = (bundle >> 1) * NGA;                 // first sequence group of the bundle
    trace_round = bundle == 0 ? 0 : -1;      // (trace builds: the cluster that serves work item 0)
    // stationary weights -> registers
    u32x4 wh[2][KS2], wl[2][KS2];
    {
      const u32x4* wp = wf + ((((int64_t)(dir * G + g) * NW + wave) * 2) * KS2 * 2) * 64 + lane;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          wh[rb][ks] = wp[(int64_t)((rb * KS2 + ks) * 2 + 0) * 64];
          wl[rb][ks] = wp[(int64_t)((rb * KS2 + ks) * 2 + 1) * 64];
        }
    }
    float cst[NGA][2];
#pragma unroll
    for (int p = 0; p < NGA; ++p) cst[p][0] = cst[p][1] = 0.f;
    // ONE group per cluster (small batches): the products of the workgroup's OWN 64 units of h_t (k-steps 0 and 1) are
    // formed right behind the barrier that completes h_t -- from the fp32 values in LDS, rounded as the peers will see
    // them -- while the publish travels; the phase of step t + 1 starts from these sums and multiplies the eight k-steps
    // of the peers' slices only
    f32x4 accn[2];
    accn[0] = accn[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t SN = layout ? 1 : T, ST = layout ? 32 : 1;
    // The per-lane offsets and masks of the io arm are REBUILT in every phase from an opaque copy of the thread index
    // (a dozen vector instructions): as loop invariants the compiler hoisted six masked offsets per group out of the
    // step loop -- 24 registers it did not have -- and reloaded them from scratch in front of every store
    // (`s_waitcnt vmcnt(0)`: the whole asynchronous io arm serialised).
    auto seq0_of = [&](int p) { return (sg0 + p) * SQ; };
    // The gate tile in LDS is [16-unit block q][sequence 16][unit-in-block 16, XOR-swizzled with the sequence] f32x4; an io
    // wave moves it in 1-KB pieces = 4 sequences x 16 units: piece i of its four is block i % QB of sequence chunk
    // iow + NIO (i / QB) (NW = 8: its own chunk of all four blocks; NW = 4: two chunks of the two blocks).  The c / h rows
    // are moved by thread <-> (sequence, unit quad) as in the exchange arm.
    struct IoLane { unsigned goff[4], coff, hoff; bool uok[4]; };
    auto io_piece = [&](int i, int& q, int& c) __attribute__((always_inline)) { q = i % QB; c = iow + NIO * (i / QB); };
    auto io_lane = [&](int p) __attribute__((always_inline)) {
      int tv = tid;
      asm volatile("" : "+v"(tv));
      const int s2v = (tv & (64 * NIO - 1)) / QW, uqv = tv % QW;       // c / h rows
      const unsigned lrv = (unsigned)(s2v * SN);
      IoLane L;
      const bool rok = seq0_of(p) + s2v < N;
      const bool f4 = UW * g + 4 * uqv + 4 <= H;
      L.coff = (rok && f4) ? (lrv * 2u * (unsigned)H + 4u * (unsigned)uqv) * 4u : OOR;
      L.hoff = (rok && f4) ? (lrv * (unsigned)ldo + 4u * (unsigned)uqv) * 4u : OOR;
      const int sl = (tv & 63) >> 4, ub = tv & 15;                      // gate-tile pieces: sequence within the chunk, unit slot
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q, c;
        io_piece(i, q, c);
        const int sv = 4 * c + sl, usw = ub ^ sv;
        L.goff[i] = ((unsigned)(sv * SN) * 2u * (unsigned)H + (unsigned)usw) * 16u + (unsigned)q * 256u;
        L.uok[i] = seq0_of(p) + sv < N && UW * g + 16 * q + usw < H;
      }
      return L;
    };

    // (group, step) of phase i of the loop iteration that starts at step `base`
    auto grp_of = [](int i) { return i % NGA; };
    auto stp_of = [](int64_t base, int i) { return base + i / NGA; };
    auto srd_at = [&](const void* ptr) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7fffffff, 0x00020000);
    };
    auto gates_base = [&](int p, int64_t st) {
      const int64_t t_ = dir ? T - 1 - st : st;
      return reinterpret_cast<char*>(gates) + (((ROW(seq0_of(p), 0) + t_ * ST) * 2 + dir) * (int64_t)H + UW * g) * 16;
    };
    auto ring_of = [&](auto slot_tag) -> f32x4* {
      constexpr int S = decltype(slot_tag)::value;
      return S == 0 ? ring0 : S == 1 ? ring1 : S == 2 ? ring2 : ring3;
    };
    // asynchronous HBM -> LDS copy of the gate tile of (group p, step st) into ring slot S (four instructions)
    auto io_dma = [&](auto slot_tag, int p, int64_t st) __attribute__((always_inline)) {
      f32x4* rg = ring_of(slot_tag);
      const char* gb = gates_base(p, st);
      const IoLane L = io_lane(p);
      // inline assembly, so that the compiler does not know that LDS is written: with the builtin it put a `vmcnt(0)` in
      // front of every read of a gate tile -- also of the tiles that landed phases ago -- which waits for the copy just
      // requested for two phases ahead.  The io waves wait explicitly (`vmcnt(4)` before the barrier that opens a
      // phase).  Scalar base + 32-bit lane offset: no 64-bit per-lane addresses to keep (or spill).  (No immediate
      // offset: the instruction adds it to the LDS address as well as to the global one.)
#define DMA16(I_)                                                                                                          \
      {                                                                                                                \
        /* lanes outside N / H read the block's first bytes (a valid address) into cells nobody uses: the copy is     \
           issued unconditionally, so that every phase queues exactly four copies and six stores (vmcnt arithmetic) */ \
        const unsigned vo = L.uok[I_] ? L.goff[I_] : 0u;                                                               \
        int q_, c_;                                                                                                    \
        io_piece(I_, q_, c_);                                                                                          \
        const int la = __builtin_amdgcn_readfirstlane(                                                                 \
            (int)(uintptr_t)(__attribute__((address_space(3))) void*)&rg[(q_ * 4 + c_) * 64]);                         \
        if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"                         \
                             :: "v"(vo), "s"(gb), "s"(la) : "memory");                                                 \
        else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                               \
                          :: "v"(vo), "s"(gb), "s"(la) : "memory");                                                    \
      }
      DMA16(0) DMA16(1) DMA16(2) DMA16(3)
#undef DMA16
    };
    // flush of (group p, step st): c, h, then the activations of ring slot S (six stores)
    auto io_flush = [&](auto slot_tag, int p, int64_t st) __attribute__((always_inline)) {
      f32x4* rg = ring_of(slot_tag);
      const int64_t t_ = dir ? T - 1 - st : st;
      const int64_t row = ROW(seq0_of(p), 0) + t_ * ST;
      const auto rc = srd_at(reinterpret_cast<char*>(cell) + ((row * 2 + dir) * (int64_t)H + UW * g) * 4);
      const auto rh = srd_at(reinterpret_cast<char*>(hout) + (row * ldo + dir * dstride + UW * g) * 4);
      const auto rs = srd_at(gates_base(p, st));
      const IoLane L = io_lane(p);
      const f32x4 cq = *reinterpret_cast<const f32x4*>(cellb + s2 * PUBP + 4 * uq);
      const f32x4 hq = *reinterpret_cast<const f32x4*>(pub + s2 * PUBP + 4 * uq);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cq), rc, (int)L.coff, 0, NT ? 2 : 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hq), rh, (int)L.hoff, 0, NT ? 2 : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int q, c;
        io_piece(i, q, c);
        const f32x4 v = rg[(q * 4 + c) * 64 + lane];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)(L.uok[i] ? L.goff[i] : OOR), 0, NT ? 2 : 0);
      }
    };
    // exchange: payload of (group p): [slot 2][G][SQ][UPW] values as 8-byte granules of two
    auto payload_srd = [&](int p) {
      const int64_t item = ((sg0 + p) << 1) | dir;
      return __builtin_amdgcn_make_buffer_rsrc(xpayload + item * 2 * G * SQ * UW, 0, 2 * G * SQ * UW * 4, 0x00020000);
    };
    u32x4 vg[GMAX];
    // Two groups per cluster: the NEXT phase's h (requested beside this phase's MFMAs) is decoded into its operand image at
    // the END of this phase's cell update when it has arrived by then -- where the exchange waves otherwise wait at the
    // second barrier for the io waves, which lose the MFMA arbitration and finish ~800 cycles later (trace, round 5) --
    // instead of at the start of the next phase with every other wave parked at its first barrier.  Per wave: a wave
    // whose granules are not all there decodes at the next phase's start as before.
    int early_decoded = 0;
    constexpr bool ASMG = ONCHIP16_ASM_GATHER && NGA == 2 && !(ONCHIP16_ABL & 24);      // (the request beside the MFMAs)
    auto gather_issue = [&](int p, int64_t st) __attribute__((always_inline)) {      // h_{st-1} of group p
      const auto prs = payload_srd(p);
      const int slot = (int)((st - 1) & 1);
      if constexpr (ASMG) {
        const int64_t item = ((sg0 + p) << 1) | dir;
        const float* pb = xpayload + item * 2 * G * SQ * UW;
        int tvg = tid;
        asm volatile("" : "+v"(tvg));
        const unsigned lo = (unsigned)(((tvg & (64 * NEX - 1)) / QW * UW + 4 * (tvg % QW)) * 4);
#pragma unroll
        for (int gs = 0; gs < GMAX; ++gs) {
          if (gs < G) asm_gather_load(vg[gs], pb, lo + (unsigned)((slot * G + gs) * SQ * UW * 4));
          else vg[gs] = u32x4{0u, 0u, 0u, 0u};
        }
        return;
      }
#pragma unroll
      for (int gs = 0; gs < GMAX; ++gs)
        vg[gs] = (gs < G && !(OWN_EARLY && gs == g))
                     ? __builtin_amdgcn_raw_buffer_load_b128(prs, (((slot * G + gs) * SQ + s2) * UW + 4 * uq) * 4, 0, SC1)
                     : u32x4{0u, 0u, 0u, 0u};
    };
    // every granule this lane requested carries the tag of step st (no re-request: the early decode below only looks)
    auto gather_arrived = [&](int64_t st) __attribute__((always_inline)) {
      const unsigned want = mk_tag(tagbase, st);
      bool ok = true;
#pragma unroll
      for (int gs = 0; gs < GMAX; ++gs)
        ok = ok && (gs >= G || (OWN_EARLY && gs == g) || ((vg[gs][0] & 0xffffu) == want && (vg[gs][2] & 0xffffu) == want));
      return ok;
    };
    auto gather_decode = [&](int p) __attribute__((always_inline)) {
      // (the lane part of the operand-image addresses is REBUILT here from an opaque copy of the thread index: as loop
      // invariants the compiler kept one address register per peer -- ten with four-wave workgroups -- spilled them, and
      // reloaded each from scratch in front of its LDS write: eight serialized memory round trips, 2 us per phase)
      int tvd = tid;
      asm volatile("" : "+v"(tvd));
      const int s2d = (tvd & (64 * NEX - 1)) / QW, uqd = tvd % QW;
      char* hh = hs + (p * 2 + 0) * SQ * HP2 + s2d * HP2 + 8 * uqd;
      char* hl = hs + (p * 2 + 1) * SQ * HP2 + s2d * HP2 + 8 * uqd;
#pragma unroll
      for (int gs = 0; gs < GMAX; ++gs) {
        if (gs < G && !(OWN_EARLY && gs == g)) {
          const int k = __builtin_amdgcn_readfirstlane(UW * (gs >= g ? gs - g : gs - g + G));      // (the k axis starts at the own slice)
          const u32x4 w = vg[gs];
          unsigned h0, l0, h1, l1;
          split2(granule_a(w[0], w[1]), granule_b(w[1]), h0, l0);
          split2(granule_a(w[2], w[3]), granule_b(w[3]), h1, l1);
          *reinterpret_cast<u32x2*>(hh + 2 * k) = u32x2{h0, h1};
          *reinterpret_cast<u32x2*>(hl + 2 * k) = u32x2{l0, l1};
        }
      }
    };
    auto gather_finish = [&](int p, int64_t st) __attribute__((always_inline)) {
      const auto prs = payload_srd(p);
      const int slot = (int)((st - 1) & 1);
      const unsigned want = mk_tag(tagbase, st);
      int spins = 0;
      bool fail = false;
      if constexpr (ASMG) {
        // issued since the requests: exactly ONE publish store (every phase publishes); scratch traffic, if any, only adds
        // operations behind the loads -- the count stays a lower bound
        static_assert(GMAX == 5 || GMAX == 10, "operand list below");
        if constexpr (GMAX == 5)
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]), "+v"(vg[4]) :: "memory");
        else
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]), "+v"(vg[4]), "+v"(vg[5]),
                       "+v"(vg[6]), "+v"(vg[7]), "+v"(vg[8]), "+v"(vg[9]) :: "memory");
      }
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int gs = 0; gs < GMAX; ++gs)
          ok = ok && (gs >= G || (OWN_EARLY && gs == g) || ((vg[gs][0] & 0xffffu) == want && (vg[gs][2] & 0xffffu) == want));
        if (ok || (ONCHIP16_ABL & 16)) break;
        if (++spins > SPIN_LIMIT) { fail = true; break; }
        if constexpr (ONCHIP16_TRACE) trace_spins = spins;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int gs = 0; gs < GMAX; ++gs)
          if (gs < G && !(OWN_EARLY && gs == g) && !((vg[gs][0] & 0xffffu) == want && (vg[gs][2] & 0xffffu) == want))
            vg[gs] = __builtin_amdgcn_raw_buffer_load_b128(prs, (((slot * G + gs) * SQ + s2) * UW + 4 * uq) * 4, 0, SC1);
      }
      if (fail) s_fail = 1;
      if constexpr (ONCHIP16_TRACE) {      // the peers' values have arrived (decode follows)
        if (trace_round == 0 && st >= 128 && st < 132 && lane == 0 && wave == 0 && g == 0)
          reinterpret_cast<unsigned long long*>(xhead + 96)[68 + (int)(st - 128)] = __builtin_readcyclecounter();
      }
      gather_decode(p);
    };
    auto publish = [&](int p, int64_t st) __attribute__((always_inline)) {
      const auto prs = payload_srd(p);
      const unsigned tag = mk_tag(tagbase, st + 1);
      const int slot = (int)(st & 1);
      int tvp = tid;                      // (lane offsets rebuilt from an opaque copy: see gather_finish)
      asm volatile("" : "+v"(tvp));
      const int s2p = (tvp & (64 * NEX - 1)) / QW, uqp = tvp % QW;
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pub + s2p * PUBP + 4 * uqp);
      const u32x2 ga = pack_granule(tag, pv[0], pv[1]), gb = pack_granule(tag, pv[2], pv[3]);
      const int wgoff = __builtin_amdgcn_readfirstlane((slot * G + g) * SQ * UW * 4);
      // (the workgroup's offset rides in the LANE offset, soffset stays the immediate 0: a > 64-bit buffer store with a register
      // soffset is not guarded against a VALU write of its data -- tools/scan_store_hazard.py)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{ga[0], ga[1], gb[0], gb[1]}, prs, (s2p * UW + 4 * uqp) * 4 + wgoff, 0, SC0);
    };

    // ---- one phase: ring slot S (static), group P (static), step st.  IO: the role of this wave -- the step loop
    // exists once per role (below), so that neither the registers nor the compiler's wait-count bookkeeping of one
    // role meet the other's at a join (a merged loop copied the prefetched gather registers behind the join, i.e.
    // waited for the loads it had just issued)
    auto phase = [&](auto io_tag, auto slot_tag, auto grp_tag, int64_t st, int64_t base) __attribute__((always_inline)) {
      constexpr bool IO = decltype(io_tag)::value;
      constexpr int S = decltype(slot_tag)::value, P = decltype(grp_tag)::value;
      if (st >= T) return;                                     // (T tail of NGA < 4; uniform)
      constexpr int FGAT = NGA == 2 ? ONCHIP16_FWD_GATHER : 0;
      f32x4* rg = ring_of(slot_tag);
      if constexpr (ONCHIP16_TRACE) trace_spins = 0;
      stamp(IO ? 1 : 0, st, 0);
      if constexpr (!IO) {
        // (NGA = 4: this phase's h was decoded into the operand image behind the previous phase's publish -- below --
        // so nothing stands between the exchange waves and the barrier)
        if (NGA < 4 && st > 0 && !(ONCHIP16_ABL & 8) && !early_decoded) gather_finish(P, st);
        early_decoded = 0;
      } else {
        // the tile of this phase has landed: it was requested two phases ago, and exactly six stores + four copies
        // (the previous phase's) were queued behind it -- those may still be in flight (a store acknowledgement
        // takes longer than a phase under load: waiting for them paced the whole kernel)
        // (one group: the copies are requested at the START of a phase -- below -- so six more stores queue behind them)
        if (NGA == 1 && ONCHIP16_FWD_EARLY_DMA) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      }
      stamp(IO ? 1 : 0, st, 1);
      lds_barrier();
      stamp(IO ? 1 : 0, st, 2);
      if (s_fail) return;
      // (group, step) of the next phase and of the phase after it
      constexpr int I1 = (S + 1) & 3, I2 = (S + 2) & 3;
      const int64_t b1 = I1 == 0 ? base + 4 / NGA : base, b2 = I2 < 2 ? base + 4 / NGA : base;
      constexpr int P1 = I1 % NGA, P2 = I2 % NGA;
      const int64_t st1 = b1 + I1 / NGA, st2 = b2 + I2 / NGA;
      if constexpr (IO && NGA == 1 && ONCHIP16_FWD_EARLY_DMA) {
        // ONE group per cluster (small batches): the exchange chain is the critical path and everything the io waves do
        // between two phases delays the publish -> the tile of phase + 2 is requested here, beside the MFMAs (its slot was
        // flushed two phases ago): 0.758 -> 0.724 ms at 32 sequences, 1.79 -> 1.52 at 768.  With two groups the same change
        // costs 8-15 % (profiles/r3_onchip16_early_dma.jsonl)
        if (!(ONCHIP16_ABL & 4)) io_dma(std::integral_constant<int, I2>{}, P2, st2 < T ? st2 : T - 1);
      }
      if constexpr (!IO && FGAT >= 1) {
        if (st1 > 0 && st1 < T && (FGAT == 1 || st == 0) && !(ONCHIP16_ABL & 8)) gather_issue(P1, st1);
      }
      f32x4 acc[2];
      acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int KS0 = OWN_EARLY ? KOWN : 0;   // (OWN_EARLY: the k-steps of the own slice are in accn already)
      if constexpr (OWN_EARLY) { acc[0] = accn[0]; acc[1] = accn[1]; }
      if (st > 0 && !(ONCHIP16_ABL & 1)) {
        const char* hh = hs + (P * 2 + 0) * SQ * HP2 + foff + KS0 * 64;
        const char* hl = hs + (P * 2 + 1) * SQ * HP2 + foff + KS0 * 64;
        // fragments one k-step ahead, pinned: hoisting all twenty reads would cost 80 registers the kernel does not have
        bf16x8 bh = *reinterpret_cast<const bf16x8*>(hh), bl = *reinterpret_cast<const bf16x8*>(hl);
#pragma unroll
        for (int ks = KS0; ks < KS2; ++ks) {
          bf16x8 nh = bh, nl = bl;
          if (ks + 1 < KS2) {
            nh = *reinterpret_cast<const bf16x8*>(hh + (ks + 1 - KS0) * 64);
            nl = *reinterpret_cast<const bf16x8*>(hl + (ks + 1 - KS0) * 64);
          }
          __builtin_amdgcn_sched_barrier(0);
          constexpr int NRB = (ONCHIP16_ABL & 32) ? 1 : 2;
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[rb] = MFMA16_BF16(as_bf16x8(wl[rb][ks]), bh, acc[rb]);
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[rb] = MFMA16_BF16(as_bf16x8(wh[rb][ks]), bl, acc[rb]);
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[rb] = MFMA16_BF16(as_bf16x8(wh[rb][ks]), bh, acc[rb]);
          __builtin_amdgcn_sched_barrier(0);
          bh = nh; bl = nl;
          if constexpr (!IO && FGAT >= 2) {
            if (ks == FGAT - 2 && st1 > 0 && st1 < T && !(ONCHIP16_ABL & 8)) gather_issue(P1, st1);
          }
        }
      }
      stamp(IO ? 1 : 0, st, 3);
      // cell update: lane (sequence j, unit 8 wave + 4 rb + up), the four gates in acc[rb]
      // (addresses rebuilt from an opaque copy of the lane index: hoisted out of the loop they were eight more
      // registers -- one per ring slot and row block -- that ended up in scratch)
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const int jv = lv & 15, upv = lv >> 4;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int ul = 8 * wave + 4 * rb + upv;
        f32x4* xp = &rg[(ul >> 4) * 256 + jv * 16 + ((ul & 15) ^ jv)];
        const f32x4 gx = *xp;
        const float a0 = acc[rb][0] + gx[0], a1 = acc[rb][1] + gx[1], a2 = acc[rb][2] + gx[2], a3 = acc[rb][3] + gx[3];
        const bool fake = (ONCHIP16_ABL & 2) != 0;
        const float ig = fake ? a0 : fast_sigmoid(a0), fg = fake ? a1 : fast_sigmoid(a1), gg = fake ? a2 : fast_tanh(a2), og = fake ? a3 : fast_sigmoid(a3);
        const float cn = fg * cst[P][rb] + ig * gg;
        cst[P][rb] = cn;
        *xp = f32x4{ig, fg, gg, og};
        pub[jv * PUBP + ul] = (UW * g + ul < H) ? og * (fake ? cn : fast_tanh(cn)) : 0.f;
        cellb[jv * PUBP + ul] = cn;
      }
      stamp(IO ? 1 : 0, st, 4);
      if constexpr (!IO && NGA == 2 && ONCHIP16_FWD_EARLY_DECODE) {
        if (st1 > 0 && st1 < T && !(ONCHIP16_ABL & 8)) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the requests of k-step FGAT - 2: long back)
          if (__builtin_amdgcn_ballot_w64(!gather_arrived(st1)) == 0) {
            gather_decode(P1);
            early_decoded = 1;
          }
        }
      }
      lds_barrier();
      stamp(IO ? 1 : 0, st, 5);
      // OWN_EARLY: h_t of the own slice x its weight columns, for step t + 1 (every wave: it needs the whole slice as B)
      auto own_products = [&]() __attribute__((always_inline)) {
        accn[0] = accn[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (st + 1 < T && !(ONCHIP16_ABL & 1)) {
          auto r24 = [](float x) { return __uint_as_float((__float_as_uint(x) + 0x80u) & 0xffffff00u); };      // = the granules
#pragma unroll
          for (int ks = 0; ks < KOWN; ++ks) {
            const float* pr = pub + jv * PUBP + 32 * ks + 8 * upv;
            const f32x4 a = *reinterpret_cast<const f32x4*>(pr), b = *reinterpret_cast<const f32x4*>(pr + 4);
            unsigned h0, l0, h1, l1, h2, l2, h3, l3;
            split2(r24(a[0]), r24(a[1]), h0, l0);
            split2(r24(a[2]), r24(a[3]), h1, l1);
            split2(r24(b[0]), r24(b[1]), h2, l2);
            split2(r24(b[2]), r24(b[3]), h3, l3);
            const bf16x8 bh = as_bf16x8(u32x4{h0, h1, h2, h3}), bl = as_bf16x8(u32x4{l0, l1, l2, l3});
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) accn[rb] = MFMA16_BF16(as_bf16x8(wl[rb][ks]), bh, accn[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) accn[rb] = MFMA16_BF16(as_bf16x8(wh[rb][ks]), bl, accn[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) accn[rb] = MFMA16_BF16(as_bf16x8(wh[rb][ks]), bh, accn[rb]);
          }
        }
      };
      if constexpr (!IO) {
        if (!(ONCHIP16_ABL & 8)) {
          publish(P, st);
          if constexpr (OWN_EARLY) own_products();
          stamp(0, st, 6);
          if (NGA < 4) {
            // (two groups: the next phase's h is decoded in front of its barrier -- decoding it here, behind the publish,
            // measured slower even with the early request: 1.42 -> 1.50 ms at 768 sequences, 5.5 -> 5.9 at 3 072)
            if (FGAT == 0 && st1 > 0 && st1 < T) {
              // (one group: the peers publish these values at this very moment -- a request issued at once finds stale tags
              // and pays a second round trip; 14 x 64 cycles later it usually finds them: 0.722 -> 0.685 ms at 32 sequences,
              // 0.743 -> 0.70 at 160; 8: 0.70, 20: 0.72, 28: 0.78 -- profiles/r3_onchip16_g1_delay.jsonl)
              if (NGA == 1 && !OWN_EARLY && ONCHIP16_FWD_G1_DELAY) __builtin_amdgcn_s_sleep(ONCHIP16_FWD_G1_DELAY);
              if (OWN_EARLY && ONCHIP16_FWD_G1_DELAY_OWN) __builtin_amdgcn_s_sleep(ONCHIP16_FWD_G1_DELAY_OWN);
              gather_issue(P1, st1);
            }
          } else {
            // the NEXT phase's h (requested one phase ago) -> its operand image, while the io waves flush; then the
            // request for the phase after that: two phases of latency budget per gather
            if (st1 > 0 && st1 < T) gather_finish(P1, st1);
            if (st2 > 0 && st2 < T) gather_issue(P2, st2);
          }
        }
      } else if (!(ONCHIP16_ABL & 4)) {
        // (storing the activations a phase later, beside the next phase's MFMAs -- what the backward does with its d(gates) --
        // measured SLOWER here: 1.37 -> 1.49 ms at 768 sequences, 5.5 -> 5.9 at 3 072)
        io_flush(slot_tag, P, st);
        if constexpr (OWN_EARLY) own_products();
        stamp(1, st, 6);
        // (beyond the last step the copy is repeated for step T - 1 into a slot nobody reads any more: every phase
        // queues exactly four copies, or `vmcnt(10)` above would not cover the tiles of the last phases)
        if constexpr (!(NGA == 1 && ONCHIP16_FWD_EARLY_DMA)) io_dma(std::integral_constant<int, I2>{}, P2, st2 < T ? st2 : T - 1);
      }
      stamp(IO ? 1 : 0, st, 7);
    };
    auto run = [&](auto io_tag) __attribute__((always_inline)) {
      constexpr bool IO = decltype(io_tag)::value;
      if constexpr (IO) {      // prologue: the tiles of phases 0 and 1
        io_dma(std::integral_constant<int, 0>{}, grp_of(0), stp_of(0, 0));
        if (stp_of(0, 1) < T) io_dma(std::integral_constant<int, 1>{}, grp_of(1), stp_of(0, 1));
      }
      // the stationary weights have arrived before the loop (inputs of an empty asm: the compiler waits HERE, and its
      // wait-count bookkeeping enters the loop with nothing pending -- otherwise the first MFMA of every phase carries
      // a `vmcnt(0)` that drains the io arm)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) asm volatile("" :: "v"(wh[rb][ks]), "v"(wl[rb][ks]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int64_t base = 0; base < T; base += 4 / NGA) {
        phase(io_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 0 % NGA>{}, base + 0 / NGA, base);
        phase(io_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NGA>{}, base + 1 / NGA, base);
        phase(io_tag, std::integral_constant<int, 2>{}, std::integral_constant<int, 2 % NGA>{}, base + 2 / NGA, base);
        phase(io_tag, std::integral_constant<int, 3>{}, std::integral_constant<int, 3 % NGA>{}, base + 3 / NGA, base);
        if (s_fail) break;
      }
    };
    if (io_wave) run(std::true_type{}); else run(std::false_type{});
    if (s_fail) {
      if (tid == 0) atomicExch(err, 3);
      return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
}


// =====================================================================================================
// Backward recurrence, interleaved variant (round 3): the rotation of blstm_onchip16_fwd_kernel applied to the
// reduce-scatter of dh, with the forward's division of labour -- waves 0-3 exchange, waves 4-7 move HBM data.
//  * `v_mfma_f32_16x16x32_bf16`: A = 16 output units x 32 own gate columns, B = 32 gate columns x 16 sequences.  320
//    output units = 20 tiles; wave w owns tiles 2w, 2w + 1 over all 8 k-steps and k-steps 4 (w & 1) .. + 3 of shared
//    tile 16 + (w >> 1): 20 fragments x (hi, lo) = 160 registers, 60 MFMAs per wave and phase;
//  * a ring of four LDS slots per phase {gate activations 16 KB, c_(t-1) 4 KB, dh 4 KB}, filled by asynchronous copies
//    two phases ahead (inline assembly + `vmcnt(10)`, as in the forward); the cell backward (thread <-> sequence
//    tid / 32, unit pair tid % 32) overwrites the gate tile with d(gates), which the io waves flush a phase later;
//  * the exchange waves add the partial sums of the other workgroups (requested a phase ahead, compiler-visible loads)
//    and the own partial (LDS, per group) into the dh tile before the barrier that opens the phase, and publish this
//    phase's partial sums behind the MFMAs;
//  * per-group state in registers: the dc chain and c_t of the two cells of a lane.
// Three barriers per phase (dh complete | d(gates) image complete | partial sums complete).
// Where the exchange waves request the partial sums of the NEXT phase: 0 = behind this phase's publish (the loads'
// round trip through L2 then sits in front of the next phase's first barrier), 1 = behind the first barrier, 2 = behind
// the second (in front of the MFMAs: the round trip hides behind them), k >= 3 = after k MFMA k-steps.  The peers
// published those sums a phase earlier -- with ONE group per cluster they are this phase's own, so an early request only
// finds stale tags and is repeated (7.8 -> 8.9 ms): position 0 there.  Two groups, 3 072 sequences: 7.1 -> 6.7 ms
// (positions 2, 3, 5 equal; 1: 6.96), 768: 1.74 -> 1.56 (profiles/r3_onchip16_bwd_gather.jsonl).
// 1: the partial sums of a wave's own two tiles are published straight from its accumulators behind its MFMAs (64-byte
// runs per sequence); only the shared tiles (block 4) still go through LDS and the barrier
#ifndef ONCHIP16_BWD_DIRECT
#define ONCHIP16_BWD_DIRECT 1
#endif
// 1: the io waves store a phase's d(gates) during the NEXT phase's MFMA section instead of right behind its last barrier
// (two groups and more; with ONE group the publish chain is the critical path and the stores in front of the io waves'
// MFMAs delay it: 0.74 -> 0.77 ms at 32 sequences)
// 1 (two groups and more, with the deferred flush): the d(gates) of the previous phase are stored by the EXCHANGE waves,
// not by the io waves.  The trace of round 5 shows the io waves on the critical path of every section of a backward phase
// (tiles -> cell backward -> four stores -> MFMAs, which they start a thousand cycles late -> publish -> six LDS copies:
// ~7 100 of a 7 600-cycle phase) while the exchange waves idle ~2 500 cycles at the barriers.  With the stores behind the
// exchange waves' gather requests the in-order counter needs the exact wait in front of the tag check (vmcnt(6): four
// stores + the two direct-publish stores issued since), i.e. the inline-assembly gather of ONCHIP16_ASM_GATHER.
// Measured (round 5, alternating builds on one box): 3 072 sequences 6.04 -> 5.95 ms (-1.4 %), 768 / 1 536 within noise:
// WHO issues the stores hardly matters -- the launch moves its 28.9 GB at the box's copy rate either way.  Default 0 (the
// inline-assembly gather is not worth 0.15 ms per step); profiles/r5_onchip16_tails.jsonl.
#ifndef ONCHIP16_BWD_FLUSH_BY_EXCHANGE
#define ONCHIP16_BWD_FLUSH_BY_EXCHANGE 0
#endif
#ifndef ONCHIP16_BWD_DEFER_FLUSH
#define ONCHIP16_BWD_DEFER_FLUSH 1
#endif
// ONE group per cluster: the tiles of phase + 2 are requested in the MFMA section (see the forward): 0.88 -> 0.76 ms at 160
// sequences, 1.80 -> 1.59 at 768, equal at 32; with two groups it costs 6 % (profiles/r3_onchip16_early_dma.jsonl)
#ifndef ONCHIP16_BWD_EARLY_DMA_G2
#define ONCHIP16_BWD_EARLY_DMA_G2 0      // (the same for two groups and more: 6.15 -> 7.91 ms at 3 072 sequences, round 5 -- the copies'
                                         //  issue stalls block the io waves' MFMAs; profiles/r5_onchip16_tails.jsonl)
#endif
#ifndef ONCHIP16_BWD_EARLY_DMA_G1
#define ONCHIP16_BWD_EARLY_DMA_G1 1
#endif
#ifndef ONCHIP16_BWD_GATHER
#define ONCHIP16_BWD_GATHER 2
#endif
// 1: the unit axis of every workgroup is rotated so that its OWN 64 output units are the block of the shared tiles: all
// sixteen whole tiles of the waves then belong to peers and leave straight from the accumulators, the shared tiles (own
// destination) stay in LDS -- no publish behind the third barrier any more.  ONE group per cluster in addition forms the own
// block's products AFTER the peers' tiles have left (the step chain waits for the peers' data, not for its own)
#ifndef ONCHIP16_BWD_OWN_SHARED
#define ONCHIP16_BWD_OWN_SHARED 1
#endif
constexpr int DP2 = 256 * 2 + 16;            // 528 B per sequence row of the bf16 d(gates) image (conflict-free: 4 mod 64 dwords)
constexpr int PP2 = 5 * UPW + 4;             // floats per row of partial dh
constexpr int OWNP = UPW + 4;                // floats per row of the own slice

__global__ void pack_onchip16_bwd_kernel(const float* w_hh_f, const float* w_hh_r, int H, int G, u32x4* wb) {
  // wb[dir][g][wave 8][frag 20][hl 2][lane 64]: frag f < 16: tile 2 wave + f / 8, k-step f % 8; f >= 16: shared tile
  // 16 + (wave >> 1), k-step 4 (wave & 1) + f - 16.  Lane (i = lane % 16, kg = lane / 16): output unit 16 tile + i,
  // own gate column c = 32 ks + 8 kg + j = 4 ul + gate of unit 64 g + ul
  const int64_t n = (int64_t)2 * G * 8 * 20 * 2 * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int lane = (int)(r & 63); r >>= 6;
    const int hl = (int)(r & 1); r >>= 1;
    const int f = (int)(r % 20); r /= 20;
    const int wave = (int)(r & 7); r >>= 3;
    const int g = (int)(r % G);
    const int d = (int)(r / G);
    const int tile = f < 16 ? 2 * wave + f / 8 : 16 + (wave >> 1);
    const int ks = f < 16 ? f % 8 : 4 * (wave & 1) + f - 16;
    const int i = lane & 15, kg = lane >> 4;
    // (ONCHIP16_BWD_OWN_SHARED: tile block b' of workgroup g holds the units of workgroup (b' + g + 1) mod G -- block 4,
    // the shared tiles, its own)
    const int uo = ONCHIP16_BWD_OWN_SHARED ? 64 * ((tile / 4 + g + 1) % G) + 16 * (tile % 4) + i : 16 * tile + i;
    const float* w = d ? w_hh_r : w_hh_f;
    float x[8];
    for (int j = 0; j < 8; ++j) {
      const int c = 32 * ks + 8 * kg + j;
      const int ui = 64 * g + (c >> 2), gate = c & 3;
      x[j] = (ui < H && uo < H) ? w[(int64_t)(gate * H + ui) * H + uo] : 0.f;
    }
    unsigned h[4], l[4];
    for (int j = 0; j < 4; ++j) split2(x[2 * j], x[2 * j + 1], h[j], l[j]);
    wb[e] = hl ? u32x4{l[0], l[1], l[2], l[3]} : u32x4{h[0], h[1], h[2], h[3]};
  }
}

template <int NGA, bool NT>
__global__ __launch_bounds__(512, 2) void blstm_onchip16_bwd_kernel(
    float* __restrict__ gates, const float* __restrict__ cell, const float* __restrict__ dhout, int64_t ldo,
    int64_t dstride, const u32x4* __restrict__ wb, unsigned* __restrict__ xhead, float* __restrict__ xpayload,
    int* __restrict__ err, int64_t N, int64_t T, int H, int G, int nclusters, int layout) {
  const unsigned tagbase = tag16_base(err);
  __shared__ f32x4 ringg[4][TILE4];                                          // gate activations -> d(gates)
  __shared__ __attribute__((aligned(16))) float ringc[4][SQ * UPW];         // c_(t-1)  [seq][unit]
  __shared__ __attribute__((aligned(16))) float ringd[4][SQ * UPW];         // dh (+ partial sums)
  __shared__ __attribute__((aligned(16))) char dg_hi[SQ * DP2];
  __shared__ __attribute__((aligned(16))) char dg_lo[SQ * DP2];
  // [seq][unit] partial dh, k half 0 of the shared tiles (with the direct publish only those: units 256 .. 319)
  constexpr int PSW = ONCHIP16_BWD_DIRECT ? OWNP : PP2, PSO = ONCHIP16_BWD_DIRECT ? 0 : 256;
  __shared__ __attribute__((aligned(16))) float psum[SQ * PSW];
  // the peers' + the own partial sums of this phase's dh, added to the dh tile by the cell backward.  (NOT added into the
  // ring slot by the exchange waves: they run in front of the barrier that follows the io waves' `vmcnt` wait, i.e.
  // nothing orders their read of the slot behind the asynchronous copy that fills it)
  __shared__ __attribute__((aligned(16))) float dsum[SQ * UPW];
  __shared__ __attribute__((aligned(16))) float psum2[SQ * OWNP];           // shared tiles, k half 1: [seq][unit - 256]
  __shared__ __attribute__((aligned(16))) float pown[NGA * SQ * OWNP];      // own 64 units of every group
  __shared__ int s_fail, s_mem[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_fail = 0;
  layout &= 1;
  const int64_t ng16 = (N + SQ - 1) / SQ;
  const int64_t nb_dir = (ng16 + NGA - 1) / NGA;
  const Membership mem = join_cluster<true>(xhead, G, (int)(2 * nb_dir), s_mem);
  if (!mem.ok) return;
  const int g = __builtin_amdgcn_readfirstlane(mem.g);
  const int Hp = G * UPW;
  const bool io_wave = wave >= 4;
  const int iow = wave & 3;
  constexpr unsigned OOR = 0x80000000u;
  // (trace builds: stamps of steps 128, 129 of work item 0, workgroup 0: [role 2][step 2][index 10] at words 96 ..)
  int trace_on = 0;
  auto stamp = [&](int role, int64_t st, int idx) __attribute__((always_inline)) {
    if constexpr (ONCHIP16_TRACE) {
      if (trace_on && st >= 128 && st < 130 && lane == 0 && (wave == 0 || wave == 4) && g == 0)
        reinterpret_cast<unsigned long long*>(xhead + 96)[(role * 2 + (int)(st - 128)) * 10 + idx] = __builtin_readcyclecounter();
    }
  };

  for (int round = 0;; ++round) {
    const int64_t bundle = next_item<true>(xhead, mem, round, (int)(2 * nb_dir), nclusters, s_mem);
    if (bundle >= 2 * nb_dir) {
      if (bundle > 2 * nb_dir && tid == 0) atomicExch(err, 5);
      break;
    }
    const int dir = (int)(bundle & 1);
    const int64_t sg0 = (bundle >> 1) * NGA;
    trace_on = bundle == 0;
    u32x4 wh[20], wl[20];
    {
      const u32x4* wp = wb + (((int64_t)(dir * G + g) * 8 + wave) * 20 * 2) * 64 + lane;
#pragma unroll
      for (int f = 0; f < 20; ++f) {
        wh[f] = wp[(int64_t)(f * 2 + 0) * 64];
        wl[f] = wp[(int64_t)(f * 2 + 1) * 64];
      }
    }
    const int64_t SN = layout ? 1 : T, ST = layout ? 32 : 1;
    auto seq0_of = [&](int p) { return (sg0 + p) * SQ; };
    auto t_of = [&](int64_t st) { return dir ? st : T - 1 - st; };
    auto rowb = [&](int p, int64_t t_) { return ROW(seq0_of(p), 0) + t_ * ST; };
    auto srd_at = [&](const void* ptr) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7fffffff, 0x00020000);
    };
    auto payload_srd = [&](int p) {
      const int64_t item = ((sg0 + p) << 1) | dir;
      return __builtin_amdgcn_make_buffer_rsrc(xpayload + item * 2 * G * SQ * Hp, 0, 2 * G * SQ * Hp * 4, 0x00020000);
    };
    // carries of this lane's two cells (sequence tid / 32, units 2 (tid % 32), + 1) per group: dc chain and c_t
    float dcr[NGA][2], ctc[NGA][2];
#pragma unroll
    for (int p = 0; p < NGA; ++p) {
      dcr[p][0] = dcr[p][1] = 0.f;
      const int s = tid >> 5, up = tid & 31;
      const bool ok = seq0_of(p) + s < N && 64 * g + 2 * up < H;
      const float2 c0 = ok ? *reinterpret_cast<const float2*>(cell + ((rowb(p, t_of(0)) + s * SN) * 2 + dir) * (int64_t)H + 64 * g + 2 * up)
                           : make_float2(0.f, 0.f);
      ctc[p][0] = c0.x; ctc[p][1] = c0.y;
    }

    // ---- io arm (waves 4-7): thread <-> (row s2, unit 16 q + (uq ^ s2)) for the gate tile, (row s2, units 4 uq ..) for c / dh
    struct IoLane { unsigned goff0, coff, hoff; bool uok[4], rok, f4; };
    auto io_lane = [&](int p) __attribute__((always_inline)) {
      int tv = tid;
      asm volatile("" : "+v"(tv));
      const int s2v = (tv & 255) >> 4, uqv = tv & 15, uswv = uqv ^ s2v;
      const unsigned lrv = (unsigned)(s2v * SN);
      IoLane L;
      L.rok = seq0_of(p) + s2v < N;
      L.f4 = L.rok && 64 * g + 4 * uqv + 4 <= H;
      L.goff0 = (lrv * 2u * (unsigned)H + (unsigned)uswv) * 16u;
      L.coff = (lrv * 2u * (unsigned)H + 4u * (unsigned)uqv) * 4u;
      L.hoff = (lrv * (unsigned)ldo + 4u * (unsigned)uqv) * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) L.uok[q] = L.rok && 64 * g + 16 * q + uswv < H;
      return L;
    };
    auto dma16 = [&](const char* base, unsigned voff, const void* ldsp) __attribute__((always_inline)) {
      const int la = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) void*)ldsp);
      if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(base), "s"(la) : "memory");
      else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(base), "s"(la) : "memory");
    };
    // the three tiles of (group p, step st) -> ring slot S: six copies, issued unconditionally (lanes outside N / H
    // copy the block's first bytes into cells nobody uses)
    auto io_dma = [&](int S, int p, int64_t st) __attribute__((always_inline)) {
      const IoLane L = io_lane(p);
      const int64_t t_ = t_of(st);
      const int64_t tp_ = st + 1 < T ? (dir ? t_ + 1 : t_ - 1) : t_;
      const char* gb = reinterpret_cast<const char*>(gates) + ((rowb(p, t_) * 2 + dir) * (int64_t)H + 64 * g) * 16;
      const char* cb = reinterpret_cast<const char*>(cell) + ((rowb(p, tp_) * 2 + dir) * (int64_t)H + 64 * g) * 4;
      const char* hb = reinterpret_cast<const char*>(dhout) + (rowb(p, t_) * ldo + dir * dstride + 64 * g) * 4;
      if (ONCHIP16_ABL & (4 | 128)) return;      // (experiment builds: no tile copies)
#pragma unroll
      for (int q = 0; q < ((ONCHIP16_ABL & 256) ? 2 : 4); ++q) dma16(gb, L.uok[q] ? L.goff0 + q * 256 : 0u, &ringg[S][(q * 4 + iow) * 64]);      // (256: half of the gate tile -- what 8-byte activations would move)
      dma16(cb, L.f4 ? L.coff : 0u, &ringc[S][iow * 256]);
      dma16(hb, L.f4 ? L.hoff : 0u, &ringd[S][iow * 256]);
    };
    // d(gates) of (group p, step st) from ring slot S -> HBM: four stores
    // (`live` = false: the same four stores out of range -- the vector-memory count per phase must not depend on the step)
    auto io_flush = [&](int S, int p, int64_t st, bool live) __attribute__((always_inline)) {
      const IoLane L = io_lane(p);
      const auto rs = srd_at(reinterpret_cast<char*>(gates) + ((rowb(p, t_of(st)) * 2 + dir) * (int64_t)H + 64 * g) * 16);
      if (ONCHIP16_ABL & (4 | 64)) return;       // (experiment builds: no d(gates) stores)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int lvf = lane;
        asm volatile("" : "+v"(lvf));
        const f32x4 v = ringg[S][(q * 4 + iow) * 64 + lvf];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)((live && L.uok[q]) ? L.goff0 + q * 256 : OOR), 0, NT ? 2 : 0);
      }
    };

    // ---- exchange arm (waves 0-3): thread <-> (sequence s2, own unit quad uq)
    u32x4 vg[4];
    int srcw[4];                   // the G - 1 other workgroups, ascending (G = 5)
#pragma unroll
    for (int i = 0; i < 4; ++i) srcw[i] = i + (i >= g ? 1 : 0);
    // (per-lane indices rebuilt from an opaque copy of the thread index in every use: see the forward kernel)
#define EX_LANE() int tvx = tid; asm volatile("" : "+v"(tvx)); const int s2 = (tvx & 255) >> 4, uq = tvx & 15
    // (the request behind the second barrier, two stores -- the direct publish -- behind it: see ONCHIP16_ASM_GATHER)
    constexpr bool FBX = ONCHIP16_BWD_FLUSH_BY_EXCHANGE && NGA >= 2 && ONCHIP16_BWD_DEFER_FLUSH && ONCHIP16_BWD_GATHER == 2 &&
                         ONCHIP16_BWD_DIRECT && ONCHIP16_BWD_OWN_SHARED;
    constexpr bool ASMG = (ONCHIP16_ASM_GATHER || FBX) && NGA >= 2 && ONCHIP16_BWD_GATHER == 2 && ONCHIP16_BWD_DIRECT && ONCHIP16_BWD_OWN_SHARED;
    auto gather_issue = [&](int p, int64_t st) __attribute__((always_inline)) {      // partial dh for step st of group p
      const auto prs = payload_srd(p);
      const int slot = (int)((st - 1) & 1);
      EX_LANE();
      if constexpr (ASMG) {
        const int64_t item = ((sg0 + p) << 1) | dir;
        const float* pb = xpayload + item * 2 * G * SQ * Hp;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm_gather_load(vg[i], pb, (unsigned)((((slot * G + srcw[i]) * SQ + s2) * Hp + 64 * g + 4 * uq) * 4));
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        vg[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, (((slot * G + srcw[i]) * SQ + s2) * Hp + 64 * g + 4 * uq) * 4, 0, SC1);
    };
    auto gather_finish = [&](int S, int p, int64_t st) __attribute__((always_inline)) {
      const auto prs = payload_srd(p);
      const int slot = (int)((st - 1) & 1);
      const unsigned want = mk_tag(tagbase, st);
      EX_LANE();
      int spins = 0;
      bool fail = false;
      if constexpr (ASMG) {
        // issued since the requests: the two direct-publish stores of the requesting phase -- unless that phase was a group's
        // LAST step (no partial sums to publish): st + 1 < T is a lower bound for both groups' predecessors
        // (FBX: + the four d(gates) stores of the requesting phase, issued unconditionally right behind the requests)
        if (FBX) {
          if (st + 1 < T) asm volatile("s_waitcnt vmcnt(6)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]) :: "memory");
          else asm volatile("s_waitcnt vmcnt(4)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]) :: "memory");
        } else if (st + 1 < T) asm volatile("s_waitcnt vmcnt(2)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(vg[0]), "+v"(vg[1]), "+v"(vg[2]), "+v"(vg[3]) :: "memory");
      }
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) ok = ok && (vg[i][0] & 0xffffu) == want && (vg[i][2] & 0xffffu) == want;
        if (ok) break;
        if (++spins > SPIN_LIMIT) { fail = true; break; }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!((vg[i][0] & 0xffffu) == want && (vg[i][2] & 0xffffu) == want))
            vg[i] = __builtin_amdgcn_raw_buffer_load_b128(prs, (((slot * G + srcw[i]) * SQ + s2) * Hp + 64 * g + 4 * uq) * 4, 0, SC1);
      }
      if (fail) s_fail = 1;
      // fixed order: sources 0 .. G - 1 ascending, the own partial in its place (value selects between static elements)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 own = *reinterpret_cast<const f32x4*>(pown + (p * SQ + s2) * OWNP + 4 * uq);
#pragma unroll
      for (int gs = 0; gs < 5; ++gs) {
        const u32x4 wlo = vg[gs < 4 ? gs : 3], whi = vg[gs > 0 ? gs - 1 : 0];
        const u32x4 w = gs < g ? wlo : whi;
        const f32x4 pv = {granule_a(w[0], w[1]), granule_b(w[1]), granule_a(w[2], w[3]), granule_b(w[3])};
        acc += gs == g ? own : pv;
      }
      *reinterpret_cast<f32x4*>(&dsum[s2 * UPW + 4 * uq]) = acc;
    };
    auto publish = [&](int p, int64_t st) __attribute__((always_inline)) {
      const auto prs = payload_srd(p);
      const unsigned tag = mk_tag(tagbase, st + 1);
      const int slot = (int)(st & 1);
      EX_LANE();
#pragma unroll
      for (int b = ONCHIP16_BWD_DIRECT ? 4 : 0; b < 5; ++b) {           // unit block b = the 64 units of workgroup b
        f32x4 v = *reinterpret_cast<const f32x4*>(psum + s2 * PSW + 64 * b - (ONCHIP16_BWD_DIRECT ? 256 : 0) + 4 * uq);
        if (b == 4) v += *reinterpret_cast<const f32x4*>(psum2 + s2 * OWNP + 4 * uq);
        if constexpr (ONCHIP16_BWD_DIRECT && ONCHIP16_BWD_OWN_SHARED) {      // the shared tiles ARE the own block
          *reinterpret_cast<f32x4*>(pown + (p * SQ + s2) * OWNP + 4 * uq) = v;
          continue;
        }
        const u32x2 ga = pack_granule(tag, v[0], v[1]), gb = pack_granule(tag, v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{ga[0], ga[1], gb[0], gb[1]}, prs,
                                               (int)(b != g ? (unsigned)((((slot * G + g) * SQ + s2) * Hp + 64 * b + 4 * uq) * 4) : OOR), 0, SC0);
        if (b == g) *reinterpret_cast<f32x4*>(pown + (p * SQ + s2) * OWNP + 4 * uq) = v;
      }
    };

    // ---- one phase: ring slot S = phase index (static), group P (static), step st
    auto phase = [&](auto io_tag, auto slot_tag, auto grp_tag, int64_t st, int64_t base) __attribute__((always_inline)) {
      constexpr bool IO = decltype(io_tag)::value;
      constexpr int S = decltype(slot_tag)::value, P = decltype(grp_tag)::value;
      if (st >= T) return;
      const bool has_prev = st + 1 < T;
      constexpr int TR = IO ? 1 : 0;
      stamp(TR, P == 0 ? st : -1, 0);
      if constexpr (!IO) {
        if (st > 0) gather_finish(S, P, st);
      } else {
        // this phase's tiles have landed (behind them: four stores + six copies, + the two direct publish stores)
        if (ONCHIP16_BWD_DIRECT && ((NGA == 1 && ONCHIP16_BWD_EARLY_DMA_G1) || (NGA >= 2 && ONCHIP16_BWD_EARLY_DMA_G2))) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (ONCHIP16_BWD_DIRECT && FBX) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // (no stores of their own: 2 + 6)
        else if (ONCHIP16_BWD_DIRECT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      }
      stamp(TR, P == 0 ? st : -1, 1);
      lds_barrier();
      stamp(TR, P == 0 ? st : -1, 2);
      if (s_fail) return;
      constexpr int GAT = NGA == 1 ? 0 : ONCHIP16_BWD_GATHER;
      constexpr int I1 = (S + 1) & 3, I2 = (S + 2) & 3;
      const int64_t b1 = I1 == 0 ? base + 4 / NGA : base, b2 = I2 < 2 ? base + 4 / NGA : base;
      constexpr int P1 = I1 % NGA, P2 = I2 % NGA;
      const int64_t st1 = b1 + I1 / NGA, st2 = b2 + I2 / NGA;
      if constexpr (!IO && GAT == 1) {
        if (st1 > 0 && st1 < T) gather_issue(P1, st1);
      }
      // cell backward for this lane's two cells; d(gates) replaces the activations in the tile
      {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int s = tv >> 5, up = tv & 31;
        const bool valid = seq0_of(P) + s < N && 64 * g + 2 * up < H;
        float2 dh = *reinterpret_cast<const float2*>(&ringd[S][s * UPW + 2 * up]);
        if (st > 0) {
          const float2 ds = *reinterpret_cast<const float2*>(&dsum[s * UPW + 2 * up]);
          dh.x += ds.x; dh.y += ds.y;
        }
        const float2 cpv = has_prev ? *reinterpret_cast<const float2*>(&ringc[S][s * UPW + 2 * up]) : make_float2(0.f, 0.f);
        const float dhq[2] = {dh.x, dh.y}, cpq[2] = {cpv.x, cpv.y};
        f32x4 d[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int u = 2 * up + q;
          f32x4* xp = &ringg[S][(u >> 4) * 256 + s * 16 + ((u & 15) ^ s)];
          d[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (valid) {
            const f32x4 gq = *xp;
            const float tc = fast_tanh(ctc[P][q]);
            const float d_o = dhq[q] * tc;
            const float dc = dhq[q] * gq[3] * (1.f - tc * tc) + dcr[P][q];
            dcr[P][q] = dc * gq[1];
            d[q][0] = dc * gq[2] * gq[0] * (1.f - gq[0]);
            d[q][1] = dc * cpq[q] * gq[1] * (1.f - gq[1]);
            d[q][2] = dc * gq[0] * (1.f - gq[2] * gq[2]);
            d[q][3] = d_o * gq[3] * (1.f - gq[3]);
            *xp = d[q];
          }
          ctc[P][q] = cpq[q];
        }
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split2(d[0][0], d[0][1], h0, l0);
        split2(d[0][2], d[0][3], h1, l1);
        split2(d[1][0], d[1][1], h2, l2);
        split2(d[1][2], d[1][3], h3, l3);
        const int o = s * DP2 + 16 * up;                  // 8 gate columns = 16 bytes of bf16
        *reinterpret_cast<u32x4*>(dg_hi + o) = u32x4{h0, h1, h2, h3};
        *reinterpret_cast<u32x4*>(dg_lo + o) = u32x4{l0, l1, l2, l3};
      }
      stamp(TR, P == 0 ? st : -1, 3);
      lds_barrier();
      stamp(TR, P == 0 ? st : -1, 4);
      if constexpr (!IO && (GAT >= 2)) {
        if (st1 > 0 && st1 < T && (GAT == 2 || !has_prev)) gather_issue(P1, st1);
      }
      if constexpr (IO && ((NGA == 1 && ONCHIP16_BWD_EARLY_DMA_G1) || (NGA >= 2 && ONCHIP16_BWD_EARLY_DMA_G2))) io_dma(I2, P2, st2 < T ? st2 : T - 1);      // (one group: as in the forward)
      if constexpr ((FBX ? !IO : IO) && NGA >= 2 && ONCHIP16_BWD_DEFER_FLUSH) {
        // the d(gates) of the PREVIOUS phase leave here, beside the MFMAs, not between two phases where the publish waits
        // for issue slots (its ring slot is refilled a phase later)
        constexpr int SP = (S + 3) & 3, PP = SP % NGA;
        const int64_t stp = (SP == 3 ? base - 4 / NGA : base) + SP / NGA;
        const bool livep = stp >= 0 && stp < T;
        io_flush(SP, PP, livep ? stp : 0, livep);
        // (requesting the tiles of phase + 2 here as well measured -5 % with ONE group at 768 / 3 072 sequences, +6 % at the 32
        // sequences one group is used for, and +6 % with two groups: profiles/r3_onchip16_deferred_flush.jsonl -- not kept)
      }
      stamp(TR, P == 0 ? st : -1, 5);
      if (has_prev) {
        // partial dh_(t-1): two own tiles over 8 k-steps + 4 k-steps of a shared tile (three accumulator chains)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, accs = acc0;
        int lvm = lane;
        asm volatile("" : "+v"(lvm));
        const int foff = (lvm & 15) * DP2 + (lvm >> 4) * 16;       // B fragment offset (row = sequence, 16-byte k chunk)
        // fragments one k-step ahead, pinned (all sixteen reads hoisted would cost 64 registers: W fragments spilled)
        bf16x8 bh = *reinterpret_cast<const bf16x8*>(dg_hi + foff), bl = *reinterpret_cast<const bf16x8*>(dg_lo + foff);
        // (one group, own block = shared tiles: the shared tile's products follow the peers' tiles, below)
        constexpr bool OWN_LAST = NGA == 1 && ONCHIP16_BWD_DIRECT && ONCHIP16_BWD_OWN_SHARED;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          bf16x8 nh = bh, nl = bl;
          if (ks + 1 < 8) {
            nh = *reinterpret_cast<const bf16x8*>(dg_hi + foff + (ks + 1) * 64);
            nl = *reinterpret_cast<const bf16x8*>(dg_lo + foff + (ks + 1) * 64);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc0 = MFMA16_BF16(as_bf16x8(wl[ks]), bh, acc0);
          acc1 = MFMA16_BF16(as_bf16x8(wl[8 + ks]), bh, acc1);
          // (wave-uniform: this k-step also belongs to the wave's half of the shared tile)
          if (!OWN_LAST && (ks >> 2) == (wave & 1)) accs = MFMA16_BF16(as_bf16x8(wl[16 + (ks & 3)]), bh, accs);
          acc0 = MFMA16_BF16(as_bf16x8(wh[ks]), bl, acc0);
          acc1 = MFMA16_BF16(as_bf16x8(wh[8 + ks]), bl, acc1);
          if (!OWN_LAST && (ks >> 2) == (wave & 1)) accs = MFMA16_BF16(as_bf16x8(wh[16 + (ks & 3)]), bl, accs);
          acc0 = MFMA16_BF16(as_bf16x8(wh[ks]), bh, acc0);
          acc1 = MFMA16_BF16(as_bf16x8(wh[8 + ks]), bh, acc1);
          if (!OWN_LAST && (ks >> 2) == (wave & 1)) accs = MFMA16_BF16(as_bf16x8(wh[16 + (ks & 3)]), bh, accs);
          __builtin_amdgcn_sched_barrier(0);
          bh = nh; bl = nl;
          if constexpr (!IO && GAT >= 3) {
            if (ks == GAT - 1 && st1 > 0 && st1 < T) gather_issue(P1, st1);
          }
        }
        stamp(TR, P == 0 ? st : -1, 6);
        const int j = lvm & 15, r4 = 4 * (lvm >> 4);
        if constexpr (ONCHIP16_BWD_DIRECT) {
          // the wave's own two tiles (units 32 wave .. + 31 of block wave / 2) leave from the accumulators: a lane holds four
          // consecutive units of one sequence = one 16-byte pair of granules; no LDS round trip, no barrier in front
          const int bw = wave >> 1;
          const auto prs = payload_srd(P);
          const unsigned tag = mk_tag(tagbase, st + 1);
          const u32x2 a0 = pack_granule(tag, acc0[0], acc0[1]), a1 = pack_granule(tag, acc0[2], acc0[3]);
          const u32x2 c0 = pack_granule(tag, acc1[0], acc1[1]), c1 = pack_granule(tag, acc1[2], acc1[3]);
          if constexpr (ONCHIP16_BWD_OWN_SHARED) {
            // tile block bw holds the units of workgroup (bw + g + 1) mod G: always a peer
            int bd = bw + g + 1;
            bd = bd >= G ? bd - G : bd;
            const unsigned off = (unsigned)(((((int)(st & 1) * G + g) * SQ + j) * Hp + 64 * bd + 32 * (wave & 1) + r4) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0[0], a0[1], a1[0], a1[1]}, prs, (int)off, 0, SC0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{c0[0], c0[1], c1[0], c1[1]}, prs, (int)(off + 64u), 0, SC0);
          } else {
            const unsigned off = (unsigned)(((((int)(st & 1) * G + g) * SQ + j) * Hp + 32 * wave + r4) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{a0[0], a0[1], a1[0], a1[1]}, prs, (int)(bw != g ? off : OOR), 0, SC0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{c0[0], c0[1], c1[0], c1[1]}, prs, (int)(bw != g ? off + 64u : OOR), 0, SC0);
            if (bw == g) {
              *reinterpret_cast<f32x4*>(pown + (P * SQ + j) * OWNP + 32 * (wave & 1) + r4) = acc0;
              *reinterpret_cast<f32x4*>(pown + (P * SQ + j) * OWNP + 32 * (wave & 1) + 16 + r4) = acc1;
            }
          }
          if constexpr (OWN_LAST) {
            // the own block's partial sums: this wave's four k-steps of shared tile 16 + (wave >> 1), behind the peers' stores
            const int kb = __builtin_amdgcn_readfirstlane(4 * (wave & 1));
            __builtin_amdgcn_sched_barrier(0);      // (the stores above are ISSUED first)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bf16x8 sh_ = *reinterpret_cast<const bf16x8*>(dg_hi + foff + (kb + q) * 64);
              const bf16x8 sl_ = *reinterpret_cast<const bf16x8*>(dg_lo + foff + (kb + q) * 64);
              accs = MFMA16_BF16(as_bf16x8(wl[16 + q]), sh_, accs);
              accs = MFMA16_BF16(as_bf16x8(wh[16 + q]), sl_, accs);
              accs = MFMA16_BF16(as_bf16x8(wh[16 + q]), sh_, accs);
            }
          }
        } else {
          *reinterpret_cast<f32x4*>(psum + j * PP2 + 32 * wave + r4) = acc0;
          *reinterpret_cast<f32x4*>(psum + j * PP2 + 32 * wave + 16 + r4) = acc1;
        }
        float* sh = (wave & 1) ? psum2 + j * OWNP : psum + j * PSW + PSO;
        *reinterpret_cast<f32x4*>(sh + 16 * (wave >> 1) + r4) = accs;
      } else if constexpr (IO && ONCHIP16_BWD_DIRECT) {
        // (last step: the io waves' vector-memory count per phase must not depend on the step)
        const auto prs = payload_srd(P);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, prs, (int)OOR, 0, SC0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, prs, (int)OOR, 0, SC0);
      }
      stamp(TR, P == 0 ? st : -1, 7);
      lds_barrier();
      stamp(TR, P == 0 ? st : -1, 8);
      if constexpr (!IO) {
        if (has_prev) publish(P, st);
        if constexpr (GAT == 0) {
          // (no pause in front of the request as in the forward: the direct publish is earlier here, a pause costs 1-9 %)
          if (st1 > 0 && st1 < T) gather_issue(P1, st1);
        }
      } else {
        if constexpr (!(NGA >= 2 && ONCHIP16_BWD_DEFER_FLUSH)) io_flush(S, P, st, true);
        if constexpr (!((NGA == 1 && ONCHIP16_BWD_EARLY_DMA_G1) || (NGA >= 2 && ONCHIP16_BWD_EARLY_DMA_G2))) io_dma(I2, P2, st2 < T ? st2 : T - 1);
      }
      stamp(TR, P == 0 ? st : -1, 9);
    };
    auto run = [&](auto io_tag) __attribute__((always_inline)) {
      constexpr bool IO = decltype(io_tag)::value;
      if constexpr (IO) {      // prologue: the tiles of phases 0 and 1
        io_dma(0, 0 % NGA, 0);
        io_dma(1, 1 % NGA, (1 / NGA) < T ? 1 / NGA : T - 1);
      }
#pragma unroll
      for (int f = 0; f < 20; ++f) asm volatile("" :: "v"(wh[f]), "v"(wl[f]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int64_t base = 0; base < T; base += 4 / NGA) {
        phase(io_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 0 % NGA>{}, base + 0 / NGA, base);
        phase(io_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NGA>{}, base + 1 / NGA, base);
        phase(io_tag, std::integral_constant<int, 2>{}, std::integral_constant<int, 2 % NGA>{}, base + 2 / NGA, base);
        phase(io_tag, std::integral_constant<int, 3>{}, std::integral_constant<int, 3 % NGA>{}, base + 3 / NGA, base);
        if (s_fail) break;
      }
      if constexpr (FBX ? !IO : IO) {
        // (deferred flush: the last phase's d(gates) still sit in their ring slot)
        if (NGA >= 2 && ONCHIP16_BWD_DEFER_FLUSH && !s_fail) io_flush((int)((T * NGA - 1) & 3), NGA - 1, T - 1, true);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (io_wave) run(std::true_type{}); else run(std::false_type{});
    if (s_fail) {
      if (tid == 0) atomicExch(err, 4);
      return;
    }
    lds_barrier();
  }
}

}  // namespace

extern "C" int tssep_lstm_onchip_supported(int H) { return H > 0 && H <= KP ? 1 : 0; }

extern "C" int64_t tssep_lstm_onchip_pack_floats(int H, int which) {
  const int G = (H + UPW - 1) / UPW;
  return which == 0 ? (int64_t)2 * G * 8 * KS * 2 * 64 * 4 : (int64_t)2 * G * 10 * 16 * 2 * 64 * 4;
}

extern "C" int tssep_lstm_pack_onchip(const float* w_hh_f, const float* w_hh_r, int H, float* wf,
                                      float* wb, void* stream) {
  if (!w_hh_f || !w_hh_r || !wf || !wb) return TSSEP_E_NULL;
  if (!tssep_lstm_onchip_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(wf) || !aligned16(wb)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  const int64_t total = (tssep_lstm_onchip_pack_floats(H, 0) + tssep_lstm_onchip_pack_floats(H, 1)) / 4;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_onchip_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     w_hh_f, w_hh_r, H, G, (u32x4*)wf, (u32x4*)wb);
  return tssep_launch_status();
}

// bytes: header 64 | flags (items*2*G*4, rounded to 64) | payload
static void xbuf_layout(int64_t N, int G, int pw, int64_t* items, int64_t* payload_bytes) {
  *items = 2 * ((N + SEQS - 1) / SEQS);
  *payload_bytes = *items * 2 * G * SEQS * pw * 4;      // compact granules: 4 bytes per value
}

extern "C" int64_t tssep_lstm_onchip_xbuf_bytes(int64_t N, int H, int backward) {
  const int G = (H + UPW - 1) / UPW;
  int64_t items, pb;
  xbuf_layout(N, G, backward ? G * UPW : UPW, &items, &pb);
  return HDR_BYTES + pb;
}

// (granule tags carry a per-launch epoch kept in device memory: common.h; tag16_base above)

// Longest sequence a W-stationary launch takes: a lane addresses its sequence's rows of gates / cell / hout with a 32-bit
// byte offset from the work item's base -- (seqs - 1) T 2H 16 bytes for the gate tensor must stay below 2^31.  (The step
// field of the exchange tags wraps: no limit from there any more.)  H = 300: 14 913 frames (238 s) for the 16-sequence
// kernels, 7 215 for the 32-sequence ones; tssep/train/rnnp.py:111-173 has no limit -- beyond, the streaming kernels run.
extern "C" int64_t tssep_lstm_onchip_max_steps(int H, int seqs_per_item) {
  if (H <= 0 || seqs_per_item < 2) return 0;
  return (((int64_t)1 << 31) - 8192) / ((int64_t)(seqs_per_item - 1) * 2 * H * 16);
}

// grid: cross-XCD mode -> exactly the clusters wanted; XCD-local mode -> whole clusters per XCD
// (workgroup b is observed on XCD b % 8; a cluster needs G workgroups of ONE XCD) plus one spare
// workgroup per XCD for uneven dispatch.  Leftover workgroups exit (join_cluster).
static unsigned onchip_grid(int64_t items, int G, int max_wgs, bool xcd, int* nclusters) {
  if (!xcd) {
    const int cap = max_wgs / G;
    *nclusters = (int)(items < cap ? items : cap);
    return (unsigned)(*nclusters * G);
  }
  const int per_xcd_cap = (max_wgs / 8) / G;                       // 32 CUs / 5 = 6
  int per_xcd = (int)((items + 7) / 8);
  if (per_xcd > per_xcd_cap) per_xcd = per_xcd_cap;
  *nclusters = 8 * per_xcd;
  int wgs = 8 * (per_xcd * G + 1);
  return (unsigned)(wgs < max_wgs ? wgs : max_wgs);
}

extern "C" int tssep_blstm_onchip_fwd(float* gates, float* cell, float* hout, int64_t ldo,
                                      int64_t dstride, const float* wf, void* xbuf, int* err,
                                      int64_t N, int64_t T, int H, int max_wgs, int layout,
                                      void* stream) {
  if (!gates || !cell || !hout || !wf || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!tssep_lstm_onchip_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  if (max_wgs < G) return TSSEP_E_SHAPE;
  int64_t items, pb;
  xbuf_layout(N, G, UPW, &items, &pb);
  if (items >= 0xffff || T > tssep_lstm_onchip_max_steps(H, SEQS)) return TSSEP_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (tssep_xbuf_reset(xbuf, (size_t)(HDR_BYTES + pb), err, s) != TSSEP_OK) return TSSEP_E_LAUNCH;
  // XCD-local clusters (48 on MI355X) unless asked otherwise -- or unless the cross-XCD packing
  // (51 clusters) saves a whole resident round
  const int xcd_cap = 8 * ((max_wgs / 8) / G), flat_cap = max_wgs / G;
  const bool xcd = !(layout & 8) && xcd_cap > 0 && !(items > xcd_cap && items <= flat_cap);
  int nc;
  const unsigned grid = onchip_grid(items, G, max_wgs, xcd, &nc);
  char* base = (char*)xbuf;
  // bit 4 of the kernel's layout word: non-temporal activation stream (layout bit 32 forces it off)
  const int klayout = (layout & 1) | ((N >= 160 && !(layout & 32)) ? 16 : 0);
  if (xcd)
    hipLaunchKernelGGL(blstm_onchip_fwd_kernel<true>, dim3(grid), dim3(512), 0, s, gates, cell, hout,
                       ldo, dstride, (const u32x4*)wf, (unsigned*)base, (float*)(base + HDR_BYTES), err,
                       N, T, H, G, nc, klayout);
  else
    hipLaunchKernelGGL(blstm_onchip_fwd_kernel<false>, dim3(grid), dim3(512), 0, s, gates, cell, hout,
                       ldo, dstride, (const u32x4*)wf, (unsigned*)base, (float*)(base + HDR_BYTES), err,
                       N, T, H, G, nc, klayout);
  return tssep_launch_status();
}

extern "C" int tssep_blstm_onchip_bwd(float* gates, const float* cell, const float* dhout,
                                      int64_t ldo, int64_t dstride, const float* wb, void* xbuf,
                                      int* err, int64_t N, int64_t T, int H, int max_wgs, int layout,
                                      void* stream) {
  if (!gates || !cell || !dhout || !wb || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!tssep_lstm_onchip_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  if (max_wgs < G) return TSSEP_E_SHAPE;
  int64_t items, pb;
  xbuf_layout(N, G, G * UPW, &items, &pb);
  if (items >= 0xffff || T > tssep_lstm_onchip_max_steps(H, SEQS)) return TSSEP_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (tssep_xbuf_reset(xbuf, (size_t)(HDR_BYTES + pb), err, s) != TSSEP_OK) return TSSEP_E_LAUNCH;
  // XCD-local clusters (48 on MI355X) unless asked otherwise -- or unless the cross-XCD packing
  // (51 clusters) saves a whole resident round
  const int xcd_cap = 8 * ((max_wgs / 8) / G), flat_cap = max_wgs / G;
  const bool xcd = !(layout & 8) && xcd_cap > 0 && !(items > xcd_cap && items <= flat_cap);
  int nc;
  const unsigned grid = onchip_grid(items, G, max_wgs, xcd, &nc);
  char* base = (char*)xbuf;
  if (xcd)
    hipLaunchKernelGGL(blstm_onchip_bwd_kernel<true>, dim3(grid), dim3(512), 0, s, gates, cell, dhout,
                       ldo, dstride, (const u32x4*)wb, (unsigned*)base, (float*)(base + HDR_BYTES), err, N,
                       T, H, G, nc, layout & 1);
  else
    hipLaunchKernelGGL(blstm_onchip_bwd_kernel<false>, dim3(grid), dim3(512), 0, s, gates, cell, dhout,
                       ldo, dstride, (const u32x4*)wb, (unsigned*)base, (float*)(base + HDR_BYTES), err, N,
                       T, H, G, nc, layout & 1);
  return tssep_launch_status();
}

// ---- interleaved forward (blstm_onchip16_fwd_kernel): own weight pack, own exchange layout (16-sequence items)
// `waves` = waves per workgroup: 8 (five workgroups of 64 units per cluster, one per CU) or 4 (ten of 32 units, two per CU)
// (ABI 4: the four-wave variant -- built, parity-green, 1.4-1.8x slower at every size, profiles/r5_onchip16_w4_rejected.jsonl
// -- exists in the EXPERIMENT build only (-DTSSEP_GEMM_EXP, `make exp`); the product exports the eight-wave entry
// points and instantiates no four-wave kernel)
#ifdef TSSEP_GEMM_EXP
#define W16_API extern "C"
static inline bool waves_ok(int waves) { return waves == 8 || waves == 4; }
#else
#define W16_API static
static inline bool waves_ok(int waves) { return waves == 8; }
#endif
static inline int g_of(int H, int waves) { return (H + 8 * waves - 1) / (8 * waves); }
W16_API int64_t tssep_lstm_onchip16w_pack_floats(int H, int waves) {
  if (!waves_ok(waves)) return 0;
  return (int64_t)2 * g_of(H, waves) * waves * 2 * KS2 * 2 * 64 * 4;
}
extern "C" int64_t tssep_lstm_onchip16_pack_floats(int H) { return tssep_lstm_onchip16w_pack_floats(H, 8); }
W16_API int tssep_lstm_pack_onchip16w(const float* w_hh_f, const float* w_hh_r, int H, int waves, float* wf, void* stream) {
  if (!w_hh_f || !w_hh_r || !wf) return TSSEP_E_NULL;
  if (H <= 0 || H > KP2 || !waves_ok(waves)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(wf)) return TSSEP_E_ALIGN;
  const int G = g_of(H, waves);
  const int64_t total = tssep_lstm_onchip16w_pack_floats(H, waves) / 4;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_onchip16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_hh_f, w_hh_r, H, G,
                     waves, (u32x4*)wf);
  return tssep_launch_status();
}
extern "C" int tssep_lstm_pack_onchip16(const float* w_hh_f, const float* w_hh_r, int H, float* wf, void* stream) {
  return tssep_lstm_pack_onchip16w(w_hh_f, w_hh_r, H, 8, wf, stream);
}
W16_API int64_t tssep_lstm_onchip16w_xbuf_bytes(int64_t N, int H, int waves) {
  if (!waves_ok(waves)) return 0;
  return HDR_BYTES + 2 * ((N + SQ - 1) / SQ) * 2 * g_of(H, waves) * SQ * (8 * waves) * 4;
}
extern "C" int64_t tssep_lstm_onchip16_xbuf_bytes(int64_t N, int H) { return tssep_lstm_onchip16w_xbuf_bytes(N, H, 8); }
// groups per cluster the launcher would use for N sequences (0: shape not supported -> use tssep_blstm_onchip_fwd):
// 2 where that divides the number of 16-sequence groups and still gives every cluster a bundle, else 1 (four groups
// -- `groups` = 4 -- stay available: slower than two since the two-group kernel requests its operands early).
// Four-wave workgroups: ONE group per workgroup -- the second chain of a CU is its second workgroup.
W16_API int tssep_blstm_onchip16w_groups(int64_t N, int H, int max_wgs, int waves) {
  if (!waves_ok(waves)) return 0;
  const int G = g_of(H, waves), slots = max_wgs * (8 / waves);
  if (N <= 0 || H <= 0 || H > KP2 || (H & 3) || G > 40 / waves || slots < 8 * G) return 0;
  if (waves == 4) return 1;
  const int64_t ng16 = (N + SQ - 1) / SQ;
  const int64_t ncl = 8 * ((slots / 8) / G);
  for (int nga = 2; nga >= 1; nga >>= 1)
    if (ng16 % nga == 0 && (2 * ng16 / nga >= ncl || nga == 1)) return nga;
  return 1;
}
extern "C" int tssep_blstm_onchip16_groups(int64_t N, int H, int max_wgs) { return tssep_blstm_onchip16w_groups(N, H, max_wgs, 8); }
W16_API int tssep_blstm_onchip16w_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                                         const float* wf, void* xbuf, int* err, int64_t N, int64_t T, int H,
                                         int max_wgs, int layout, int groups, int waves, void* stream) {
  if (!gates || !cell || !hout || !wf || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!waves_ok(waves)) return TSSEP_E_UNSUPPORTED;
  const int G = g_of(H, waves), slots = max_wgs * (8 / waves);      // (four-wave workgroups: two per CU)
  const int nga = groups > 0 ? groups : tssep_blstm_onchip16w_groups(N, H, max_wgs, waves);
  const int64_t ng16 = (N + SQ - 1) / SQ;
  if (nga != 1 && nga != 2 && !(nga == 4 && waves == 8)) return TSSEP_E_UNSUPPORTED;
  if (H > KP2 || (H & 3) || (ldo & 3) || (dstride & 3) || ng16 % nga || G > 40 / waves || slots < 8 * G) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf) || !aligned16(cell) || !aligned16(hout)) return TSSEP_E_ALIGN;
  if (2 * ng16 >= 0xffff || T > tssep_lstm_onchip_max_steps(H, SQ)) return TSSEP_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (tssep_xbuf_reset(xbuf, (size_t)tssep_lstm_onchip16w_xbuf_bytes(N, H, waves), err, s) != TSSEP_OK) return TSSEP_E_LAUNCH;
  int nc;
  const unsigned grid = onchip_grid(2 * ng16 / nga, G, slots, true, &nc);
  char* base = (char*)xbuf;
  const int klayout = layout & 1;
  const bool nt = N >= 160 && !(layout & 32);          // non-temporal activation stream (as in the 32-sequence kernel)
#define L16(NGA_, NW_) if (nt) L16B(NGA_, true, NW_); else L16B(NGA_, false, NW_)
#define L16B(NGA_, NT_, NW_) hipLaunchKernelGGL((blstm_onchip16_fwd_kernel<NGA_, NT_, NW_>), dim3(grid), dim3(64 * NW_), 0, s, gates, cell, hout, \
                    ldo, dstride, (const u32x4*)wf, (unsigned*)base, (float*)(base + HDR_BYTES), err, N, T, H, G, nc, klayout)
#ifdef TSSEP_GEMM_EXP
  if (waves == 4) { if (nga == 2) { L16(2, 4); } else { L16(1, 4); } } else
#endif
  if (nga == 4) { L16(4, 8); } else if (nga == 2) { L16(2, 8); } else { L16(1, 8); }
#undef L16
#undef L16B
  return tssep_launch_status();
}
extern "C" int tssep_blstm_onchip16_fwd(float* gates, float* cell, float* hout, int64_t ldo, int64_t dstride,
                                        const float* wf, void* xbuf, int* err, int64_t N, int64_t T, int H,
                                        int max_wgs, int layout, int groups, void* stream) {
  return tssep_blstm_onchip16w_fwd(gates, cell, hout, ldo, dstride, wf, xbuf, err, N, T, H, max_wgs, layout, groups, 8, stream);
}

// ---- interleaved backward (blstm_onchip16_bwd_kernel)
extern "C" int64_t tssep_lstm_onchip16_bwd_pack_floats(int H) {
  const int G = (H + UPW - 1) / UPW;
  return (int64_t)2 * G * 8 * 20 * 2 * 64 * 4;
}
extern "C" int tssep_lstm_pack_onchip16_bwd(const float* w_hh_f, const float* w_hh_r, int H, float* wb, void* stream) {
  if (!w_hh_f || !w_hh_r || !wb) return TSSEP_E_NULL;
  if (H <= 0 || H > KP2) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(wb)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  const int64_t total = tssep_lstm_onchip16_bwd_pack_floats(H) / 4;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_onchip16_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w_hh_f, w_hh_r, H,
                     G, (u32x4*)wb);
  return tssep_launch_status();
}
extern "C" int64_t tssep_lstm_onchip16_bwd_xbuf_bytes(int64_t N, int H) {
  const int G = (H + UPW - 1) / UPW;
  return HDR_BYTES + 2 * ((N + SQ - 1) / SQ) * 2 * G * SQ * (int64_t)(G * UPW) * 4;
}
extern "C" int tssep_blstm_onchip16_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo, int64_t dstride,
                                        const float* wb, void* xbuf, int* err, int64_t N, int64_t T, int H, int max_wgs,
                                        int layout, int groups, void* stream) {
  if (!gates || !cell || !dhout || !wb || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  const int G = (H + UPW - 1) / UPW;
  const int nga = groups > 0 ? groups : tssep_blstm_onchip16_groups(N, H, max_wgs);
  const int64_t ng16 = (N + SQ - 1) / SQ;
  if (nga != 1 && nga != 2 && nga != 4) return TSSEP_E_UNSUPPORTED;
  // (five workgroups per cluster: the exchange arm gathers exactly four peers)
  if (H > KP2 || (H & 3) || G != 5 || (ldo & 3) || (dstride & 3) || ng16 % nga || max_wgs < 8 * G) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf) || !aligned16(cell) || !aligned16(dhout)) return TSSEP_E_ALIGN;
  if (2 * ng16 >= 0xffff || T > tssep_lstm_onchip_max_steps(H, SQ)) return TSSEP_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (tssep_xbuf_reset(xbuf, (size_t)tssep_lstm_onchip16_bwd_xbuf_bytes(N, H), err, s) != TSSEP_OK) return TSSEP_E_LAUNCH;
  int nc;
  const unsigned grid = onchip_grid(2 * ng16 / nga, G, max_wgs, true, &nc);
  char* base = (char*)xbuf;
  const bool nt = N >= 160 && !(layout & 32);
#define LB16(NGA_) if (nt) LB16B(NGA_, true); else LB16B(NGA_, false)
#define LB16B(NGA_, NT_) hipLaunchKernelGGL((blstm_onchip16_bwd_kernel<NGA_, NT_>), dim3(grid), dim3(512), 0, s, gates, cell, dhout, ldo, \
                     dstride, (const u32x4*)wb, (unsigned*)base, (float*)(base + HDR_BYTES), err, N, T, H, G, nc, layout & 1)
  if (nga == 4) { LB16(4); } else if (nga == 2) { LB16(2); } else { LB16(1); }
#undef LB16
#undef LB16B
  return tssep_launch_status();
}
