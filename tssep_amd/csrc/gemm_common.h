// Shared pieces of the two GEMM kernels (exact fp32 and split-bf16): output mapping + epilogue.
#pragma once
#include "common.h"
#include "gemm_dispatch.h"

namespace gemm_detail {

// CU count of the CURRENT device for the persistent kernels' grids, remembered per device ordinal (a process with mixed
// devices gets each device's own count -- ADVICE r4; the value of a device never changes, so the memo is not state a
// caller could observe)
static inline int current_device_cus() {
  static int memo[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&memo[dev], __ATOMIC_RELAXED);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&memo[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

// Tanh of the fused store (act = 1), the same in every GEMM kernel: 1 - 2 / (1 + e^(2x)) on v_exp_f32 / v_rcp_f32 (the
// recurrence kernels' fast_tanh): absolute error <= 3e-7, saturates correctly for large |x|, four instructions instead of
// the ~40 of OCML's tanhf -- which cost a third of a K = 600 tile's life in the persistent kernels' store (240 calls per lane).
__device__ __forceinline__ float gemm_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681472f * x));
}


constexpr int BM = 128, BN = 128, NTHREADS = 256;

// ---- XCD-aware workgroup -> tile map --------------------------------------------------------
// MI355X: 8 XCDs with private 4 MB L2s; workgroup b is observed to run on XCD b % 8 (a speed
// hint only -- any placement is correct).  With fp32 operands a 128x128 tile has an arithmetic
// intensity of only 32 FLOP/byte, so the GEMMs are L2/HBM-bound unless operand panels are reused
// out of the SAME L2:
//  * splitk == 1 (M large): XCD x owns the m-tiles {x, x+8, ...}; inside an XCD the order is
//    n-group (NG n-tiles whose weight slice stays in L2) > m-tile > n-tile, so an A panel is
//    fetched from HBM once per n-group and the weight slice once per XCD.
//  * splitk  > 1 (wgrad, K huge): the split index is the fastest-varying part of the id, so all
//    tiles of one K slab run on one XCD at the same time and share the slab through its L2.
constexpr int NXCD = 8, NGROUP_MAX = 8;
struct TileMap {
  int MT, NT, S, MTx, NGc, NG;      // NG = n-tiles per group (<= 8), chosen to leave few idle ids
};
__host__ __device__ inline TileMap make_tile_map(int64_t mtiles, int64_t ntiles, int splitk) {
  TileMap t;
  t.MT = (int)mtiles; t.NT = (int)ntiles; t.S = splitk;
  t.MTx = (int)((mtiles + NXCD - 1) / NXCD);
  t.NGc = (int)((ntiles + NGROUP_MAX - 1) / NGROUP_MAX);
  t.NG = (int)((ntiles + t.NGc - 1) / t.NGc);
  return t;
}
__host__ __device__ inline int64_t tile_map_blocks(const TileMap& t) {
  if (t.S > 1) return (int64_t)t.S * t.MT * t.NT;
  return (int64_t)NXCD * t.MTx * t.NGc * t.NG;
}
__device__ __forceinline__ bool tile_map_decode(const TileMap& t, int64_t bid, int& mt, int& nt, int& z) {
  if (t.S > 1) {
    z = (int)(bid % t.S);
    const int64_t tile = bid / t.S;
    mt = (int)(tile / t.NT);
    nt = (int)(tile % t.NT);
    return true;
  }
  z = 0;
  const int xcd = (int)(bid % NXCD);
  const int64_t l = bid / NXCD;
  const int per_group = t.MTx * t.NG;
  const int ng = (int)(l / per_group);
  const int r = (int)(l % per_group);
  mt = (r / t.NG) * NXCD + xcd;
  nt = ng * t.NG + (r % t.NG);
  return mt < t.MT && nt < t.NT;
}

struct StoreMap {
  int64_t ldc;
  int32_t remap;
  int64_t T, K, sb, sk, st, cm, co;
  const int32_t* perm; int64_t perm_ld;
  const float* aux; int64_t ldaux;      // act == 2: C = acc * (1 - aux[m, n]^2)
};

inline StoreMap make_store_map(const tssep_gemm_args* g) {
  StoreMap sm;
  sm.ldc = g->ldc; sm.remap = g->c_remap;
  sm.T = g->c_T > 0 ? g->c_T : 1; sm.K = g->c_K > 0 ? g->c_K : 1;
  sm.sb = g->c_sb; sm.sk = g->c_sk; sm.st = g->c_st;
  sm.cm = g->c_cm > 0 ? g->c_cm : (g->N > 0 ? g->N : 1); sm.co = g->c_co;
  sm.perm = g->c_perm; sm.perm_ld = g->c_perm_ld;
  sm.aux = g->aux; sm.ldaux = g->ldaux;
  return sm;
}

// Epilogue for a wave that owns TM x TN MFMA 32x32 tiles starting at (mrow0, ncol0):
// D[i][j] of a tile: lane holds column j = lane&31, rows (e&3)+8*(e>>2)+4*(lane>>5).
// bias / tanh only on the final (non split-K) pass; optional accumulate; optional layout remap
// (row m = (b*K + k)*T + t, column n = q*cm + r -> b*sb + k*sk + t*st + perm(q)*co + r).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM][TN], float* __restrict__ Cz,
                                              int64_t M, int64_t N, int64_t mrow0, int64_t ncol0,
                                              int lane, const float* __restrict__ bias, int act,
                                              int accumulate, const StoreMap& sm, bool final_pass) {
  int64_t ncol[TN], coff[TN], cq[TN];
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int64_t n = ncol0 + j * 32 + (lane & 31);
    ncol[j] = n;
    bv[j] = (final_pass && bias && n < N) ? bias[n] : 0.f;
    if (sm.remap) {
      cq[j] = n / sm.cm;
      coff[j] = n - cq[j] * sm.cm;
    } else {
      cq[j] = 0;
      coff[j] = n;
    }
  }
#pragma unroll 1
  for (int i = 0; i < TM; ++i) {
#pragma unroll 4
    for (int e = 0; e < 16; ++e) {
      const int64_t m = mrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      int64_t roff, b = 0;
      if (sm.remap) {
        const int64_t t = m % sm.T, q = m / sm.T;
        const int64_t k = q % sm.K;
        b = q / sm.K;
        roff = b * sm.sb + k * sm.sk + t * sm.st;
      } else {
        roff = m * sm.ldc;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (ncol[j] >= N) continue;
        int64_t a = roff + coff[j];
        if (sm.remap) {
          const int64_t cqq = sm.perm ? (int64_t)sm.perm[b * sm.perm_ld + cq[j]] : cq[j];
          a += cqq * sm.co;
        }
        float v = 0.f;      // static register index: no dynamic indexing of the accumulators
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
          for (int ee = 0; ee < 16; ++ee)
            if (ii == i && ee == e) v = acc[ii][j][ee];
        v += bv[j];
        if (final_pass && act == 1) v = gemm_tanh(v);
        if (final_pass && act == 2) { const float y = sm.aux[m * sm.ldaux + ncol[j]]; v *= 1.f - y * y; }
        if (accumulate) v += Cz[a];
        Cz[a] = v;
      }
    }
  }
}

// Row-contiguous epilogue for a wave that owns a 64x64 block (2x2 MFMA tiles) and a private LDS
// scratch of 64 x EPITCH floats: the accumulators are transposed through LDS so that every lane
// stores 16 contiguous bytes and a wave instruction covers 4 rows x 256 B (the per-element variant
// above writes 2 rows x 128 B per instruction with 4-byte lanes and measured 2.4 TB/s alone).
// Plain row-major C only (no remap); falls back to scalar stores on ragged / unaligned edges.
constexpr int EPITCH = 68;
__device__ __forceinline__ void gemm_epilogue_rows(const f32x16 (&acc)[2][2], float* __restrict__ stage,
                                                   float* __restrict__ Cz, int64_t M, int64_t N,
                                                   int64_t mrow0, int64_t ncol0, int lane,
                                                   const float* __restrict__ bias, int act,
                                                   int accumulate, int64_t ldc, bool final_pass,
                                                   const float* __restrict__ aux = nullptr, int64_t ldaux = 0) {
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPITCH + j * 32 + col] = acc[i][j][e];
  // same wave writes and reads: LDS operations of one wave complete in order
  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const int64_t n = ncol0 + c4;
  const bool vec = ((ldc & 3) == 0) && ((((uintptr_t)Cz) & 15) == 0) && n + 3 < N;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (final_pass && bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = n + q < N ? bias[n + q] : 0.f;
  }
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int row = r * 4 + r0;
    const int64_t m = mrow0 + row;
    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * EPITCH + c4);
    if (m >= M || n >= N) continue;
    v += bv;
    if (final_pass && act == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = gemm_tanh(v[q]);
    }
    if (final_pass && act == 2) {
      const float* ya = aux + m * ldaux + n;
      if (vec && ((ldaux & 3) == 0) && ((((uintptr_t)aux) & 15) == 0)) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(ya);
        v *= 1.f - y * y;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < N) v[q] *= 1.f - ya[q] * ya[q];
      }
    }
    float* dst = Cz + m * ldc + n;
    if (vec) {
#ifdef TSSEP_GEMM_EXP
      if (accumulate == 1) v += *reinterpret_cast<const f32x4*>(dst);
      if (accumulate == 2) *reinterpret_cast<f32x4*>(dst) = v;      // (timing probe HACK & 64: temporal stores)
      else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#else
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);    // a boolean, as in every other epilogue
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#endif
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (n + q < N) dst[q] = accumulate ? dst[q] + v[q] : v[q];
    }
  }
}

// Same staging for the REMAPPED store (row m = (b*K + k)*T + t, column n = q*cm + r ->
// b*sb + k*sk + t*st + perm(q)*co + r): a lane owns one column of the 64-wide block, so its column
// decomposition (q, r) is computed once, the row decomposition once per row (wave-uniform), and a
// wave instruction writes one row's 64 consecutive columns (runs of <= cm floats are contiguous in
// the destination).  The per-element variant did two 64-bit divisions per element and ran the
// logit GEMM at 111 TFLOP/s.
__device__ __forceinline__ void gemm_epilogue_rows_remap(const f32x16 (&acc)[2][2], float* __restrict__ stage,
                                                         float* __restrict__ Cz, int64_t M, int64_t N,
                                                         int64_t mrow0, int64_t ncol0, int lane,
                                                         const float* __restrict__ bias, int act,
                                                         int accumulate, const StoreMap& sm) {
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPITCH + j * 32 + col] = acc[i][j][e];
  const int64_t n = ncol0 + lane;
  if (n >= N) return;
  const int64_t cq = n / sm.cm, cr = n - cq * sm.cm;
  const float bv = bias ? bias[n] : 0.f;
  // (b, k, t) of the block's first row by division, of the following rows by carries (wave-uniform): the per-row
  // 64-bit divisions cost the logit GEMM (K = 320) as much as its MFMAs
  int64_t t = mrow0 % sm.T;
  const int64_t q0 = mrow0 / sm.T;
  int64_t k = q0 % sm.K, b = q0 / sm.K;
  int64_t bprev = -1, coff = 0;
  float* rowp = Cz + b * sm.sb + k * sm.sk + t * sm.st;
  for (int row = 0; row < 64; ++row) {
    const int64_t m = mrow0 + row;
    if (m >= M) break;
    if (b != bprev) {            // wave-uniform: the speaker permutation changes with the utterance
      const int64_t cqq = sm.perm ? (int64_t)sm.perm[b * sm.perm_ld + cq] : cq;
      coff = cqq * sm.co + cr;
      bprev = b;
    }
    float v = stage[row * EPITCH + lane] + bv;
    if (act == 1) v = gemm_tanh(v);
    if (act == 2) { const float y = sm.aux[m * sm.ldaux + n]; v *= 1.f - y * y; }
    float* dst = rowp + coff;
    if (accumulate) v += *dst;
    *dst = v;
    rowp += sm.st;
    if (++t == sm.T) {
      t = 0;
      if (++k == sm.K) { k = 0; ++b; }
      rowp = Cz + b * sm.sb + k * sm.sk;
    }
  }
}


// Vector variant of the remapped store for the speaker (un-)combination (net.py:608-611 and its inverse):
// no permutation table, cm and every stride a multiple of 4 floats, T >= 64.  Same 16-byte lanes as
// gemm_epilogue_rows -- the remap only changes each row's base and each lane's column offset: a lane's four
// consecutive columns lie in one column group (cm % 4 == 0), and the (b, k, t) of a row follow from the wave's
// first row by at most one carry per level (64 rows < T), so no per-row division.  The scalar variant above
// stores 4 bytes per lane and ran the combined projection / the un-combining d(input) GEMM ~10 % slower.
__device__ __forceinline__ bool remap_vec_ok(const StoreMap& sm, const float* C) {
  return sm.remap && !sm.perm && sm.T >= 64 && ((sm.cm | sm.co | sm.sb | sm.sk | sm.st) & 3) == 0 &&
         ((((uintptr_t)C) & 15) == 0);
}
__device__ __forceinline__ void gemm_epilogue_rows_remap_vec(const f32x16 (&acc)[2][2], float* __restrict__ stage,
                                                             float* __restrict__ Cz, int64_t M, int64_t N,
                                                             int64_t mrow0, int64_t ncol0, int lane,
                                                             const float* __restrict__ bias, int act,
                                                             int accumulate, const StoreMap& sm) {
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPITCH + j * 32 + col] = acc[i][j][e];
  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const int64_t n = ncol0 + c4;
  if (n >= N) return;
  const bool full = n + 3 < N;
  const int64_t cq = n / sm.cm, coff = cq * sm.co + (n - cq * sm.cm);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = n + q < N ? bias[n + q] : 0.f;
  }
  const bool auxvec = ((sm.ldaux & 3) == 0) && ((((uintptr_t)sm.aux) & 15) == 0);
  // wave-uniform decomposition of the block's first row
  const int64_t q0 = mrow0 / sm.T;
  const int t0 = (int)(mrow0 - q0 * sm.T);
  const int64_t b0 = q0 / sm.K;
  const int k0 = (int)(q0 - b0 * sm.K);
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int row = r * 4 + r0;
    const int64_t m = mrow0 + row;
    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * EPITCH + c4);
    if (m >= M) continue;
    int t = t0 + row, k = k0;
    int64_t b = b0;
    if (t >= sm.T) { t -= (int)sm.T; ++k; }
    if (k >= sm.K) { k -= (int)sm.K; ++b; }
    v += bv;
    if (act == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = gemm_tanh(v[q]);
    }
    if (act == 2) {
      const float* ya = sm.aux + m * sm.ldaux + n;
      if (full && auxvec) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(ya);
        v *= 1.f - y * y;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < N) v[q] *= 1.f - ya[q] * ya[q];
      }
    }
    float* dst = Cz + b * sm.sb + k * sm.sk + t * sm.st + coff;
    if (full) {
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (n + q < N) dst[q] = accumulate ? dst[q] + v[q] : v[q];
    }
  }
}

// Wide variant of the remapped store for everything the vector variant above does not take -- column groups that are
// not multiples of four floats and / or a permutation table (the logit layer: net.py:629-666, 928-967 -- 513 bins per
// speaker, speakers un-permuted per utterance; K = 320, so the 4-byte-per-lane stores of the scalar variant were a third
// of the kernel: 181 TFLOP/s).  Same 16-byte lanes as gemm_epilogue_rows: a lane's four consecutive columns go out as
// ONE store wherever they lie in one column group -- at a 4-byte-aligned address (global memory takes unaligned
// vectors) -- and one by one where they straddle two groups (one lane in cm / 4) or the matrix edge.  A 64-row block
// spans at most two utterances (T >= 64): the column offsets of both are computed once per lane, rows follow from
// the block's first row by carries.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// (StoreMap::remap == 2: the dispatcher's TSSEP_GEMM_REMAP_WIDE=0 -- the scalar variant, for alternating A/Bs)
__device__ __forceinline__ bool remap_wide_ok(const StoreMap& sm) { return sm.remap == 1 && sm.T >= 64 && sm.cm >= 4; }
__device__ __forceinline__ void gemm_epilogue_rows_remap_wide(const f32x16 (&acc)[2][2], float* __restrict__ stage,
                                                              float* __restrict__ Cz, int64_t M, int64_t N,
                                                              int64_t mrow0, int64_t ncol0, int lane,
                                                              const float* __restrict__ bias, int act,
                                                              int accumulate, const StoreMap& sm) {
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPITCH + j * 32 + col] = acc[i][j][e];
  const int c4 = (lane & 15) * 4, r0 = lane >> 4;
  const int64_t n = ncol0 + c4;
  if (n >= N || mrow0 >= M) return;
  const int64_t cq0 = n / sm.cm;
  const int cr0 = (int)(n - cq0 * sm.cm);
  const bool one = n + 3 < N && cr0 + 3 < sm.cm;          // the four columns: inside the matrix and in one group
  // wave-uniform decomposition of the block's first row; the utterance of its last row (at most one further)
  const int64_t q0 = mrow0 / sm.T;
  const int t0 = (int)(mrow0 - q0 * sm.T);
  const int64_t b0 = q0 / sm.K;
  const int k0 = (int)(q0 - b0 * sm.K);
  const int64_t mlast = (mrow0 + 63 < M ? mrow0 + 63 : M - 1);
  const int64_t b1 = mlast / sm.T / sm.K;
  // column offsets of the lane's four columns for utterance b0 ([0]) and b1 ([1])
  int64_t cof[2][4];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool wrap = cr0 + q >= sm.cm;
    const int64_t cq = cq0 + (wrap ? 1 : 0);
    const int64_t cr = cr0 + q - (wrap ? sm.cm : 0);
    const bool in = n + q < N;
    cof[0][q] = ((sm.perm && in) ? (int64_t)sm.perm[b0 * sm.perm_ld + cq] : cq) * sm.co + cr;
    cof[1][q] = ((sm.perm && in) ? (int64_t)sm.perm[b1 * sm.perm_ld + cq] : cq) * sm.co + cr;
    if (bias && in) bv[q] = bias[n + q];
  }
  const bool auxvec = one && ((sm.ldaux & 3) == 0) && ((((uintptr_t)sm.aux) & 15) == 0);
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int row = r * 4 + r0;
    const int64_t m = mrow0 + row;
    f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * EPITCH + c4);
    if (m >= M) continue;
    int t = t0 + row, k = k0;
    int64_t b = b0;
    if (t >= sm.T) { t -= (int)sm.T; ++k; }
    if (k >= sm.K) { k -= (int)sm.K; ++b; }
    v += bv;
    if (act == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = gemm_tanh(v[q]);
    }
    if (act == 2) {
      const float* ya = sm.aux + m * sm.ldaux + n;
      if (auxvec) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(ya);
        v *= 1.f - y * y;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < N) v[q] *= 1.f - ya[q] * ya[q];
      }
    }
    float* rowp = Cz + b * sm.sb + k * sm.sk + t * sm.st;
    const bool second = b != b0;
    if (one) {
      float* dst = rowp + (second ? cof[1][0] : cof[0][0]);
      if (accumulate) v += *reinterpret_cast<const f32x4u*>(dst);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4u*>(dst));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (n + q < N) {
          float* dst = rowp + (second ? cof[1][q] : cof[0][q]);
          *dst = accumulate ? *dst + v[q] : v[q];
        }
    }
  }
}

// ---- pieces shared by the split-bf16 kernels (gemm_bf16x3.hip, gemm_bf16x3_stream.hip) -------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Buffer loads for the steady-state operand streams: one buffer resource (4 SGPRs: base + range) per operand,
// ONE 32-bit VGPR byte offset per lane and operand, the per-load part of the address (row group, K tile) in an
// SGPR.  With plain pointers the compiler kept a 64-bit VGPR address per load (8 in the weight-gradient kernel,
// 6 in the tall one) and re-derived them with v_lshl_add_u64 / v_mov_b64 every K tile: 12-16 VGPRs and ~20 VALU
// instructions per tile.  The range only has to cover one tile's rows (offsets stay far below 2 GB).
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(srd_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ float bload1(srd_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}

// hi = bf16_rne(x), lo = bf16_rne(x - hi) for two values, packed as the MFMA operands want them
__device__ __forceinline__ void split2n(float a, float b, unsigned& hi, unsigned& lo) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);          // v_cvt_pk_bf16_f32 (RNE)
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(v - hf, bf16x2);     // v - hf is exact in fp32
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// bf16_rne of two values, packed (the single-product side line: precision = 3)
__device__ __forceinline__ unsigned bf16_pair(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

}  // namespace gemm_detail

// split-bf16 (bf16x3) family, defined in gemm_bf16x3*.hip.  Every launcher checks what its kernel REQUIRES and returns
// TSSEP_E_UNSUPPORTED otherwise; with call.dry it stops in front of the launch (plan query).  gemm_dispatch.h.
int tssep_gemm_bf16x3_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, gemm_detail::GemmCall& call);
// persistent streaming variant (row x row, plain store), gemm_bf16x3_stream.hip
int tssep_gemm_bf16x3_stream_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call);
// 256 x 256 tile with 128 x 128 wave tiles, one wave per SIMD (gemm_bf16x3_big.hip); every epilogue option
int tssep_gemm_bf16x3_big_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call);
// ... persistent over the tile list, plain / bias / Tanh stores (gemm_bf16x3_bigp.hip)
int tssep_gemm_bf16x3_bigp_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call);
// ... 192 x 320 tile for N = 320 q (gemm_bf16x3_bigp320.hip)
int tssep_gemm_bf16x3_bigp320_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call);
// weight gradients (both operands k-major, no time shift): 512 x 128 tile, 128 x 128 wave tiles, three LDS stages
// (gemm_bf16x3_tn_big.hip)
int tssep_gemm_bf16x3_tn_big_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two, const gemm_detail::GemmCall& call);
// ... 192 x 320 tile for N = 320 q (+ the ones column), wave tiles of 96 x 160 (gemm_bf16x3_tn_p320.hip)
int tssep_gemm_bf16x3_tn_p320_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two, const gemm_detail::GemmCall& call);
// (gemm_bf16x3_tn_w160.hip, gemm_bf16x3_tn_h160.hip)
namespace gemm_detail {
// tn_w160 runs eight-wave workgroups (512 threads, ONE per CU; masks by out-of-range loads) where the shape allows:
// 5 -> 256 x 320 tiles: M and the MFMA columns multiples of 4; an even number of 160-column tiles without the ones column
//      (also time-shifted), or (unshifted) N = 320 q (+ the ones column), or a ragged last tile where 320-wide tiles pad no
//      more than 128-wide ones, with one more real column (N % 4 == 1: the pre-net's 553) and the ones column on the VALU;
// 4 -> 256 x 256 tiles (unshifted): N = 256 q + XR + XO, q >= 2, XR <= 1 more real column and the ones column (XO) on the
//      VALU -- dW_ih of birnn0: 513 + 1;
// 7 -> 256 x 320 tiles with SWAPPED operands (unshifted): few output rows (192 < M <= 320: one 320-column tile of the
//      swapped problem), many columns (N = 4 q >= 512, padding to 256-row tiles by at most 30 %) -- the projection weight
//      gradients (320 x 600 + 1); the ones column becomes a row of ones on the MFMAs;
// 0 -> the 256 x 160 workgroups.  The launcher, the dispatcher and the split rule (gemm.hip) all ask here.
inline int tn_w160_wide(const tssep_gemm_args* g) {
  if (g->M & 3) return 0;
  const int64_t xo = g->b_ones_col ? 1 : 0, nr = g->N - xo;
  if (g->kperiod <= 0 && g->M > 192 && g->M <= 320 && nr >= 512 && (nr & 3) == 0 && ((nr + xo + 255) / 256 * 256) * 10 <= nr * 13) return 7;
  if (g->kperiod <= 0 && nr >= 320 && nr % 320 == 0) return 5;
  if (!xo && (g->N & 3) == 0 && (((g->N + 159) / 160) & 1) == 0) return 5;
  if (g->kperiod <= 0 && nr >= 512 && nr % 256 <= 1) return 4;
  if (g->kperiod <= 0 && nr > 320 && (nr & 3) <= 1 && (nr - (nr & 3) + 319) / 320 * 320 <= (nr + 127) / 128 * 128) return 5;
  return 0;
}
// columns of the MFMA tiles of such a request (the one more real column and the ones column come on top)
inline int64_t tn_w160_wide_cols(const tssep_gemm_args* g, int wide) {
  const int64_t nr = g->N - (g->b_ones_col ? 1 : 0);
  if (wide == 7) return g->M;      // (of the swapped problem)
  return wide == 4 ? nr - nr % 256 : nr - (nr & 3);
}
}  // namespace gemm_detail
int tssep_gemm_bf16x3_tn_w160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two, const gemm_detail::GemmCall& call);
int tssep_gemm_bf16x3_tn_h160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk, int two, const gemm_detail::GemmCall& call);
// (gemm_bf16x3_nt_w160.hip)
int tssep_gemm_bf16x3_nt_w160_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, const gemm_detail::GemmCall& call);
