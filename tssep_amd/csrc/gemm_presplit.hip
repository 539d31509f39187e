// EXPERIMENT (round-1 probe for the next GEMM design, see DESIGN.md "Next" 1): split-bf16 GEMM whose
// operands arrive PRE-SPLIT (bf16 hi and lo planes, [rows][Kp] with Kp = K rounded up to 16, zero padded),
// so that a K tile is staged by asynchronous global -> LDS copies only: no split VALU, no ds_write, no
// staging registers, and a three-deep LDS ring at the same 72 KB per workgroup the production kernel
// uses for two stages.  Same tile (256 x 128 x 16, 4 waves, wave tile 128 x 64) and the same MFMA order
// as gemm_bf16x3_tall_kernel, so the results are bit-identical to it.
//
// LDS image of one plane: rows of 16 bf16 = 32 B, two 16-B slots per row; a copy instruction moves 1 KB
// = 32 rows lane-linearly, so the layout is fixed -- the slot order inside a row is XOR-swizzled with
// bit 3 of the row ON THE SOURCE SIDE so that the 16 rows of a ds_read_b128 lane group hit 16 distinct
// 16-byte bank groups.
// The copies are issued with inline asm (the compiler orders every LDS read behind any copy it knows of
// with vmcnt(0), which would serialise the ring) and waited for with an explicit vmcnt(6): the six
// copies of the NEXT tile may stay in flight.
#include "gemm_common.h"

namespace {
using namespace gemm_detail;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PBM = 256, PBK = 16;
constexpr int PA = PBM * 32;                           // bytes of one A plane of one stage

// ktile_major: element (r, k) at ((k / 16) * rows + r) * 16 + k % 16 -- the 32 rows x 16 k of one copy
// instruction are then 1 KB CONTIGUOUS (whole cache lines), instead of 32-byte pieces of 32 different lines
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int64_t rows,
                                                           int64_t K, int64_t ld, int64_t Kp,
                                                           __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                           int ktile_major) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Kp) return;
  const int64_t r = i / Kp, k = i - r * Kp;
  const float v = k < K ? x[r * ld + k] : 0.f;
  const __bf16 h = (__bf16)v;
  const int64_t o = ktile_major ? ((k >> 4) * rows + r) * 16 + (k & 15) : i;
  hi[o] = h;
  lo[o] = (__bf16)(v - (float)h);
}

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
  // 64 lanes x 16 B -> LDS [lds_addr, lds_addr + 1024), lane-linear
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}

// WN = 2: 256 x 128 tile, 4 waves, two workgroups per CU.  WN = 4: 256 x 256 tile, 8 waves, one workgroup per
// CU (32 KB per stage: 33 % fewer operand bytes per MFMA).
template <int WN>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) void gemm_presplit_kernel(
    const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al, const __bf16* __restrict__ Bh,
    const __bf16* __restrict__ Bl, float* __restrict__ C, int64_t M, int64_t N, int64_t Kp,
    StoreMap sm, const float* __restrict__ bias, int act, int accumulate, TileMap tmap, int ring_flags) {
  // ablation flags (timing experiments only, results are then wrong): 256 = no MFMA, 512 = no fragment
  // reads, 1024 = no copies, 2048 = no barriers
  const int ring = ring_flags & 255;
  const bool ktm = ring_flags & 4096;       // planes are k-tile-major
  const bool no_mfma = ring_flags & 256, no_frag = ring_flags & 512, no_dma = ring_flags & 1024, no_bar = ring_flags & 2048;
  constexpr int PBN = 64 * WN, PB = PBN * 32, PSTAGE = 2 * PA + 2 * PB, NW = 2 * WN;
  constexpr int NPIECE = PSTAGE / 1024, PPW = NPIECE / NW;     // copies per stage, per wave
  constexpr int EPI = NW * 64 * EPITCH * 4;
  __shared__ __attribute__((aligned(1024))) char lds[3 * PSTAGE > EPI ? 3 * PSTAGE : EPI];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * PBM, n0 = (int64_t)nt * PBN;
  const int64_t ktiles = Kp / PBK;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // copies of this wave: pieces p = wave + 4 q (q = 0..5) of the stage's 24 KB; piece p covers 32 rows
  // of plane {A hi: 0-7, A lo: 8-15, B hi: 16-19, B lo: 20-23}; lane l fetches row l/2, 16-B slot
  // (l&1) ^ bit3(row)
  const char* src[PPW];
  int64_t kstride[PPW];
  unsigned dst[PPW];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = wave + NW * q;
    const bool isA = p < 16;
    const int blk = isA ? (p & 7) : ((p - 16) % (2 * WN));
    const bool lo = isA ? p >= 8 : p >= 16 + 2 * WN;
    const int r = blk * 32 + (lane >> 1);
    const int h = (lane & 1) ^ ((r >> 3) & 1);
    int64_t grow = (isA ? m0 : n0) + r;
    const int64_t lim = (isA ? M : N) - 1;
    grow = grow > lim ? lim : grow;
    const __bf16* plane = isA ? (lo ? Al : Ah) : (lo ? Bl : Bh);
    src[q] = reinterpret_cast<const char*>(plane + (ktm ? grow * 16 : grow * Kp)) + h * 16;
    kstride[q] = ktm ? (isA ? M : N) * 32 : PBK * 2;
    dst[q] = (unsigned)((isA ? (lo ? PA : 0) : 2 * PA + (lo ? PB : 0)) + blk * 1024);
  }
  auto issue = [&](int64_t kt, int stage) {
    if (no_dma) return;
#pragma unroll
    for (int q = 0; q < PPW; ++q) dma16(src[q] + kt * kstride[q], lds_base + stage * PSTAGE + dst[q]);
  };
  // fragment slots of this lane
  const int fr = lane & 31, fh = lane >> 5;
  int aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 128 + i * 32 + fr;
    aoff[i] = (2 * r + (fh ^ ((r >> 3) & 1))) * 16;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wn * 64 + j * 32 + fr;
    boff[j] = 2 * PA + (2 * r + (fh ^ ((r >> 3) & 1))) * 16;
  }
  bf16x8 kfrag;
#pragma unroll
  for (int e = 0; e < 8; ++e) kfrag[e] = (__bf16)(0.001f * (lane + e));
  auto compute = [&](const char* st) {
    bf16x8 ah[4], al[4], bh[2], bl[2];
    if (no_frag) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ah[i] = kfrag; al[i] = kfrag; }
#pragma unroll
      for (int j = 0; j < 2; ++j) { bh[j] = kfrag; bl[j] = kfrag; }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(st + aoff[i]);
      al[i] = *reinterpret_cast<const bf16x8*>(st + PA + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(st + boff[j]);
      bl[j] = *reinterpret_cast<const bf16x8*>(st + PB + boff[j]);
    }
    }
    if (no_mfma) {
      acc[0][0][0] += (float)ah[0][0] + (float)al[3][1] + (float)bh[1][2] + (float)bl[0][3] + (float)ah[2][4] + (float)bh[0][5];
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };

  // ring of `ring` stages (2 or 3): tile kt lives in stage kt % ring; `ring - 1` tiles are in flight
  issue(0, 0);
  if (ring == 3 && ktiles > 1) issue(1, 1);
  int cur = 0;
  for (int64_t kt = 0; kt < ktiles; ++kt) {
    const int64_t nxt_tile = kt + ring - 1;
    // tile kt has landed when at most the copies of the younger in-flight tile are outstanding
    if (ring == 3 && kt + 1 < ktiles) {
      if constexpr (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!no_bar) __syncthreads();          // every wave's copies of tile kt are in; every wave is done with tile kt - 1
    if (nxt_tile < ktiles) {
      int st = cur + ring - 1;
      st = st >= ring ? st - ring : st;
      issue(nxt_tile, st);
    }
    compute(lds + cur * PSTAGE);
    cur = cur + 1 == ring ? 0 : cur + 1;
  }
  __syncthreads();
  float* stage = reinterpret_cast<float*>(lds) + wave * 64 * EPITCH;
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {
    f32x16 a2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) a2[i][j] = acc[2 * ih + i][j];
    if (!sm.remap)
      gemm_epilogue_rows(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64, lane,
                         bias, act, accumulate, sm.ldc, true, sm.aux, sm.ldaux);
    else if (remap_vec_ok(sm, C))
      gemm_epilogue_rows_remap_vec(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64,
                                   lane, bias, act, accumulate, sm);
    else
      gemm_epilogue_rows_remap(a2, stage, C, M, N, m0 + (int64_t)wm * 128 + ih * 64, n0 + (int64_t)wn * 64,
                               lane, bias, act, accumulate, sm);
  }
}


// ---- weight-gradient counterpart: C[M,N] = A^T B, A = dY [K rows][M], B = X [K rows][N], both given as
// planes [cols/16][K][16] (the SAME k-tile-major planes the forward / dgrad GEMMs of the same tensors would
// read).  A K tile of 32 rows of a 16-column chunk is 1 KB contiguous: one copy instruction; in LDS it stays
// [chunk][32 rows][16 cols] and the MFMA fragments come from ds_read_b64_tr_b16 exactly as in
// gemm_bf16x3_tn_kernel (a 16-lane group reads 4 rows x 16 columns = 128 contiguous bytes).  Same tile
// (128 x 128 x 32), split-K and MFMA order as that kernel -> bit-identical partials.
typedef short s16x4p __attribute__((ext_vector_type(4)));
typedef short s16x8p __attribute__((ext_vector_type(8)));
constexpr int QCH = 1024 + 128;               // LDS stride of a chunk (the pad separates the two 16-lane groups)
constexpr int QPLANE = 8 * QCH, QSTAGE = 4 * QPLANE;      // A hi, A lo, B hi, B lo = 36 864 B

__device__ __forceinline__ bf16x8 tr_frag_p(const char* p) {
  const s16x4p a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4p*)(p));
  const s16x4p b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4p*)(p + 4 * 32));
  const s16x8p v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int RING>
__global__ __launch_bounds__(256, RING == 2 ? 2 : 1) void gemm_presplit_tn_kernel(
    const __bf16* __restrict__ Ah, const __bf16* __restrict__ Al, const __bf16* __restrict__ Bh,
    const __bf16* __restrict__ Bl, float* __restrict__ C, int64_t M, int64_t N, int64_t K, int64_t ldc,
    int splitk, int64_t c_split_stride, int accumulate, TileMap tmap) {
  constexpr int ring = RING;
  __shared__ __attribute__((aligned(1024))) char lds[RING * QSTAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * 128, n0 = (int64_t)nt * 128;
  const int64_t ktiles = K / 32;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // copies of this wave: pieces p = wave + 4 q (q = 0..7); piece p = plane p / 8, chunk p % 8
  const char* src[8];
  unsigned dst[8];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int64_t mchunks = (M + 15) / 16, nchunks = (N + 15) / 16;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int p = wave + 4 * q, plane = p >> 3, ch = p & 7;
    const bool isA = plane < 2;
    int64_t c = (isA ? m0 : n0) / 16 + ch;
    const int64_t cl = (isA ? mchunks : nchunks) - 1;
    c = c > cl ? cl : c;
    const __bf16* base = plane == 0 ? Ah : plane == 1 ? Al : plane == 2 ? Bh : Bl;
    src[q] = reinterpret_cast<const char*>(base + (c * K) * 16) + lane * 16;          // + k0 * 32 bytes per tile row
    dst[q] = (unsigned)(plane * QPLANE + ch * QCH);
  }
  auto issue = [&](int64_t kt, int stage) {
#pragma unroll
    for (int q = 0; q < 8; ++q) dma16(src[q] + kt * (32 * 32), lds_base + stage * QSTAGE + dst[q]);
  };
  const int ii = lane & 15, g2 = (lane >> 4) & 1, hk = lane >> 5;
  const int foff = (8 * hk + (ii >> 2)) * 32 + (ii & 3) * 8;
  auto compute = [&](const char* st) {
    bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = (wm * 4 + 2 * i + g2) * QCH + ks * 16 * 32 + foff;
        const int cb = (wn * 4 + 2 * i + g2) * QCH + ks * 16 * 32 + foff;
        ah[ks][i] = tr_frag_p(st + ca);
        al[ks][i] = tr_frag_p(st + QPLANE + ca);
        bh[ks][i] = tr_frag_p(st + 2 * QPLANE + cb);
        bl[ks][i] = tr_frag_p(st + 3 * QPLANE + cb);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    }
  };
  if (kt_begin < kt_end) {
    issue(kt_begin, 0);
    if (ring == 3 && kt_begin + 1 < kt_end) issue(kt_begin + 1, 1);
    int cur = 0;
    for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
      if (ring == 3 && kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int64_t nxt = kt + ring - 1;
      if (nxt < kt_end) {
        int stg = cur + ring - 1;
        stg = stg >= ring ? stg - ring : stg;
        issue(nxt, stg);
      }
      compute(lds + cur * QSTAGE);
      cur = cur + 1 == ring ? 0 : cur + 1;
    }
  }
  __syncthreads();
  float* Cz = C + (int64_t)zsplit * c_split_stride;
  float* stage = reinterpret_cast<float*>(lds) + wave * 64 * EPITCH;
  static_assert(4 * 64 * EPITCH * 4 <= RING * QSTAGE, "epilogue scratch");
  gemm_epilogue_rows(acc, stage, Cz, M, N, m0 + (int64_t)wm * 64, n0 + (int64_t)wn * 64, lane, nullptr, 0,
                     accumulate, ldc, splitk == 1);
}

StoreMap plain_map(int64_t ldc, int64_t N) {
  StoreMap sm;
  sm.ldc = ldc; sm.remap = 0; sm.T = 1; sm.K = 1; sm.sb = 0; sm.sk = 0; sm.st = 0; sm.cm = N > 0 ? N : 1; sm.co = 0;
  sm.perm = nullptr; sm.perm_ld = 0;
  return sm;
}

}  // namespace

extern "C" int tssep_probe_split_planes(const float* x, int64_t rows, int64_t K, int64_t ld, void* hi,
                                        void* lo, int ktile_major, void* stream) {
  if (!x || !hi || !lo) return TSSEP_E_NULL;
  if (rows <= 0 || K <= 0 || ld < K) return TSSEP_E_SHAPE;
  const int64_t Kp = (K + 15) / 16 * 16;
  const int64_t n = rows * Kp;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, rows, K, ld, Kp, (__bf16*)hi, (__bf16*)lo, ktile_major);
  return tssep_launch_status();
}

extern "C" int tssep_probe_gemm_presplit(const void* a_hi, const void* a_lo, const void* b_hi,
                                         const void* b_lo, float* C, int64_t M, int64_t N, int64_t K,
                                         int64_t ldc, int ring, void* stream) {
  if (!a_hi || !a_lo || !b_hi || !b_lo || !C) return TSSEP_E_NULL;
  const int flags = ring & ~255;
  ring &= 255;
  if (M <= 0 || N <= 0 || K <= 0 || ldc < N || (ring != 2 && ring != 3 && ring != 12 && ring != 13)) return TSSEP_E_SHAPE;
  if (!aligned16(a_hi) || !aligned16(a_lo) || !aligned16(b_hi) || !aligned16(b_lo)) return TSSEP_E_ALIGN;
  const int64_t Kp = (K + 15) / 16 * 16;
  // ring 2 / 3: 256 x 128 tile (4 waves); ring 12 / 13: 256 x 256 tile (8 waves), ring 2 / 3
  if (ring >= 12) {
    const TileMap tm = make_tile_map((M + PBM - 1) / PBM, (N + 255) / 256, 1);
    hipLaunchKernelGGL(gemm_presplit_kernel<4>, dim3((unsigned)tile_map_blocks(tm)), dim3(512), 0,
                       (hipStream_t)stream, (const __bf16*)a_hi, (const __bf16*)a_lo, (const __bf16*)b_hi,
                       (const __bf16*)b_lo, C, M, N, Kp, plain_map(ldc, N), nullptr, 0, 0, tm, (ring - 10) | flags);
    return tssep_launch_status();
  }
  const TileMap tm = make_tile_map((M + PBM - 1) / PBM, (N + 127) / 128, 1);
  hipLaunchKernelGGL(gemm_presplit_kernel<2>, dim3((unsigned)tile_map_blocks(tm)), dim3(256), 0,
                     (hipStream_t)stream, (const __bf16*)a_hi, (const __bf16*)a_lo, (const __bf16*)b_hi,
                     (const __bf16*)b_lo, C, M, N, Kp, plain_map(ldc, N), nullptr, 0, 0, tm, ring | flags);
  return tssep_launch_status();
}

extern "C" int tssep_probe_gemm_presplit_tn(const void* a_hi, const void* a_lo, const void* b_hi,
                                            const void* b_lo, float* C, int64_t M, int64_t N, int64_t K,
                                            int64_t ldc, int splitk, int64_t c_split_stride, int ring,
                                            void* stream) {
  if (!a_hi || !a_lo || !b_hi || !b_lo || !C) return TSSEP_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || ldc < N || splitk < 1 || (ring != 2 && ring != 3)) return TSSEP_E_SHAPE;
  const TileMap tm = make_tile_map((M + 127) / 128, (N + 127) / 128, splitk);
  if (ring == 2)
    hipLaunchKernelGGL(gemm_presplit_tn_kernel<2>, dim3((unsigned)tile_map_blocks(tm)), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)a_hi, (const __bf16*)a_lo, (const __bf16*)b_hi,
                       (const __bf16*)b_lo, C, M, N, K, ldc, splitk, c_split_stride, 0, tm);
  else
    hipLaunchKernelGGL(gemm_presplit_tn_kernel<3>, dim3((unsigned)tile_map_blocks(tm)), dim3(256), 0,
                       (hipStream_t)stream, (const __bf16*)a_hi, (const __bf16*)a_lo, (const __bf16*)b_hi,
                       (const __bf16*)b_lo, C, M, N, K, ldc, splitk, c_split_stride, 0, tm);
  return tssep_launch_status();
}

// GEMM on plane operands with the production argument block (groundwork for round 2: producers emit the
// planes, this replaces tssep_gemm_f32 precision 1).  g->A / g->B are the HI planes, a_lo / b_lo the LO
// planes; lda / ldb are ignored (the planes are k-tile-major, see tssep_probe_split_planes).
//   a_kmajor = b_kmajor = 0: C = epilogue(A B^T), planes [K/16][M][16] and [K/16][N][16]: bias, tanh,
//                            accumulate, store remap as in tssep_gemm_f32.
//   a_kmajor = b_kmajor = 1: C (split-K partials) = A^T B, planes [M/16][K][16] and [N/16][K][16], K % 32 == 0
//                            (producers pad the rows with zeros), accumulate; no time shift, no virtual
//                            ones column (the producer writes a real one).
extern "C" int tssep_gemm_planes(const tssep_gemm_args* g, const void* a_lo, const void* b_lo, void* stream) {
  if (!g || !g->A || !g->B || !g->C || !a_lo || !b_lo) return TSSEP_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(g->A) || !aligned16(g->B) || !aligned16(a_lo) || !aligned16(b_lo)) return TSSEP_E_ALIGN;
  if (g->kperiod > 0 || g->b_ones_col || g->a_kmajor != g->b_kmajor) return TSSEP_E_UNSUPPORTED;
  const int splitk = g->splitk > 1 ? g->splitk : 1;
  hipStream_t s = (hipStream_t)stream;
  if (!g->a_kmajor) {
    if (splitk > 1) return TSSEP_E_UNSUPPORTED;
    const int64_t Kp = (g->K + 15) / 16 * 16;
    const TileMap tm = make_tile_map((g->M + PBM - 1) / PBM, (g->N + 127) / 128, 1);
    hipLaunchKernelGGL(gemm_presplit_kernel<2>, dim3((unsigned)tile_map_blocks(tm)), dim3(256), 0, s,
                       (const __bf16*)g->A, (const __bf16*)a_lo, (const __bf16*)g->B, (const __bf16*)b_lo, g->C,
                       g->M, g->N, Kp, make_store_map(g), g->bias, g->act, g->accumulate, tm, 2 | 4096);
    return tssep_launch_status();
  }
  if ((g->K & 31) || g->bias || g->act || g->c_remap) return TSSEP_E_UNSUPPORTED;
  const TileMap tm = make_tile_map((g->M + 127) / 128, (g->N + 127) / 128, splitk);
  hipLaunchKernelGGL(gemm_presplit_tn_kernel<2>, dim3((unsigned)tile_map_blocks(tm)), dim3(256), 0, s,
                     (const __bf16*)g->A, (const __bf16*)a_lo, (const __bf16*)g->B, (const __bf16*)b_lo, g->C, g->M,
                     g->N, g->K, g->ldc, splitk, g->c_split_stride, g->accumulate, tm);
  return tssep_launch_status();
}
