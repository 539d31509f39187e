// Exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.Linear / the LSTM input projection of the reference
// (tssep/train/rnnp.py:88-96,146-161; tssep/train/net.py:663-666) and the three
// autograd GEMMs behind each of them (dgrad, wgrad).  The f32-input MFMA is
// bit-for-bit an fmaf chain, so results are fp32-exact in the sense the 1e-3
// parity bar needs; rate 64 FLOP/clk/SIMD (157 TFLOP/s chip peak).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 2x2 MFMA tiles of 32x32 -> 64 accumulator VGPRs), BK = 16.  Operands are
// staged global -> registers -> LDS (register prefetch of tile t+1 while tile t is
// multiplied, two LDS buffers, one barrier per K tile).  Within a K tile the
// MFMA k-pair of step s is {s, s+8}: the lower half-wave walks k = 0..7, the upper
// k = 8..15, so a "row" operand (k contiguous in memory) is read from LDS with two
// ds_read_b128 per 32 rows and 8 MFMA steps.  The [BK+4]-float row pitch makes
// those reads bank-conflict free (pitch 80 B: 16-B slot index 5*row mod 16 is a
// bijection over the 16 rows of a ds_read_b128 lane group).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NTHREADS = 256;
constexpr int ROW_PITCH = BK + 4;     // floats, "row" operand  S[128][20]
constexpr int COL_PITCH = BM + 4;     // floats, "col" operand  S[16][132]
constexpr int OP_FLOATS = BM * ROW_PITCH;  // 2560 >= 16*132 = 2112

struct StoreMap {
  int64_t ldc;
  int32_t remap;
  int64_t T, K, sb, sk, st, cm, co;
  const int32_t* perm; int64_t perm_ld;
};

__device__ __forceinline__ int64_t c_addr(const StoreMap& s, int64_t m, int64_t n) {
  if (!s.remap) return m * s.ldc + n;
  int64_t t = m % s.T, q = m / s.T;
  int64_t k = q % s.K, b = q / s.K;
  int64_t cq = n / s.cm, cr = n - cq * s.cm;
  if (s.perm) cq = s.perm[b * s.perm_ld + cq];
  return b * s.sb + k * s.sk + t * s.st + cq * s.co + cr;
}

// ---- global -> register tile loads -------------------------------------------------
// "row" operand: element (r, k) at P[r*ld + k]; tile = 128 rows x 16 k; thread holds 2 float4.
__device__ __forceinline__ void load_row_tile(const float* __restrict__ P, int64_t ld, int64_t R,
                                              int64_t K, int64_t r0, int64_t k0, int tid,
                                              f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    int64_t r = r0 + (q >> 2);
    int64_t k = k0 + ((q & 3) << 2);
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const float* p = P + r * ld + k;
      if (k + 4 <= K) {
        x = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
      }
    }
    v[i] = x;
  }
}
// "col" operand: element (k, c) at P[k*ld + c]; tile = 16 k x 128 cols; thread holds 2 float4.
// kshift/kperiod: row k is read at k+kshift and is zero when (k % kperiod)+kshift leaves [0,kperiod).
__device__ __forceinline__ void load_col_tile(const float* __restrict__ P, int64_t ld, int64_t Ccols,
                                              int64_t K, int64_t c0, int64_t k0, int tid,
                                              int64_t kshift, int64_t kperiod, f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    int64_t k = k0 + (q >> 5);
    int64_t c = c0 + ((q & 31) << 2);
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    bool ok = k < K;
    int64_t kk = k;
    if (kperiod > 0) {
      int64_t ph = (k % kperiod) + kshift;
      ok = ok && ph >= 0 && ph < kperiod;
      kk = k + kshift;
    }
    if (ok) {
      const float* p = P + kk * ld + c;
      if (c + 4 <= Ccols) {
        x = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c + e < Ccols) x[e] = p[e];
      }
    }
    v[i] = x;
  }
}
__device__ __forceinline__ void store_row_tile(float* S, int tid, const f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    *reinterpret_cast<f32x4*>(S + (q >> 2) * ROW_PITCH + ((q & 3) << 2)) = v[i];
  }
}
__device__ __forceinline__ void store_col_tile(float* S, int tid, const f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    *reinterpret_cast<f32x4*>(S + (q >> 5) * COL_PITCH + ((q & 31) << 2)) = v[i];
  }
}
// ---- LDS -> MFMA operand fragments: frag[s] = element (row = base + (lane&31), k = 8*(lane>>5) + s)
template <bool KMAJOR>
__device__ __forceinline__ void read_frag(const float* S, int base, int lane, float (&f)[8]) {
  const int r = base + (lane & 31), h = lane >> 5;
  if (!KMAJOR) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(S + r * ROW_PITCH + h * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(S + r * ROW_PITCH + h * 8 + 4);
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
    f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = S[(h * 8 + s) * COL_PITCH + r];
  }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
    int64_t b_kshift, int64_t kperiod, const float* __restrict__ bias, int act, int accumulate,
    StoreMap sm, int splitk, int64_t c_split_stride) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][OP_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  // split-K range (in K tiles)
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)blockIdx.z * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[2], rb[2];
  auto gload = [&](int64_t kt) {
    const int64_t k0 = kt * BK;
    if (!A_KMAJOR) load_row_tile(A, lda, M, K, m0, k0, tid, ra);
    else load_col_tile(A, lda, M, K, m0, k0, tid, 0, 0, ra);
    if (!B_KMAJOR) load_row_tile(B, ldb, N, K, n0, k0, tid, rb);
    else load_col_tile(B, ldb, N, K, n0, k0, tid, b_kshift, kperiod, rb);
  };
  auto sstore = [&](int buf) {
    if (!A_KMAJOR) store_row_tile(lds[buf][0], tid, ra); else store_col_tile(lds[buf][0], tid, ra);
    if (!B_KMAJOR) store_row_tile(lds[buf][1], tid, rb); else store_col_tile(lds[buf][1], tid, rb);
  };

  if (kt_begin < kt_end) {
    gload(kt_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 1 < kt_end;
      if (more) gload(kt + 1);
      float fa[2][8], fb[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i) read_frag<A_KMAJOR>(lds[buf][0], wm * 64 + i * 32, lane, fa[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) read_frag<B_KMAJOR>(lds[buf][1], wn * 64 + j * 32, lane, fb[j]);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
      if (more) sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // epilogue: D[i][j] of a 32x32 tile: lane holds column j = lane&31, rows (e&3)+8*(e>>2)+4*(lane>>5)
  float* Cz = C + (int64_t)blockIdx.z * c_split_stride;
  const bool final_pass = splitk == 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
      if (n >= N) continue;
      const float bv = (final_pass && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        float v = acc[i][j][e] + bv;
        if (final_pass && act == 1) v = tanhf(v);
        const int64_t a = c_addr(sm, m, n);
        if (accumulate) v += Cz[a];
        Cz[a] = v;
      }
    }
}

}  // namespace

extern "C" int tssep_gemm_f32(const tssep_gemm_args* g, void* stream) {
  if (!g || !g->A || !g->B || !g->C) return TSSEP_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(g->A) || !aligned16(g->B) || (g->lda & 3) || (g->ldb & 3)) return TSSEP_E_ALIGN;
  const int splitk = g->splitk > 1 ? g->splitk : 1;
  if (splitk > 1 && (g->bias || g->act || g->c_remap)) return TSSEP_E_UNSUPPORTED;
  if (g->a_kmajor && !g->b_kmajor) return TSSEP_E_UNSUPPORTED;
  if (g->kperiod > 0 && !g->b_kmajor) return TSSEP_E_UNSUPPORTED;
  StoreMap sm;
  sm.ldc = g->ldc; sm.remap = g->c_remap;
  sm.T = g->c_T > 0 ? g->c_T : 1; sm.K = g->c_K > 0 ? g->c_K : 1;
  sm.sb = g->c_sb; sm.sk = g->c_sk; sm.st = g->c_st;
  sm.cm = g->c_cm > 0 ? g->c_cm : (g->N > 0 ? g->N : 1); sm.co = g->c_co;
  sm.perm = g->c_perm; sm.perm_ld = g->c_perm_ld;
  dim3 grid((unsigned)((g->N + BN - 1) / BN), (unsigned)((g->M + BM - 1) / BM), (unsigned)splitk);
  if (grid.y > 65535u) return TSSEP_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(AK, BKM)                                                                          \
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM>), grid, dim3(NTHREADS), 0, s, g->A, g->B, g->C,   \
                     g->M, g->N, g->K, g->lda, g->ldb, g->b_kshift, g->kperiod, g->bias, g->act, \
                     g->accumulate, sm, splitk, g->c_split_stride)
  if (!g->a_kmajor && !g->b_kmajor) LAUNCH(false, false);
  else if (!g->a_kmajor && g->b_kmajor) LAUNCH(false, true);
  else LAUNCH(true, true);
#undef LAUNCH
  return tssep_launch_status();
}
