// Shared pieces of the two GEMM kernels (exact fp32 and split-bf16): output mapping + epilogue.
#pragma once
#include "common.h"

namespace gemm_detail {

constexpr int BM = 128, BN = 128, NTHREADS = 256;

struct StoreMap {
  int64_t ldc;
  int32_t remap;
  int64_t T, K, sb, sk, st, cm, co;
  const int32_t* perm; int64_t perm_ld;
};

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
        if (final_pass && act == 1) v = tanhf(v);
        if (accumulate) v += Cz[a];
        Cz[a] = v;
      }
    }
  }
}

}  // namespace gemm_detail

// split-bf16 (bf16x3) variant, defined in gemm_bf16x3.hip
int tssep_gemm_bf16x3_launch(const tssep_gemm_args* g, const gemm_detail::StoreMap& sm, int splitk,
                             void* stream);
