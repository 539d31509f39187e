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
#include "gemm_common.h"

namespace {

using namespace gemm_detail;
constexpr int BK = 16;
constexpr int ROW_PITCH = BK + 4;     // floats, "row" operand  S[128][20]
constexpr int COL_PITCH = BM + 4;     // floats, "col" operand  S[16][132]
constexpr int OP_FLOATS = BM * ROW_PITCH;  // 2560 >= 16*132 = 2112

// ---- per-thread load descriptors -----------------------------------------------------------
// Every thread owns two 16-byte loads per operand per K tile.  Pointers are set up once (rows /
// columns outside the matrix are CLAMPED to a valid address: their products only reach outputs
// that are never stored), so the steady-state loop issues four unconditional global_load_dwordx4
// and no address arithmetic beyond one add.  Only the last, partial K tile applies element masks.
// ROWS = tile extent along the operand's non-K dimension (128, or 32 for the narrow edge tile:
// then only the first 128 threads load).
struct RowLoad {                 // "row" operand: element (r, k) at P[r*ld + k]
  const float* p[2];             // row base + 4*(q&3)
  int kq[2];
};
struct ColLoad {                 // "col" operand: element (k, c) at P[k*ld + c]
  const float* p[2];             // column base + krow*ld
  int krow[2];
  int cmask[2];                  // bit e set: column c+e is inside the matrix
};

template <int ROWS>
__device__ __forceinline__ RowLoad make_row_load(const float* P, int64_t ld, int64_t R, int64_t r0,
                                                 int tid) {
  RowLoad d;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    if (q >= ROWS * 4) q = 0;                     // idle slot of a narrow tile: harmless duplicate
    int64_t r = r0 + (q >> 2);
    if (r > R - 1) r = R - 1;
    d.kq[i] = (q & 3) << 2;
    d.p[i] = P + r * ld + d.kq[i];
  }
  return d;
}
template <int ROWS>
__device__ __forceinline__ ColLoad make_col_load(const float* P, int64_t ld, int64_t C, int64_t c0,
                                                 int tid) {
  constexpr int CQ = ROWS / 4;                    // float4 per k row
  ColLoad d;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    if (q >= CQ * BK) q = 0;
    const int64_t c = c0 + ((q % CQ) << 2);
    int m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= (c + e < C) ? (1 << e) : 0;
    d.cmask[i] = m;
    d.krow[i] = q / CQ;
    // keep the 16-byte load inside the operand's OWN columns: P may be a column-offset view of a wider
    // buffer ((hout, d*Hp), (gates, d*4H)), so `ld` says nothing about where the row ends -- a quad that
    // starts at or beyond C (fully masked) would read up to 4H floats past the row, i.e. past the END of the
    // buffer on its last row (found as a device fault that depended on where the allocator put the tensor)
    const int64_t cc = (c < C && c + 4 <= ld) ? c : 0;
    d.p[i] = P + cc + (int64_t)d.krow[i] * ld;
  }
  return d;
}
template <bool TAIL>
__device__ __forceinline__ void row_load(const RowLoad& d, int64_t k0, int64_t K, f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t k = k0 + d.kq[i];
    // a 16-byte load that starts at or beyond K would leave the row: read the row start instead
    const int64_t off = (!TAIL || k < K) ? k0 : -(int64_t)d.kq[i];
    f32x4 x = *reinterpret_cast<const f32x4*>(d.p[i] + off);
    if (TAIL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (k + e < K) ? x[e] : 0.f;
    }
    v[i] = x;
  }
}
// SHIFT: row k is read at k+kshift and is zero when (k % kperiod)+kshift leaves [0,kperiod)
template <bool TAIL, bool SHIFT>
__device__ __forceinline__ void col_load(const ColLoad& d, int64_t ld, int64_t k0, int64_t K,
                                         int64_t kshift, int64_t kperiod, f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t k = k0 + d.krow[i];
    bool ok = true;
    int64_t kk = k0;
    if (TAIL) {
      ok = k < K;
      if (!ok) kk = 0 - d.krow[i];            // row 0: valid address, value masked below
    }
    if (SHIFT) {
      const int64_t ph = (k % kperiod) + kshift;
      const bool in = ph >= 0 && ph < kperiod;
      if (ok && in) kk = k0 + kshift;
      ok = ok && in;
    }
    f32x4 x = *reinterpret_cast<const f32x4*>(d.p[i] + kk * ld);
    if (TAIL || SHIFT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ok ? x[e] : 0.f;
    }
    if (d.cmask[i] != 15) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = ((d.cmask[i] >> e) & 1) ? x[e] : 0.f;
    }
    v[i] = x;
  }
}
template <int ROWS>
__device__ __forceinline__ void store_row_tile(float* S, int tid, const f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    if (q < ROWS * 4) *reinterpret_cast<f32x4*>(S + (q >> 2) * ROW_PITCH + ((q & 3) << 2)) = v[i];
  }
}
template <int ROWS>
__device__ __forceinline__ void store_col_tile(float* S, int tid, const f32x4 (&v)[2]) {
  constexpr int CQ = ROWS / 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int q = tid + NTHREADS * i;
    if (q < CQ * BK) *reinterpret_cast<f32x4*>(S + (q / CQ) * COL_PITCH + ((q % CQ) << 2)) = v[i];
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

// NARROW = false: 128x128 tile (waves 2x2, 2x2 MFMA tiles each); true: 128x32 edge tile
// (waves 4x1, one MFMA tile each) used for the last N % 128 columns so that e.g. N = 513 does not
// pay a whole 128-wide tile for one column.
template <bool A_KMAJOR, bool B_KMAJOR, bool SHIFT, bool NARROW>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
    int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
    int64_t b_kshift, int64_t kperiod, const float* __restrict__ bias, int act, int accumulate,
    StoreMap sm, int splitk, int64_t c_split_stride, int64_t n_begin, TileMap tmap, int rows_epilogue) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][OP_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int TM = NARROW ? 1 : 2, TN = NARROW ? 1 : 2, BNT = NARROW ? 32 : BN;
  const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
  int mt, nt, zsplit;
  if (!tile_map_decode(tmap, blockIdx.x, mt, nt, zsplit)) return;
  const int64_t m0 = (int64_t)mt * BM, n0 = n_begin + (int64_t)nt * BNT;

  // split-K range (in K tiles)
  const int64_t ktiles = (K + BK - 1) / BK;
  const int64_t per = (ktiles + splitk - 1) / splitk;
  const int64_t kt_begin = (int64_t)zsplit * per;
  const int64_t kt_end = kt_begin + per < ktiles ? kt_begin + per : ktiles;
  const int64_t kt_full = K / BK;               // tiles [0, kt_full) need no K mask

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  RowLoad ra_d, rb_d;
  ColLoad ca_d, cb_d;
  if (!A_KMAJOR) ra_d = make_row_load<BM>(A, lda, M, m0, tid); else ca_d = make_col_load<BM>(A, lda, M, m0, tid);
  if (!B_KMAJOR) rb_d = make_row_load<BNT>(B, ldb, N, n0, tid); else cb_d = make_col_load<BNT>(B, ldb, N, n0, tid);

  f32x4 ra[2], rb[2];
  auto gload = [&](int64_t kt) {
    const int64_t k0 = kt * BK;
    if (kt < kt_full) {
      if (!A_KMAJOR) row_load<false>(ra_d, k0, K, ra);
      else col_load<false, false>(ca_d, lda, k0, K, 0, 1, ra);
      if (!B_KMAJOR) row_load<false>(rb_d, k0, K, rb);
      else col_load<false, SHIFT>(cb_d, ldb, k0, K, b_kshift, kperiod, rb);
    } else {
      if (!A_KMAJOR) row_load<true>(ra_d, k0, K, ra);
      else col_load<true, false>(ca_d, lda, k0, K, 0, 1, ra);
      if (!B_KMAJOR) row_load<true>(rb_d, k0, K, rb);
      else col_load<true, SHIFT>(cb_d, ldb, k0, K, b_kshift, kperiod, rb);
    }
  };
  auto sstore = [&](int buf) {
    if (!A_KMAJOR) store_row_tile<BM>(lds[buf][0], tid, ra); else store_col_tile<BM>(lds[buf][0], tid, ra);
    if (!B_KMAJOR) store_row_tile<BNT>(lds[buf][1], tid, rb); else store_col_tile<BNT>(lds[buf][1], tid, rb);
  };

  if (kt_begin < kt_end) {
    gload(kt_begin);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
      const bool more = kt + 1 < kt_end;
      if (more) gload(kt + 1);
      float fa[TM][8], fb[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<A_KMAJOR>(lds[buf][0], (wm * TM + i) * 32, lane, fa[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<B_KMAJOR>(lds[buf][1], (wn * TN + j) * 32, lane, fb[j]);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
      if (more) sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  float* Cz = C + (int64_t)zsplit * c_split_stride;
  if constexpr (!NARROW) {
    // the row-transposed store of the split-bf16 kernels (16 bytes per lane instead of 4: gemm_common.h), two waves at
    // a time through the operand stages (4 x 64 x 68 floats do not fit the 40 KB of LDS at once); TSSEP_GEMM_F32_ROWS=0
    // keeps the direct store
    static_assert(2 * 64 * EPITCH <= 4 * OP_FLOATS, "two epilogue scratches must fit in the operand stages");
    if (rows_epilogue) {
      float* stage = &lds[0][0][0] + (wave & 1) * 64 * EPITCH;
      const int64_t mr = m0 + (int64_t)wm * 64, nc = n0 + (int64_t)wn * 64;
      const float* eb = splitk == 1 ? bias : nullptr;
      const int ea = splitk == 1 ? act : 0;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if ((wave >> 1) == pass) {
          if (!sm.remap) gemm_epilogue_rows(acc, stage, Cz, M, N, mr, nc, lane, eb, ea, accumulate, sm.ldc, splitk == 1, sm.aux, sm.ldaux);
          else if (remap_vec_ok(sm, Cz)) gemm_epilogue_rows_remap_vec(acc, stage, Cz, M, N, mr, nc, lane, eb, ea, accumulate, sm);
          else if (remap_wide_ok(sm)) gemm_epilogue_rows_remap_wide(acc, stage, Cz, M, N, mr, nc, lane, eb, ea, accumulate, sm);
          else gemm_epilogue_rows_remap(acc, stage, Cz, M, N, mr, nc, lane, eb, ea, accumulate, sm);
        }
        if (pass == 0) __syncthreads();
      }
      return;
    }
  }
  gemm_epilogue<TM, TN>(acc, Cz, M, N, m0 + (int64_t)wm * TM * 32, n0 + (int64_t)wn * TN * 32, lane,
                        bias, act, accumulate, sm, splitk == 1);
}

}  // namespace

// One entry for launch, forced launch and plan query (gemm_dispatch.h): `call` says which.
static int gemm_dispatch(const tssep_gemm_args* g, GemmCall& call) {
  if (!g || !g->A || !g->B || (!g->C && !call.dry)) return TSSEP_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return TSSEP_E_SHAPE;
  if (!aligned16(g->A) || !aligned16(g->B) || (g->lda & 3) || (g->ldb & 3)) return TSSEP_E_ALIGN;
  const int splitk = g->splitk > 1 ? g->splitk : 1;
  if (splitk > 1 && (g->bias || g->act || g->c_remap)) return TSSEP_E_UNSUPPORTED;
  if (g->act < 0 || g->act > 2 || (g->act == 2 && !g->aux)) return TSSEP_E_SHAPE;
  if (g->a_kmajor && !g->b_kmajor) return TSSEP_E_UNSUPPORTED;
  if (g->kperiod > 0 && !g->b_kmajor) return TSSEP_E_UNSUPPORTED;
  const GemmSwitches sw = gemm_switches();
  StoreMap sm = make_store_map(g);
  if (sm.remap && !sw.remap_wide) sm.remap = 2;      // (experiment build: the 4-byte-per-lane remapped store)
  if (g->precision >= 1 && g->precision <= 3) return tssep_gemm_bf16x3_launch(g, sm, splitk, call);
  if (g->precision != 0 || g->b_ones_col) return TSSEP_E_UNSUPPORTED;
  if (call.force != TSSEP_GEMM_AUTO && call.force != TSSEP_GEMM_F32) return TSSEP_E_UNSUPPORTED;
  call.chosen = TSSEP_GEMM_F32;
  if (call.dry) return TSSEP_OK;
  const unsigned mtiles = (unsigned)((g->M + BM - 1) / BM);
  hipStream_t s = (hipStream_t)call.stream;
  // columns [0, n_main) by 128-wide tiles; a remainder of <= 96 columns by 32-wide edge tiles
  int64_t n_main = (g->N / BN) * BN;
  const int64_t rem = g->N - n_main;
  // a full-width tile is cheaper than 4 narrow ones; split-K launches are short and
  // under-filled already, a second serialized launch would only add a tail
  if (rem > 96 || splitk > 1) n_main = g->N;
  const bool shift = g->kperiod > 0;
  const int rows_epi = sw.f32_rows ? 1 : 0;          // (0, experiment build: the direct 4-byte-per-lane store)
#define LAUNCH(AK, BKM, SH, NARROW, GRIDX, NBEGIN)                                                 \
  do {                                                                                             \
    const TileMap tm_ = make_tile_map(mtiles, (GRIDX), splitk);                                    \
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, SH, NARROW>),                                     \
                       dim3((unsigned)tile_map_blocks(tm_)), dim3(NTHREADS), 0, s, g->A, g->B,     \
                       g->C, g->M, g->N, g->K, g->lda, g->ldb, g->b_kshift, g->kperiod, g->bias,   \
                       g->act, g->accumulate, sm, splitk, g->c_split_stride, (int64_t)(NBEGIN),    \
                       tm_, rows_epi);                                                             \
  } while (0)
#define DISPATCH(NARROW, GRIDX, NBEGIN)                                                            \
  do {                                                                                             \
    if (!g->a_kmajor && !g->b_kmajor) LAUNCH(false, false, false, NARROW, GRIDX, NBEGIN);          \
    else if (!g->a_kmajor && shift) LAUNCH(false, true, true, NARROW, GRIDX, NBEGIN);              \
    else if (!g->a_kmajor) LAUNCH(false, true, false, NARROW, GRIDX, NBEGIN);                      \
    else if (shift) LAUNCH(true, true, true, NARROW, GRIDX, NBEGIN);                               \
    else LAUNCH(true, true, false, NARROW, GRIDX, NBEGIN);                                         \
  } while (0)
  if (n_main > 0) DISPATCH(false, (unsigned)((n_main + BN - 1) / BN), 0);
  if (n_main < g->N) DISPATCH(true, (unsigned)((g->N - n_main + 31) / 32), n_main);
#undef DISPATCH
#undef LAUNCH
  return tssep_launch_status();
}

// `accumulate` is a boolean in the ABI (include/tssep_hip.h): any non-zero value means C += result in EVERY epilogue
// (the value 2 is a timing probe of the experiment build's kernels only)
static tssep_gemm_args normalised(const tssep_gemm_args* g) {
  tssep_gemm_args a = *g;
  a.accumulate = a.accumulate ? 1 : 0;
  return a;
}

extern "C" int tssep_gemm_f32(const tssep_gemm_args* g, void* stream) {
  if (!g) return TSSEP_E_NULL;
  const tssep_gemm_args a = normalised(g);
  GemmCall call{stream, false, TSSEP_GEMM_AUTO, TSSEP_GEMM_AUTO};
  return gemm_dispatch(&a, call);
}

extern "C" int tssep_gemm_f32_on(const tssep_gemm_args* g, int32_t kernel, void* stream) {
  if (!g) return TSSEP_E_NULL;
  if (kernel < TSSEP_GEMM_AUTO || kernel > TSSEP_GEMM_KERNEL_LAST) return TSSEP_E_SHAPE;
  const tssep_gemm_args a = normalised(g);
  GemmCall call{stream, false, kernel, TSSEP_GEMM_AUTO};
  return gemm_dispatch(&a, call);
}

extern "C" int tssep_gemm_plan(const tssep_gemm_args* g, int32_t force, int32_t* kernel) {
  if (!g || !kernel) return TSSEP_E_NULL;
  if (force < TSSEP_GEMM_AUTO || force > TSSEP_GEMM_KERNEL_LAST) return TSSEP_E_SHAPE;
  const tssep_gemm_args a = normalised(g);
  GemmCall call{nullptr, true, force, TSSEP_GEMM_AUTO};
  const int rc = gemm_dispatch(&a, call);
  *kernel = rc == TSSEP_OK ? call.chosen : TSSEP_GEMM_AUTO;
  return rc;
}

extern "C" const char* tssep_gemm_kernel_name(int32_t kernel) {
  static const char* const names[] = {"auto", "f32", "pipe", "tall2", "tall4", "tall4_xcol", "big", "stream", "nt_w160",
                                      "tn", "tn_tall", "tn_big", "tn_w160", "tn_h160", "big_p", "big_p320", "tn_p320"};
  return kernel >= 0 && kernel <= TSSEP_GEMM_KERNEL_LAST ? names[kernel] : "?";
}

// Split count of a weight gradient dW[M, N] = dY[K, M]^T X[K, N] (both operands k-major): `g` as for the launch,
// splitk / c_split_stride / C ignored.  The rule follows the KERNEL the dispatcher picks (asked through the plan
// query, never re-derived): tools/sweep_splitk.py, profiles/r2_splitk_sweep.jsonl, r3_wgrad_*_sweep.jsonl -- a model
// that prices whole rounds of 512 resident workgroups predicts up to 20 % from other factors; measured, the large-K
// shapes get SLOWER with more splits (the tiles of a K slab share it through one L2 only while they run together),
// and these rules are within 0..5 % of the best S on every shape of the step.
static int wgrad_split_rule(const tssep_gemm_args* g, int32_t kid);
extern "C" int tssep_gemm_wgrad_splits(const tssep_gemm_args* g) {
  if (!g) return TSSEP_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return TSSEP_E_SHAPE;
  tssep_gemm_args a = normalised(g);
  a.splitk = 8; a.c_split_stride = 0;
  auto planned = [&](int splitk, int32_t* kid) {
    a.splitk = splitk;
    GemmCall call{nullptr, true, TSSEP_GEMM_AUTO, TSSEP_GEMM_AUTO};
    const int rc = gemm_dispatch(&a, call);
    *kid = call.chosen;
    return rc;
  };
  int32_t kid = TSSEP_GEMM_AUTO;
  {
    const int rc = planned(8, &kid);
    if (rc != TSSEP_OK) return rc;
  }
  // The split rule is written for the kernel the dispatcher picks at splitk = 8; a launcher may refuse the S that rule
  // gives (a non-multiple of 8, the `S * 8 <= ktiles` limit of tn_p320) and the launch then falls through to ANOTHER
  // kernel than the one the rule was written for.  So: plan again with the S just chosen and, when the kernel
  // changed, apply THAT kernel's rule -- until the (kernel, S) pair is a fixed point (two rounds at most in practice).
  int S = wgrad_split_rule(g, kid);
  bool fixed = false;
  for (int round = 0; round < 4 && !fixed; ++round) {
    int32_t kid2 = TSSEP_GEMM_AUTO;
    if (planned(S, &kid2) != TSSEP_OK) break;
    if (kid2 == kid) { fixed = true; break; }
    kid = kid2;
    S = wgrad_split_rule(g, kid);
  }
  // (no fixed point -- kernels whose rules keep handing the request to each other, or a refused plan: 8 splits, which
  // every weight-gradient kernel takes)
  return fixed ? S : 8;
}

// The split count the rule of kernel `kernel_id` gives this request (tests: the kernel planned AT the returned S of
// tssep_gemm_wgrad_splits must be the kernel whose rule produced that S).
extern "C" int tssep_gemm_wgrad_split_rule(const tssep_gemm_args* g, int32_t kernel_id) {
  if (!g) return TSSEP_E_NULL;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return TSSEP_E_SHAPE;
  return wgrad_split_rule(g, kernel_id);
}

static int wgrad_split_rule(const tssep_gemm_args* g, int32_t kid) {
  auto cdiv = [](int64_t x, int64_t y) { return (x + y - 1) / y; };
  auto rup = [&](int64_t x, int64_t y) { return cdiv(x, y) * y; };
  const int64_t M = g->M, N = g->N, K = g->K, ktiles = cdiv(K, 16);
  const int min_ktiles = 8;
  if (kid == TSSEP_GEMM_TN_BIG && K >= 16 * 64) {
    // 512 x 128 tiles, ONE workgroup per CU, the tiles of a K slab on one XCD (32 CUs) -> as many slabs per XCD as fill
    // its CUs best; multiples of 8 only (N = 128 q + 1 | 2: the last columns ride on the VALU of the q-th column tile)
    const GemmSwitches sw = gemm_switches();
    const int ones = g->b_ones_col ? 1 : 0;
    const int64_t rem = N % 128;
    const bool xc = sw.tn_xc && N > 128 && rem >= 1 && rem <= 2 && rem - ones <= 1;
    const int64_t tiles = cdiv(M, 512) * (xc ? N / 128 : cdiv(N, 128));
    int best = 8; double waste = 1e30;
    for (int S = 8; S <= 32; S += 8) {
      const int64_t wg = tiles * (S / 8);
      const double w = (double)rup(wg, 32) / (double)wg;
      if (w < waste) { waste = w; best = S; }
    }
    return best;
  }
  if (kid == TSSEP_GEMM_TN_P320 && K >= 16 * 64) {
    // 192 x 320 tiles, ONE workgroup per CU, the tiles of a K slab on one XCD (32 CUs): the multiple of 8 that fills whole
    // rounds of 32 best (13 tiles: 56 splits = 91 workgroups per XCD in 3 rounds)
    const int64_t tiles = cdiv(M, 192) * cdiv(N - (g->b_ones_col ? 1 : 0), 320);
    int best = 8; double waste = 1e30;
    for (int S = 8; S <= 64 && (int64_t)S * 8 <= ktiles; S += 8) {
      const int64_t wg = tiles * (S / 8);
      const double w = (double)rup(wg, 32) / (double)wg;
      if (w < waste - 1e-9) { waste = w; best = S; }
    }
    return best;
  }
  if (kid == TSSEP_GEMM_TN_W160 && ktiles >= 64 * 8) {
    if (const int wide = tn_w160_wide(g)) {
      if (wide == 7) {
        // swapped operands: the tiles of the transposed problem (three for 320 x 600 + 1), up to 96 splits as on the 320 x 128
        // tile before (0.92 ms at S = 80, 1.06 at 64, 1.16 at 48; 777 216 rows)
        const int64_t tiles = cdiv(N, 256) * cdiv(M, 320);
        auto fill = [&](int S) { const int64_t wg = tiles * (S / 8); return (double)rup(wg, 32) / (double)wg; };
        double waste = 1e30;
        for (int S = 8; S <= 96 && (int64_t)S * 64 <= ktiles; S += 8) waste = fill(S) < waste ? fill(S) : waste;
        for (int S = 8; S <= 96 && (int64_t)S * 64 <= ktiles; S += 8)
          if (fill(S) <= waste * 1.07) return S;
        return 8;
      }
      // 256 x 320 tiles, ONE workgroup per CU, the tiles of a K slab on one XCD (32 CUs): the multiple of 8 that fills whole
      // rounds of 32 best, the smallest one among equals, >= 64 K tiles per split (tools/exp_wgrad_w320.py: dW_ih of birnn1,
      // 10 tiles: 3.14 ms at S = 24, 3.25 at 48, 3.75 at 16; birnn2, 40 tiles: 3.00 at 32, 3.09 at 24, 3.43 at 48; dW_hh,
      // 5 tiles: 48; tools/exp_wgrad_w320.py --sweep)
      const int64_t tiles = (rup(M, 256) / 256) * cdiv(tn_w160_wide_cols(g, wide), wide == 5 ? 320 : 256);      // (extra columns ride on the VALU)
      // (the smallest S within 7 % of the best fill: dW_ih of birnn0, 20 tiles, 5.54 ms at S = 24 (60 per XCD), 5.69-5.79 at 64
      // (160 = five full rounds) -- and a third of the partial sums to reduce)
      auto fill = [&](int S) { const int64_t wg = tiles * (S / 8); return (double)rup(wg, 32) / (double)wg; };
      double waste = 1e30;
      for (int S = 8; S <= 64 && (int64_t)S * 64 <= ktiles; S += 8) waste = fill(S) < waste ? fill(S) : waste;
      for (int S = 8; S <= 64 && (int64_t)S * 64 <= ktiles; S += 8)
        if (fill(S) <= waste * 1.07) return S;
      return 8;
    }
    // 256 x 160 tile, two workgroups per CU, one round of at most 512 (tools/sweep_wgrad_splits.py: dW_hh 2.00 ms at
    // S = 48, 2.30 at 40, 3.13 at 56)
    const int64_t tiles = (rup(M, 256) / 256) * (rup(N, 160) / 160);
    const int64_t a1 = 512 / tiles / 8 * 8, a2 = ktiles / 64 / 8 * 8;
    const int64_t v = a1 < a2 ? a1 : a2;
    return (int)(v > 8 ? v : 8);
  }
  if (kid == TSSEP_GEMM_TN_H160 && ktiles >= 64 * 8) {
    // 320 x 128 tile, two workgroups per CU, one round of at most 512, at most 96 splits
    const int64_t tiles = (rup(M, 320) / 320) * cdiv(N, 128);
    int64_t v = 512 / tiles / 8 * 8;
    if (v > 96) v = 96;
    if (v > ktiles / 16 / 8 * 8) v = ktiles / 16 / 8 * 8;
    return (int)(v > 8 ? v : 8);
  }
  const int64_t tiles = cdiv(M, 128) * cdiv(N, 128);
  // the two largest dW_ih GEMMs of the step (K = 777 216 rows, 95 / 57 tiles): the sweep's best S is the smallest one --
  // 8.37 vs 8.65 ms and 5.16 vs 5.24 ms standalone, -0.5 ms per step in an alternating A/B x3
  if (K >= 400000 && tiles >= 48) return 8;
  const int64_t smax = ktiles / min_ktiles;          // every split keeps >= min_ktiles K tiles
  if (tiles <= 16 && ktiles >= 64 * 8) {
    // few tiles (a projection weight gradient off the 320-row tile): one resident round of 512 workgroups -- 32 splits
    // 1.38 ms, the 56 of the general rule 1.51 (tools/sweep_wgrad_small.py); never more splits than K tiles allow
    int64_t v = 512 / tiles / 8 * 8;
    if (v > smax / 8 * 8) v = smax / 8 * 8;
    return (int)(v > 8 ? v : 8);
  }
  int64_t sp = cdiv(768, tiles);
  if (sp > smax) sp = smax;
  if (sp < 1) sp = 1;
  if (sp > 1) {        // multiples of the XCD count: split z runs on XCD z % 8 (gemm_common.h)
    const int64_t cap = smax / 8 * 8 > 8 ? smax / 8 * 8 : 8;
    sp = rup(sp, 8) < cap ? rup(sp, 8) : cap;
  }
  return (int)(sp < 64 ? sp : 64);
}
