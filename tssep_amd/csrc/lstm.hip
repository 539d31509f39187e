// Bidirectional LSTM recurrence (forward and backward-through-time) for gfx950.
//
// Replaces the T-sequential part of torch.nn.LSTM as the reference uses it
// (tssep/train/rnnp.py:88-95,146-153: one bidirectional layer, batch_first, zero initial
// state).  The input projection x W_ih^T + b is a plain GEMM (gemm.hip) done beforehand.
//
// Design (MI355X): sequences are independent, so the batch is split into groups of 8
// sequences x 1 direction per workgroup and NO inter-workgroup synchronisation is needed.
// W_hh (1.44 MB fp32 for H = 300) does not fit a CU, so every step streams it from L2
// straight into VGPRs (GEMV-style: 16 B/lane loads, 3-stage software ring that keeps running
// across time steps) while the previous step's h sits in LDS.  At 8 sequences the L2 stream
// (~56-64 B/clk/CU) and the fp32 matrix rate (256 FLOP/clk/CU) are balanced.
//
// Matrix instruction: v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 outer products per
// issue.  Block = hidden unit, row = gate (i,f,g,o), column = sequence: after the k loop
// every lane owns all four gate pre-activations of ONE (unit, sequence) cell, so the cell
// update is lane-local (no shuffles, no LDS).  It is exact fp32 (fmaf chain).
//
// Packed layouts (tssep_lstm_pack):
//   gate columns: [dir][unit][gate]            (4H per direction)
//   whh_f : [dir][wave 4][kq KQ3][i NB][lane 64][4]   = W_hh[g*H + u][4kq + e],
//           u = (wave*NB + i)*16 + lane/4, g = lane%4          (forward recurrence, A operand)
//   whh_b : [dir][gate 4][kq KQ3][ubb NBB][lane 64][4] = W_hh[g*H + 4kq + e][ubb*64 + lane]
//                                                      (BPTT: dh_prev = dgates x W_hh)
// with NB = ceil(ceil(H/16)/4), NBB = ceil(H/64), KQ3 = ceil(ceil(H/4)/3)*3; zero padded.
#include "common.h"

namespace {

__host__ __device__ inline int lstm_nb(int H) { return ((H + 15) / 16 + 3) / 4; }
__host__ __device__ inline int lstm_nbb(int H) { return (H + 63) / 64; }
__host__ __device__ inline int lstm_kq3(int H) { return (((H + 3) / 4 + 2) / 3) * 3; }

__device__ __forceinline__ float tanhf_acc(float x) { return tanhf(x); }

// ------------------------------------------------------------------------- pack
__global__ void lstm_pack_kernel(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f,
                                 const float* b_hh_f, const float* w_ih_r, const float* w_hh_r,
                                 const float* b_ih_r, const float* b_hh_r, int H, int I,
                                 int64_t ld_i, float* wih_p, float* bias_p, float* whh_f,
                                 float* whh_b) {
  const int NB = lstm_nb(H), NBB = lstm_nbb(H), KQ3 = lstm_kq3(H);
  const int64_t n_wih = (int64_t)8 * H * ld_i;
  const int64_t n_f = (int64_t)2 * 4 * KQ3 * NB * 256;
  const int64_t n_b = (int64_t)2 * 4 * KQ3 * NBB * 256;
  const int64_t total = n_wih + 8 * H + n_f + n_b;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (e < n_wih) {
      const int64_t row = e / ld_i, k = e - row * ld_i;
      const int d = (int)(row / (4 * H)), ug = (int)(row % (4 * H));
      const int u = ug >> 2, g = ug & 3;
      const float* w = d ? w_ih_r : w_ih_f;
      wih_p[e] = k < I ? w[(int64_t)(g * H + u) * I + k] : 0.f;
    } else if (e < n_wih + 8 * H) {
      const int row = (int)(e - n_wih);
      const int d = row / (4 * H), ug = row % (4 * H);
      const int u = ug >> 2, g = ug & 3;
      bias_p[row] = (d ? b_ih_r : b_ih_f)[g * H + u] + (d ? b_hh_r : b_hh_f)[g * H + u];
    } else if (e < n_wih + 8 * H + n_f) {
      int64_t r = e - n_wih - 8 * H;
      const int el = (int)(r & 3); r >>= 2;
      const int lane = (int)(r & 63); r >>= 6;
      const int i = (int)(r % NB); r /= NB;
      const int kq = (int)(r % KQ3); r /= KQ3;
      const int wave = (int)(r & 3);
      const int d = (int)(r >> 2);
      const int u = (wave * NB + i) * 16 + (lane >> 2), g = lane & 3, k = 4 * kq + el;
      const float* w = d ? w_hh_r : w_hh_f;
      whh_f[e - n_wih - 8 * H] = (u < H && k < H) ? w[(int64_t)(g * H + u) * H + k] : 0.f;
    } else {
      int64_t r = e - n_wih - 8 * H - n_f;
      const int el = (int)(r & 3); r >>= 2;
      const int lane = (int)(r & 63); r >>= 6;
      const int ubb = (int)(r % NBB); r /= NBB;
      const int kq = (int)(r % KQ3); r /= KQ3;
      const int g = (int)(r & 3);
      const int d = (int)(r >> 2);
      const int unit = ubb * 64 + lane, kk = 4 * kq + el;
      const float* w = d ? w_hh_r : w_hh_f;
      whh_b[e - n_wih - 8 * H - n_f] =
          (unit < H && kk < H) ? w[(int64_t)(g * H + kk) * H + unit] : 0.f;
    }
  }
}

// dst_d[(g*H + u)*ncols + k] = sum_s src[s*split_stride + (d*4H + 4u + g)*ld + k]
__global__ void lstm_unpack_kernel(const float* __restrict__ src, int64_t ld, int nsplit,
                                   int64_t split_stride, int H, int ncols, float* dst_f,
                                   float* dst_r, int accumulate) {
  const int64_t per = (int64_t)4 * H * ncols, total = 2 * per;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int d = e >= per;
    const int64_t r = e - d * per;
    const int64_t row = r / ncols;
    const int k = (int)(r - row * ncols);
    const int g = (int)(row / H), u = (int)(row % H);
    const float* p = src + ((int64_t)d * 4 * H + 4 * u + g) * ld + k;
    float s = 0.f;
    for (int i = 0; i < nsplit; ++i) s += p[i * split_stride];
    float* o = (d ? dst_r : dst_f) + r;
    *o = accumulate ? *o + s : s;
  }
}

// --------------------------------------------------------------------- forward
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int NB>
__global__ __launch_bounds__(256, 1) void blstm_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ cell, float* __restrict__ hout, int64_t ldo,
    int64_t dstride, const float* __restrict__ whh_f, int64_t N, int64_t T, int H, int KQ3) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KROW = 4 * KQ3 + 12;
  float* hs = smem;  // [2][8][KROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.y;
  const int64_t seq0 = (int64_t)blockIdx.x * 8;
  const int ub = lane >> 2, j = lane & 3;
  for (int i = tid; i < 2 * 8 * KROW; i += 256) hs[i] = 0.f;

  const f32x4* W = reinterpret_cast<const f32x4*>(whh_f) +
                   ((int64_t)(dir * 4 + wave) * KQ3) * NB * 64 + lane;
  int64_t nrow[2];
  bool nvalid[2];
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    const int64_t n = seq0 + sg * 4 + j;
    nvalid[sg] = n < N;
    nrow[sg] = nvalid[sg] ? n : 0;
  }
  int u[NB];
  bool uvalid[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    u[i] = (wave * NB + i) * 16 + ub;
    uvalid[i] = u[i] < H;
  }
  float c[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i) { c[i][0] = 0.f; c[i][1] = 0.f; }

  f32x4 w0[NB], w1[NB], w2[NB];
#define LOADW(dst, kq_)                                              \
  {                                                                  \
    const f32x4* p_ = W + (int64_t)(kq_) * NB * 64;                  \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) dst[i] = p_[i * 64]; \
  }
#define LOADH(ha_, hb_, kq_)                                            \
  {                                                                     \
    ha_ = *reinterpret_cast<const f32x4*>(hrow0 + 4 * (kq_));           \
    hb_ = *reinterpret_cast<const f32x4*>(hrow1 + 4 * (kq_));           \
  }
#define MM(src, ha_, hb_)                                               \
  {                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                     \
      _Pragma("unroll") for (int i = 0; i < NB; ++i) {                  \
        acc[i][0] = MFMA4(src[i][e], ha_[e], acc[i][0]);                \
        acc[i][1] = MFMA4(src[i][e], hb_[e], acc[i][1]);                \
      }                                                                 \
    }                                                                   \
  }
#define SB __builtin_amdgcn_sched_barrier(0)
  LOADW(w0, 0);
  LOADW(w1, 1);
  __syncthreads();

  int cur = 0;
  for (int64_t step = 0; step < T; ++step) {
    const int64_t t = dir ? T - 1 - step : step;
    f32x4 gx[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        // streamed once: non-temporal so these lines do not push W_hh out of the XCD's L2
        if (nvalid[sg] && uvalid[i])
          v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(
              gates + (((nrow[sg] * T + t) * 2 + dir) * (int64_t)H + u[i]) * 4));
        gx[i][sg] = v;
      }
    f32x4 acc[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* hrow0 = hs + (cur * 8 + j) * KROW;
    const float* hrow1 = hs + (cur * 8 + 4 + j) * KROW;
    f32x4 ha0, hb0, ha1, hb1, ha2, hb2;
    LOADH(ha0, hb0, 0);
    LOADH(ha1, hb1, 1);
    for (int kq = 0; kq < KQ3; kq += 3) {
      // 3-stage ring (W from L2, h from LDS), order pinned: the scheduler otherwise sinks the
      // loads to the loop tail and exposes their latency
      LOADW(w2, kq + 2); LOADH(ha2, hb2, kq + 2); SB;
      MM(w0, ha0, hb0); SB;
      LOADW(w0, (kq + 3 < KQ3 ? kq + 3 : 0)); LOADH(ha0, hb0, kq + 3); SB;
      MM(w1, ha1, hb1); SB;
      LOADW(w1, (kq + 4 < KQ3 ? kq + 4 : 1)); LOADH(ha1, hb1, kq + 4); SB;
      MM(w2, ha2, hb2); SB;
    }
    float* hnext = hs + ((cur ^ 1) * 8) * KROW;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        const f32x4 a = acc[i][sg] + gx[i][sg];
        const float ig = sigmoidf_acc(a[0]), fg = sigmoidf_acc(a[1]);
        const float gg = tanhf_acc(a[2]), og = sigmoidf_acc(a[3]);
        const float cn = fg * c[i][sg] + ig * gg;
        c[i][sg] = cn;
        const float h = og * tanhf_acc(cn);
        if (uvalid[i]) {
          hnext[(sg * 4 + j) * KROW + u[i]] = h;
          if (nvalid[sg]) {
            const int64_t cellidx = ((nrow[sg] * T + t) * 2 + dir) * (int64_t)H + u[i];
            __builtin_nontemporal_store(f32x4{ig, fg, gg, og},
                                        reinterpret_cast<f32x4*>(gates + cellidx * 4));
            __builtin_nontemporal_store(cn, cell + cellidx);
            __builtin_nontemporal_store(h, hout + (nrow[sg] * T + t) * ldo + dir * dstride + u[i]);
          }
        }
      }
    __syncthreads();
    cur ^= 1;
  }
#undef LOADW
}

// -------------------------------------------------------------------- backward
template <int NB, int NBB>
__global__ __launch_bounds__(256, 1) void blstm_bwd_kernel(
    float* __restrict__ gates, const float* __restrict__ cell, const float* __restrict__ dhout,
    int64_t ldo, int64_t dstride, const float* __restrict__ whh_b, int64_t N, int64_t T, int H, int KQ3) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KROW = 4 * KQ3 + 12;
  const int PROW = NBB * 64 + 4;
  float* dgs = smem;                     // [4 gates][8 seqs][KROW]
  float* part = smem + 4 * 8 * KROW;     // [4 waves][8 seqs][PROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.y;
  const int64_t seq0 = (int64_t)blockIdx.x * 8;
  const int ub = lane >> 2, j = lane & 3;
  for (int i = tid; i < 4 * 8 * KROW + 4 * 8 * PROW; i += 256) smem[i] = 0.f;

  const f32x4* W = reinterpret_cast<const f32x4*>(whh_b) +
                   ((int64_t)(dir * 4 + wave) * KQ3) * NBB * 64 + lane;
  int64_t nrow[2];
  bool nvalid[2];
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    const int64_t n = seq0 + sg * 4 + j;
    nvalid[sg] = n < N;
    nrow[sg] = nvalid[sg] ? n : 0;
  }
  int u[NB];
  bool uvalid[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    u[i] = (wave * NB + i) * 16 + ub;
    uvalid[i] = u[i] < H;
  }
  float dcc[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i) { dcc[i][0] = 0.f; dcc[i][1] = 0.f; }

  f32x4 w0[NBB], w1[NBB], w2[NBB];
#define LOADW(dst, kq_)                                                  \
  {                                                                      \
    const f32x4* p_ = W + (int64_t)(kq_) * NBB * 64;                     \
    _Pragma("unroll") for (int i = 0; i < NBB; ++i) dst[i] = p_[i * 64]; \
  }
#undef MM
#define MM(src, ha_, hb_)                                               \
  {                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                     \
      _Pragma("unroll") for (int i = 0; i < NBB; ++i) {                 \
        acc[i][0] = MFMA4(src[i][e], ha_[e], acc[i][0]);                \
        acc[i][1] = MFMA4(src[i][e], hb_[e], acc[i][1]);                \
      }                                                                 \
    }                                                                   \
  }
  LOADW(w0, 0);
  LOADW(w1, 1);
  __syncthreads();

  for (int64_t step = 0; step < T; ++step) {
    const int64_t t = dir ? step : T - 1 - step;
    const bool has_prev = step + 1 < T;          // a cell state before this step exists
    const int64_t tp = dir ? t + 1 : t - 1;      // its time index
    // ---- cell backward (lane-local) ----
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        f32x4 dg4 = {0.f, 0.f, 0.f, 0.f};
        if (uvalid[i]) {
          const int s = sg * 4 + j;
          float dh = part[(0 * 8 + s) * PROW + u[i]] + part[(1 * 8 + s) * PROW + u[i]] +
                     part[(2 * 8 + s) * PROW + u[i]] + part[(3 * 8 + s) * PROW + u[i]];
          if (nvalid[sg]) {
            const int64_t cellidx = ((nrow[sg] * T + t) * 2 + dir) * (int64_t)H + u[i];
            const f32x4 g4 =
                __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gates + cellidx * 4));
            const float ct = __builtin_nontemporal_load(cell + cellidx);
            const float cp = has_prev ? __builtin_nontemporal_load(
                                            cell + ((nrow[sg] * T + tp) * 2 + dir) * (int64_t)H + u[i])
                                      : 0.f;
            dh += __builtin_nontemporal_load(dhout + (nrow[sg] * T + t) * ldo + dir * dstride + u[i]);
            const float tc = tanhf_acc(ct);
            const float d_o = dh * tc;
            const float dc = dh * g4[3] * (1.f - tc * tc) + dcc[i][sg];
            dcc[i][sg] = dc * g4[1];
            dg4[0] = dc * g4[2] * g4[0] * (1.f - g4[0]);
            dg4[1] = dc * cp * g4[1] * (1.f - g4[1]);
            dg4[2] = dc * g4[0] * (1.f - g4[2] * g4[2]);
            dg4[3] = d_o * g4[3] * (1.f - g4[3]);
            __builtin_nontemporal_store(dg4, reinterpret_cast<f32x4*>(gates + cellidx * 4));
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) dgs[(g * 8 + s) * KROW + u[i]] = dg4[g];
        }
      }
    __syncthreads();
    // ---- dh_prev[unit, seq] = sum_{g,kk} W_hh[g*H+kk][unit] * dgates[seq][g][kk]; wave = gate
    f32x4 acc[NBB][2];
#pragma unroll
    for (int i = 0; i < NBB; ++i) {
      acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* hrow0 = dgs + (wave * 8 + j) * KROW;
    const float* hrow1 = dgs + (wave * 8 + 4 + j) * KROW;
    f32x4 ha0, hb0, ha1, hb1, ha2, hb2;
    LOADH(ha0, hb0, 0);
    LOADH(ha1, hb1, 1);
    for (int kq = 0; kq < KQ3; kq += 3) {
      // 3-stage ring (W from L2, h from LDS), order pinned: the scheduler otherwise sinks the
      // loads to the loop tail and exposes their latency
      LOADW(w2, kq + 2); LOADH(ha2, hb2, kq + 2); SB;
      MM(w0, ha0, hb0); SB;
      LOADW(w0, (kq + 3 < KQ3 ? kq + 3 : 0)); LOADH(ha0, hb0, kq + 3); SB;
      MM(w1, ha1, hb1); SB;
      LOADW(w1, (kq + 4 < KQ3 ? kq + 4 : 1)); LOADH(ha1, hb1, kq + 4); SB;
      MM(w2, ha2, hb2); SB;
    }
#pragma unroll
    for (int i = 0; i < NBB; ++i)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
        *reinterpret_cast<f32x4*>(part + (wave * 8 + sg * 4 + j) * PROW + i * 64 + 4 * ub) =
            acc[i][sg];
    __syncthreads();
  }
#undef LOADW
#undef LOADH
#undef MM
#undef SB
}

}  // namespace

extern "C" int tssep_lstm_pack_sizes(int H, int I, int64_t ld_i, tssep_lstm_sizes* out) {
  if (!out) return TSSEP_E_NULL;
  if (H <= 0 || I <= 0 || ld_i < I || (ld_i & 3)) return TSSEP_E_SHAPE;
  if (lstm_nb(H) > 8) return TSSEP_E_UNSUPPORTED;      // H <= 512
  out->wih_p = (int64_t)8 * H * ld_i;
  out->bias_p = (int64_t)8 * H;
  out->whh_f = (int64_t)2 * 4 * lstm_kq3(H) * lstm_nb(H) * 256;
  out->whh_b = (int64_t)2 * 4 * lstm_kq3(H) * lstm_nbb(H) * 256;
  return TSSEP_OK;
}

extern "C" int tssep_lstm_pack(const float* w_ih_f, const float* w_hh_f, const float* b_ih_f,
                               const float* b_hh_f, const float* w_ih_r, const float* w_hh_r,
                               const float* b_ih_r, const float* b_hh_r, int H, int I,
                               int64_t ld_i, float* wih_p, float* bias_p, float* whh_f,
                               float* whh_b, void* stream) {
  if (!w_ih_f || !w_hh_f || !b_ih_f || !b_hh_f || !w_ih_r || !w_hh_r || !b_ih_r || !b_hh_r ||
      !wih_p || !bias_p || !whh_f || !whh_b)
    return TSSEP_E_NULL;
  tssep_lstm_sizes sz;
  if (int e = tssep_lstm_pack_sizes(H, I, ld_i, &sz)) return e;
  if (!aligned16(wih_p) || !aligned16(whh_f) || !aligned16(whh_b)) return TSSEP_E_ALIGN;
  const int64_t total = sz.wih_p + sz.bias_p + sz.whh_f + sz.whh_b;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, H, I, ld_i,
                     wih_p, bias_p, whh_f, whh_b);
  return tssep_launch_status();
}

extern "C" int tssep_lstm_unpack(const float* src, int64_t ld, int nsplit, int64_t split_stride,
                                 int H, int ncols, float* dst_f, float* dst_r, int accumulate,
                                 void* stream) {
  if (!src || !dst_f || !dst_r) return TSSEP_E_NULL;
  if (H <= 0 || ncols <= 0 || nsplit <= 0) return TSSEP_E_SHAPE;
  const int64_t total = (int64_t)8 * H * ncols;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     src, ld, nsplit, split_stride, H, ncols, dst_f, dst_r, accumulate);
  return tssep_launch_status();
}

extern "C" int tssep_blstm_fwd(float* gates, float* cell, float* hout, int64_t ldo,
                               int64_t dstride, const float* whh_f, int64_t N, int64_t T, int H,
                               void* stream) {
  if (!gates || !cell || !hout || !whh_f) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || H <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!aligned16(gates) || !aligned16(whh_f)) return TSSEP_E_ALIGN;
  const int NB = lstm_nb(H), KQ3 = lstm_kq3(H);
  if (NB > 8) return TSSEP_E_UNSUPPORTED;
  dim3 grid((unsigned)((N + 7) / 8), 2);
  const size_t lds = (size_t)2 * 8 * (4 * KQ3 + 12) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define L(NB_)                                                                                 \
  hipLaunchKernelGGL((blstm_fwd_kernel<NB_>), grid, dim3(256), lds, s, gates, cell, hout, ldo, \
                     dstride, whh_f, N, T, H, KQ3)
  switch (NB) {
    case 1: L(1); break;
    case 2: L(2); break;
    case 3: L(3); break;
    case 4: L(4); break;
    case 5: L(5); break;
    case 6: L(6); break;
    case 7: L(7); break;
    default: L(8); break;
  }
#undef L
  return tssep_launch_status();
}

extern "C" int tssep_blstm_bwd(float* gates, const float* cell, const float* dhout, int64_t ldo,
                               int64_t dstride, const float* whh_b, int64_t N, int64_t T, int H,
                               void* stream) {
  if (!gates || !cell || !dhout || !whh_b) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || H <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!aligned16(gates) || !aligned16(whh_b)) return TSSEP_E_ALIGN;
  const int NB = lstm_nb(H), NBB = lstm_nbb(H), KQ3 = lstm_kq3(H);
  if (NB > 8) return TSSEP_E_UNSUPPORTED;
  dim3 grid((unsigned)((N + 7) / 8), 2);
  const size_t lds = (size_t)(4 * 8 * (4 * KQ3 + 12) + 4 * 8 * (NBB * 64 + 4)) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define L(NB_, NBB_)                                                                          \
  hipLaunchKernelGGL((blstm_bwd_kernel<NB_, NBB_>), grid, dim3(256), lds, s, gates, cell,      \
                     dhout, ldo, dstride, whh_b, N, T, H, KQ3)
  // NB = ceil(ceil(H/16)/4), NBB = ceil(H/64): NBB is NB or NB-... enumerate the reachable pairs
  if (NB == 1) L(1, 1);
  else if (NB == 2) L(2, 2);
  else if (NB == 3) L(3, 3);
  else if (NB == 4) L(4, 4);
  else if (NB == 5) L(5, 5);
  else if (NB == 6) L(6, 6);
  else if (NB == 7) L(7, 7);
  else L(8, 8);
#undef L
  return tssep_launch_status();
}
