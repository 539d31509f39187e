// Cluster (W-stationary) BLSTM recurrence for gfx950 -- the latency-optimised forward/backward
// through time used when the batch is too small to fill the chip with the sequence-parallel
// kernels of lstm.hip.
//
// Replaces the T-sequential part of torch.nn.LSTM (tssep/train/rnnp.py:88-95,146-153).
//
// Why: lstm.hip streams all of W_hh (1.44 MB) from L2 into every CU on every time step, which
// costs ~18 us/step whatever the batch.  Here a CLUSTER of G = ceil(H/32) workgroups (one per
// CU, all co-resident) shares one group of M sequences x one direction; workgroup g owns hidden
// units [32g, 32g+32) and keeps ITS slice of W_hh in VGPRs for the whole launch (152 registers
// per lane), so a step is ~1 us of MFMA + one inter-CU exchange of h:
//
//   forward : all-gather of h_t        (each WG publishes 32 x M values, reads H x M)
//   backward: reduce-scatter of dh_t-1 (each WG publishes partial sums for all H units from its
//             128 gate columns, reads G partials for its own 32 units, adds them in fixed order)
//
// Exchange protocol (MI355X_MICROARCH.md "valid forms", recipe R2 -- the data is the flag):
// every value travels as ONE naturally aligned 8-byte granule {tag = step+1, value} written by a
// relaxed agent-scope atomic store (sc1, write-through) and read by relaxed agent-scope atomic
// loads; no fences, no separate flags, placement independent.  Two slots alternate by step
// parity; a slot is rewritten only after every WG of the cluster has consumed it (publishing
// step t+2 transitively requires everyone to have gathered step t).  All granule words are zeroed
// by the caller's hipMemsetAsync before EVERY launch.  Every spin is bounded: on timeout the WG
// raises err[0] and leaves, so a broken launch ends in milliseconds instead of hanging the GPU.
//
// Matrix instruction: v_mfma_f32_4x4x1_16b_f32 exactly as in lstm.hip (block = unit, row = gate,
// column = sequence; exact fp32).
#include "common.h"

namespace {

constexpr int KH = 152;          // k-half per wave (covers H <= 304)
constexpr int UPW = 32;          // hidden units per workgroup
constexpr int SPIN_LIMIT = 1 << 21;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ void granule_store(u64* p, unsigned tag, float v) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// Read R granules whose tag must equal `want`: all loads are issued back to back (one memory
// round trip when everything has already been published), only the stragglers are re-polled.
// Returns false on timeout.  p[r] == nullptr skips entry r (value 0).
template <int R>
__device__ __forceinline__ bool gather_granules(const u64* const (&p)[R], unsigned want,
                                                float (&v)[R]) {
  u64 x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = p[r] ? granule_load(p[r]) : ((u64)want << 32);
  int spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int r = 0; r < R; ++r) ok = ok && ((unsigned)(x[r] >> 32) == want);
    if (ok) break;
    if (++spins > SPIN_LIMIT) return false;
    __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((unsigned)(x[r] >> 32) != want) x[r] = granule_load(p[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = __uint_as_float((unsigned)x[r]);
  return true;
}

// packed weights for the cluster kernels (built by tssep_lstm_pack_cluster):
//   whh_cf [dir][g][wave 4][kk KH][lane 64]    = W_hh[gate*H + u][kh*KH + kk],
//          u = 32g + 16*(wave>>1) + lane/4, gate = lane%4, kh = wave&1
//   whh_cb [dir][g][wave 4][kk 32][b 5][lane 64] = W_hh[gc][64*b + lane],
//          gc = gate*H + (32g + 8*wave + kk/4) with gate = kk%4   (own gate columns, (unit,gate) order)
__global__ void lstm_pack_cluster_kernel(const float* w_hh_f, const float* w_hh_r, int H, int G,
                                         float* cf, float* cb) {
  const int64_t n_f = (int64_t)2 * G * 4 * KH * 64;
  const int64_t n_b = (int64_t)2 * G * 4 * 32 * 5 * 64;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_f + n_b;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (e < n_f) {
      int64_t r = e;
      const int lane = (int)(r & 63); r >>= 6;
      const int kk = (int)(r % KH); r /= KH;
      const int wave = (int)(r & 3); r >>= 2;
      const int g = (int)(r % G);
      const int d = (int)(r / G);
      const int u = 32 * g + 16 * (wave >> 1) + (lane >> 2), gate = lane & 3;
      const int k = (wave & 1) * KH + kk;
      const float* w = d ? w_hh_r : w_hh_f;
      cf[e] = (u < H && k < H) ? w[(int64_t)(gate * H + u) * H + k] : 0.f;
    } else {
      int64_t r = e - n_f;
      const int lane = (int)(r & 63); r >>= 6;
      const int b = (int)(r % 5); r /= 5;
      const int kk = (int)(r & 31); r >>= 5;
      const int wave = (int)(r & 3); r >>= 2;
      const int g = (int)(r % G);
      const int d = (int)(r / G);
      const int uo = 64 * b + lane;
      const int ui = 32 * g + 8 * wave + (kk >> 2), gate = kk & 3;
      const float* w = d ? w_hh_r : w_hh_f;
      cb[e - n_f] = (uo < H && ui < H) ? w[(int64_t)(gate * H + ui) * H + uo] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------- forward
// grid = nclusters * G workgroups of 256 threads; cluster c = blockIdx.x / G walks the work items
// (sequence group, direction) c, c + nclusters, ...   MS = sequences per group / 4 (even).
template <int MS>
__global__ __launch_bounds__(256, (MS == 2 ? 2 : 1)) void blstm_cluster_fwd_kernel(
    float* __restrict__ gates, float* __restrict__ cell, float* __restrict__ hout, int64_t ldo,
    int64_t dstride, const float* __restrict__ whh_cf, u64* __restrict__ xbuf,
    int* __restrict__ err, int64_t N, int64_t T, int H, int G, int nclusters) {
  constexpr int M = 4 * MS, HQ = MS / 2;
  constexpr int KROW = 2 * KH + 8;          // 312 floats: h row incl. zero pad
  const int Hp = G * UPW;
  __shared__ __attribute__((aligned(16))) float hs[M * KROW];
  __shared__ __attribute__((aligned(16))) float part[4 * HQ * 64 * 4];
  __shared__ int s_fail, s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ub = wave >> 1, kh = wave & 1;
  const unsigned tagbase = tssep_load_tagbase(err);
  // Cluster membership by arrival ticket: the first G workgroups that START form cluster 0, the
  // next G cluster 1, ...  A cluster therefore only ever waits for workgroups that are already
  // running or will be scheduled as soon as a complete (running) cluster retires -- no assumption
  // about dispatch order or co-residency of the whole grid.
  if (tid == 0) {
    s_fail = 0;
    s_ticket = (int)atomicAdd(reinterpret_cast<unsigned*>(xbuf), 1u);
  }
  __syncthreads();
  const int g = s_ticket % G, cid = s_ticket / G;
  xbuf += 8;                                   // skip the 64-byte header
  const int j = lane & 3;
  const int unit = UPW * g + 16 * ub + (lane >> 2);
  const bool uvalid = unit < H;

  const int64_t ngroups = (N + M - 1) / M;
  for (int64_t work = cid; work < 2 * ngroups; work += nclusters) {
    const int dir = (int)(work & 1);
    const int64_t sgid = work >> 1;
    const int64_t seq0 = sgid * M;
    // stationary weights -> registers
    float w[KH];
    {
      const float* wp = whh_cf + (((int64_t)(dir * G + g) * 4 + wave) * KH) * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) w[kk] = wp[(int64_t)kk * 64];
    }
    for (int i = tid; i < M * KROW; i += 256) hs[i] = 0.f;
    float c[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) c[q] = 0.f;
    u64* xb = xbuf + (int64_t)work * 2 * M * Hp;      // [slot 2][seq M][Hp]
    __syncthreads();

    for (int64_t step = 0; step < T; ++step) {
      const int64_t t = dir ? T - 1 - step : step;
      // pre-activations of the cells this wave finalises (prefetch; consumed after the k loop)
      f32x4 gx[HQ];
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const int64_t n = seq0 + 4 * (kh * HQ + q) + j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (uvalid && n < N)
          v = *reinterpret_cast<const f32x4*>(gates + (((n * T + t) * 2 + dir) * (int64_t)H + unit) * 4);
        gx[q] = v;
      }
      // ---- gather h_{t-1} (published with tag = step) into LDS: every thread issues all of its
      // granule loads at once (one memory round trip when the peers have already published)
      if (step > 0) {
        const u64* src = xb + (int64_t)((step - 1) & 1) * M * Hp;
        constexpr int CH = 5 * MS;               // M * 320 / 256 granules per thread
        const u64* p[CH];
        int off[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          const int i = r * 256 + tid;
          const int s = i / Hp, uu = i - s * Hp;
          const bool ok = i < M * Hp && uu < H;
          p[r] = ok ? src + i : nullptr;
          off[r] = ok ? s * KROW + uu : -1;
        }
        float v[CH];
        if (!gather_granules<CH>(p, tagbase | (unsigned)step, v)) s_fail = 1;
#pragma unroll
        for (int r = 0; r < CH; ++r)
          if (off[r] >= 0) hs[off[r]] = v[r];
      }
      __syncthreads();
      if (s_fail) {
        if (tid == 0) atomicExch(err, 1);
        return;
      }
      // ---- partial gate pre-activations over this wave's k half
      f32x4 acc[MS];
#pragma unroll
      for (int q = 0; q < MS; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* hb = hs + j * KROW + kh * KH;
#pragma unroll
      for (int k4 = 0; k4 < KH / 4; ++k4) {
#pragma unroll
        for (int q = 0; q < MS; ++q) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(hb + 4 * q * KROW + 4 * k4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q] = MFMA4(w[4 * k4 + e], hv[e], acc[q]);
        }
      }
      // ---- combine the two k halves: hand the partner the quads it finalises
#pragma unroll
      for (int q = 0; q < HQ; ++q)
        *reinterpret_cast<f32x4*>(part + ((wave * HQ + q) * 64 + lane) * 4) = acc[(1 - kh) * HQ + q];
      __syncthreads();
      u64* dst = xb + (int64_t)(step & 1) * M * Hp;
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const f32x4 other =
            *reinterpret_cast<const f32x4*>(part + (((wave ^ 1) * HQ + q) * 64 + lane) * 4);
        const f32x4 a = acc[kh * HQ + q] + other + gx[q];
        const float ig = sigmoidf_acc(a[0]), fg = sigmoidf_acc(a[1]);
        const float gg = tanhf(a[2]), og = sigmoidf_acc(a[3]);
        const float cn = fg * c[q] + ig * gg;
        c[q] = cn;
        const float h = og * tanhf(cn);
        const int s = 4 * (kh * HQ + q) + j;
        const int64_t n = seq0 + s;
        if (uvalid) {
          granule_store(dst + (int64_t)s * Hp + unit, tagbase | (unsigned)(step + 1), h);
          if (n < N) {
            const int64_t cellidx = ((n * T + t) * 2 + dir) * (int64_t)H + unit;
            *reinterpret_cast<f32x4*>(gates + cellidx * 4) = f32x4{ig, fg, gg, og};
            cell[cellidx] = cn;
            hout[(n * T + t) * ldo + dir * dstride + unit] = h;
          }
        }
      }
      // hs is rewritten by the next gather only after the barrier that follows it; part is
      // rewritten after the next k loop, i.e. after that barrier as well
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ backward
// 256 threads.  Cell backward is lane-local for this WG's 32 units x M sequences (same lane map
// as forward).  dh_prev = dgates x W_hh is a reduce-scatter: wave w multiplies ITS 32 gate columns
// (units 8w..8w+7 of the WG, 4 gates each; 160 stationary registers = 32 k x 5 output blocks of
// 64 units) into partial sums for ALL output units, the 4 waves are summed through LDS and the
// WG publishes one partial per (unit, sequence); the owner of a unit adds the G partials in a
// fixed order (deterministic).
template <int MS>
__global__ __launch_bounds__(256, (MS == 2 ? 2 : 1)) void blstm_cluster_bwd_kernel(
    float* __restrict__ gates, const float* __restrict__ cell, const float* __restrict__ dhout,
    int64_t ldo, int64_t dstride, const float* __restrict__ whh_cb, u64* __restrict__ xbuf,
    int* __restrict__ err, int64_t N, int64_t T, int H, int G, int nclusters) {
  constexpr int M = 4 * MS, HQ = MS / 2;
  constexpr int DROW = 128 + 4;              // d(gate) row per sequence: own 32 units x 4 gates
  constexpr int NBLK = 5;                    // output blocks of 64 units (H <= 320)
  const int Hp = G * UPW;
  __shared__ __attribute__((aligned(16))) float dgs[M * DROW];
  __shared__ __attribute__((aligned(16))) float part[4 * NBLK * MS * 64 * 4];
  __shared__ int s_fail, s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ub = wave >> 1, kh = wave & 1;
  const unsigned tagbase = tssep_load_tagbase(err);
  if (tid == 0) {
    s_fail = 0;
    s_ticket = (int)atomicAdd(reinterpret_cast<unsigned*>(xbuf), 1u);
  }
  __syncthreads();
  const int g = s_ticket % G, cid = s_ticket / G;
  xbuf += 8;
  const int j = lane & 3;
  const int unit = UPW * g + 16 * ub + (lane >> 2);
  const int ulocal = 16 * ub + (lane >> 2);
  const bool uvalid = unit < H;

  const int64_t ngroups = (N + M - 1) / M;
  for (int64_t work = cid; work < 2 * ngroups; work += nclusters) {
    const int dir = (int)(work & 1);
    const int64_t sgid = work >> 1;
    const int64_t seq0 = sgid * M;
    float w[32 * NBLK];
    {
      const float* wp = whh_cb + (((int64_t)(dir * G + g) * 4 + wave) * 32 * NBLK) * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < 32 * NBLK; ++kk) w[kk] = wp[(int64_t)kk * 64];
    }
    for (int i = tid; i < M * DROW; i += 256) dgs[i] = 0.f;
    float dcc[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) dcc[q] = 0.f;
    u64* xb = xbuf + (int64_t)work * 2 * G * M * Hp;      // [slot 2][src G][seq M][Hp]
    __syncthreads();

    for (int64_t step = 0; step < T; ++step) {
      const int64_t t = dir ? step : T - 1 - step;
      const bool has_prev = step + 1 < T;
      const int64_t tp = dir ? t + 1 : t - 1;
      {
        // (1) saved activations of this lane's cells: issued first, independent of the exchange
        f32x4 g4[HQ];
        float ct[HQ], cp[HQ], dh[HQ];
        int64_t cellidx[HQ];
        bool nv[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
          const int64_t n = seq0 + 4 * (kh * HQ + q) + j;
          nv[q] = uvalid && n < N;
          g4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          ct[q] = cp[q] = dh[q] = 0.f;
          cellidx[q] = 0;
          if (nv[q]) {
            cellidx[q] = ((n * T + t) * 2 + dir) * (int64_t)H + unit;
            g4[q] = *reinterpret_cast<const f32x4*>(gates + cellidx[q] * 4);
            ct[q] = cell[cellidx[q]];
            cp[q] = has_prev ? cell[((n * T + tp) * 2 + dir) * (int64_t)H + unit] : 0.f;
            dh[q] = dhout[(n * T + t) * ldo + dir * dstride + unit];
          }
        }
        // (2) reduce-scatter: the G partial dh of every cell, published with tag = step
        if (step > 0 && uvalid) {
          constexpr int GMAX = 10;
          const u64* p[GMAX * HQ];
#pragma unroll
          for (int q = 0; q < HQ; ++q) {
            const int s = 4 * (kh * HQ + q) + j;
            const u64* src = xb + (int64_t)((step - 1) & 1) * G * M * Hp + (int64_t)s * Hp + unit;
#pragma unroll
            for (int gs = 0; gs < GMAX; ++gs)
              p[q * GMAX + gs] = gs < G ? src + (int64_t)gs * M * Hp : nullptr;
          }
          float v[GMAX * HQ];
          if (!gather_granules<GMAX * HQ>(p, tagbase | (unsigned)step, v)) s_fail = 1;
#pragma unroll
          for (int q = 0; q < HQ; ++q)
#pragma unroll
            for (int gs = 0; gs < GMAX; ++gs) dh[q] += v[q * GMAX + gs];   // fixed order
        }
        // (3) cell backward
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
          const int s = 4 * (kh * HQ + q) + j;
          f32x4 dg4 = {0.f, 0.f, 0.f, 0.f};
          if (nv[q]) {
            const float tc = tanhf(ct[q]);
            const float d_o = dh[q] * tc;
            const float dc = dh[q] * g4[q][3] * (1.f - tc * tc) + dcc[q];
            dcc[q] = dc * g4[q][1];
            dg4[0] = dc * g4[q][2] * g4[q][0] * (1.f - g4[q][0]);
            dg4[1] = dc * cp[q] * g4[q][1] * (1.f - g4[q][1]);
            dg4[2] = dc * g4[q][0] * (1.f - g4[q][2] * g4[q][2]);
            dg4[3] = d_o * g4[q][3] * (1.f - g4[q][3]);
            *reinterpret_cast<f32x4*>(gates + cellidx[q] * 4) = dg4;
          }
          *reinterpret_cast<f32x4*>(dgs + s * DROW + 4 * ulocal) = dg4;
        }
      }
      __syncthreads();
      if (s_fail) {
        if (tid == 0) atomicExch(err, 2);
        return;
      }
      if (has_prev) {
        // ---- partial dh_prev[uo, seq] over this wave's 32 gate columns
        f32x4 acc[NBLK][MS];
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
          for (int q = 0; q < MS; ++q) acc[b][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* db = dgs + j * DROW + 32 * wave;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          f32x4 dv[MS];
#pragma unroll
          for (int q = 0; q < MS; ++q)
            dv[q] = *reinterpret_cast<const f32x4*>(db + 4 * q * DROW + 4 * k4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
              for (int q = 0; q < MS; ++q)
                acc[b][q] = MFMA4(w[(4 * k4 + e) * NBLK + b], dv[q][e], acc[b][q]);
        }
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
          for (int q = 0; q < MS; ++q)
            *reinterpret_cast<f32x4*>(part + (((wave * NBLK + b) * MS + q) * 64 + lane) * 4) = acc[b][q];
        __syncthreads();
        // ---- sum the 4 waves and publish: pair p = (b, q); D lane (blk = lane>>2, j), reg i ->
        //      output unit 64 b + 4 blk + i, sequence 4 q + j
        u64* dst = xb + ((int64_t)(step & 1) * G + g) * M * Hp;
        for (int pr = wave; pr < NBLK * MS; pr += 4) {
          const int b = pr / MS, q = pr - b * MS;
          f32x4 sum = *reinterpret_cast<const f32x4*>(part + (((0 * NBLK + b) * MS + q) * 64 + lane) * 4);
#pragma unroll
          for (int ww = 1; ww < 4; ++ww)
            sum += *reinterpret_cast<const f32x4*>(part + (((ww * NBLK + b) * MS + q) * 64 + lane) * 4);
          const int uo0 = 64 * b + 4 * (lane >> 2);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (uo0 + i < H)
              granule_store(dst + (int64_t)(4 * q + j) * Hp + uo0 + i, tagbase | (unsigned)(step + 1), sum[i]);
        }
      }
      __syncthreads();      // dgs / part are rewritten by the next step
    }
  }
}

}  // namespace

extern "C" int tssep_lstm_cluster_supported(int H) { return H > 0 && H <= 2 * KH && (H + UPW - 1) / UPW <= 10 ? 1 : 0; }

extern "C" int64_t tssep_lstm_cluster_pack_floats(int H, int which) {
  const int G = (H + UPW - 1) / UPW;
  return which == 0 ? (int64_t)2 * G * 4 * KH * 64 : (int64_t)2 * G * 4 * 32 * 5 * 64;
}

extern "C" int tssep_lstm_pack_cluster(const float* w_hh_f, const float* w_hh_r, int H,
                                       float* whh_cf, float* whh_cb, void* stream) {
  if (!w_hh_f || !w_hh_r || !whh_cf || !whh_cb) return TSSEP_E_NULL;
  if (!tssep_lstm_cluster_supported(H)) return TSSEP_E_UNSUPPORTED;
  const int G = (H + UPW - 1) / UPW;
  const int64_t total = tssep_lstm_cluster_pack_floats(H, 0) + tssep_lstm_cluster_pack_floats(H, 1);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(lstm_pack_cluster_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, w_hh_f, w_hh_r, H, G, whh_cf, whh_cb);
  return tssep_launch_status();
}

// Plan: ms (= M/4 sequences per cluster / 4) is 2 (two workgroups per CU can be resident: 256
// registers per lane) or 4; the number of clusters is capped by what can be resident at once so
// that the exchange latency of one cluster is hidden behind its neighbours' MFMAs.
static void cluster_plan(int64_t N, int G, int max_wgs, int ms_req, int* ms, int* nclusters) {
  int pick = ms_req;
  if (pick != 2 && pick != 4) {
    const int64_t w2 = 2 * ((N + 7) / 8);
    pick = (w2 <= (2 * max_wgs) / G) ? 2 : 4;
  }
  const int cap = ((pick == 2 ? 2 : 1) * max_wgs) / G > 0 ? ((pick == 2 ? 2 : 1) * max_wgs) / G : 1;
  const int64_t work = 2 * ((N + 4 * pick - 1) / (4 * pick));
  *ms = pick;
  *nclusters = (int)(work < cap ? work : cap);
}

// ---- exchange-buffer reset + launch epoch (common.h) ------------------------------------------------
__global__ __launch_bounds__(256) void xbuf_reset_kernel(uint4* __restrict__ p, int64_t n16,
                                                         int* __restrict__ err) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
    p[i] = uint4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && threadIdx.x == 0) err[1] = (int)(((unsigned)err[1] + 1u) & 0x3fffffffu);
}

int tssep_xbuf_reset(void* xbuf, size_t bytes, int* err, hipStream_t s) {
  const int64_t n16 = (int64_t)((bytes + 15) / 16);          // callers size the buffer in 16-byte units
  int64_t blocks = (n16 + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(xbuf_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)xbuf, n16, err);
  return tssep_launch_status();
}

extern "C" int64_t tssep_lstm_cluster_xbuf_bytes(int64_t N, int H, int backward, int max_wgs,
                                                 int ms_req) {
  const int G = (H + UPW - 1) / UPW;
  int ms, nc;
  cluster_plan(N, G, max_wgs, ms_req, &ms, &nc);
  const int M = 4 * ms;
  const int64_t groups = (N + M - 1) / M;
  const int64_t per_work = (int64_t)2 * M * G * UPW * (backward ? G : 1);
  return 64 + 2 * groups * per_work * 8;
}

extern "C" int tssep_blstm_cluster_fwd(float* gates, float* cell, float* hout, int64_t ldo,
                                       int64_t dstride, const float* whh_cf, void* xbuf, int* err,
                                       int64_t N, int64_t T, int H, int max_wgs, int ms_req,
                                       void* stream) {
  if (!gates || !cell || !hout || !whh_cf || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!tssep_lstm_cluster_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  if (max_wgs < G) return TSSEP_E_SHAPE;
  int ms, nc;
  cluster_plan(N, G, max_wgs, ms_req, &ms, &nc);
  hipStream_t s = (hipStream_t)stream;
  if (T >= 0xffff) return TSSEP_E_SHAPE;                      // the step shares the tag with the epoch
  if (tssep_xbuf_reset(xbuf, (size_t)tssep_lstm_cluster_xbuf_bytes(N, H, 0, max_wgs, ms_req), err, s) !=
      TSSEP_OK)
    return TSSEP_E_LAUNCH;
  dim3 grid((unsigned)(nc * G));
  if (ms == 2)
    hipLaunchKernelGGL((blstm_cluster_fwd_kernel<2>), grid, dim3(256), 0, s, gates, cell, hout, ldo,
                       dstride, whh_cf, (u64*)xbuf, err, N, T, H, G, nc);
  else
    hipLaunchKernelGGL((blstm_cluster_fwd_kernel<4>), grid, dim3(256), 0, s, gates, cell, hout, ldo,
                       dstride, whh_cf, (u64*)xbuf, err, N, T, H, G, nc);
  return tssep_launch_status();
}

extern "C" int tssep_blstm_cluster_bwd(float* gates, const float* cell, const float* dhout,
                                       int64_t ldo, int64_t dstride, const float* whh_cb,
                                       void* xbuf, int* err, int64_t N, int64_t T, int H,
                                       int max_wgs, int ms_req, void* stream) {
  if (!gates || !cell || !dhout || !whh_cb || !xbuf || !err) return TSSEP_E_NULL;
  if (N <= 0 || T <= 0 || dstride < H || ldo < dstride + H) return TSSEP_E_SHAPE;
  if (!tssep_lstm_cluster_supported(H)) return TSSEP_E_UNSUPPORTED;
  if (!aligned16(gates) || !aligned16(xbuf)) return TSSEP_E_ALIGN;
  const int G = (H + UPW - 1) / UPW;
  if (max_wgs < G) return TSSEP_E_SHAPE;
  int ms, nc;
  cluster_plan(N, G, max_wgs, ms_req, &ms, &nc);
  hipStream_t s = (hipStream_t)stream;
  if (T >= 0xffff) return TSSEP_E_SHAPE;                      // the step shares the tag with the epoch
  if (tssep_xbuf_reset(xbuf, (size_t)tssep_lstm_cluster_xbuf_bytes(N, H, 1, max_wgs, ms_req), err, s) !=
      TSSEP_OK)
    return TSSEP_E_LAUNCH;
  dim3 grid((unsigned)(nc * G));
  if (ms == 2)
    hipLaunchKernelGGL((blstm_cluster_bwd_kernel<2>), grid, dim3(256), 0, s, gates, cell, dhout,
                       ldo, dstride, whh_cb, (u64*)xbuf, err, N, T, H, G, nc);
  else
    hipLaunchKernelGGL((blstm_cluster_bwd_kernel<4>), grid, dim3(256), 0, s, gates, cell, dhout,
                       ldo, dstride, whh_cb, (u64*)xbuf, err, N, T, H, G, nc);
  return tssep_launch_status();
}
