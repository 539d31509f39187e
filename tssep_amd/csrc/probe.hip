// Hardware self-check: dump the lane<->element maps of the two MFMA instructions the
// kernels rely on, so tests can assert the layout assumptions on the real chip.
#include "common.h"

namespace {
// out_4x4[64 lanes][4]: D of v_mfma_f32_4x4x1_16b_f32 with A = lane id, B = 1000 + lane id... encoded
// as D = A*B so tests can factor it.  out_32[64][16] likewise for 32x32x2 (k-sum of two products).
__global__ void probe_kernel(float* out4, float* out32) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // A value = 1 + lane, B value = 101 + lane  (products identify both source lanes uniquely enough)
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(101 + lane), c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out4[lane * 4 + i] = c[i];
  f32x16 d;
  for (int i = 0; i < 16; ++i) d[i] = 0.f;
  // A(i,k): lane = i + 32k ; choose A = (i+1) if k==0 else 0  -> D[i][j] = (i+1)*B(0,j)
  // B(k,j): lane = j + 32k ; B = 100 + j if k==0 else 7 (multiplied by A=0)
  const float a = lane < 32 ? (float)(lane + 1) : 0.f;
  const float b = lane < 32 ? (float)(100 + lane) : 7.f;
  d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out32[lane * 16 + i] = d[i];
  // second half (k=1): A = (i+1) for upper lanes only, B = 200 + j
  f32x16 d2;
  for (int i = 0; i < 16; ++i) d2[i] = 0.f;
  const float a2 = lane >= 32 ? (float)(lane - 32 + 1) : 0.f;
  const float b2 = lane >= 32 ? (float)(200 + lane - 32) : 7.f;
  d2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, d2, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out32[1024 + lane * 16 + i] = d2[i];
}
}  // namespace

extern "C" int tssep_probe_mfma(float* out_4x4, float* out_32x32, void* stream) {
  if (!out_4x4 || !out_32x32) return TSSEP_E_NULL;
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_4x4, out_32x32);
  return tssep_launch_status();
}

// XCD placement probe: out[b] = HW_REG_XCC_ID of workgroup b (the on-chip recurrence builds its
// clusters from workgroups of one XCD and relies on nothing but this register).
__global__ __launch_bounds__(512) void probe_xcc_kernel(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
}
extern "C" int tssep_probe_xcc(int* out, int nblocks, void* stream) {
  if (!out) return TSSEP_E_NULL;
  if (nblocks <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(probe_xcc_kernel, dim3((unsigned)nblocks), dim3(512), 0, (hipStream_t)stream, out);
  return tssep_launch_status();
}

// Shader clock under load: every wave runs `iters` x 8 back-to-back bf16 MFMAs (heavy = 1) or the
// same number of s_sleep (heavy = 0) and reports s_memtime (shader clock) and wall_clock64 (the
// constant 100 MHz reference) deltas.  out[2*b] = shader ticks, out[2*b+1] = 100 MHz ticks of block b.
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void probe_clock_kernel(long long* out, int iters, int heavy) {
  pf32x16 a0, a1, a2, a3;
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
  pbf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (threadIdx.x + e)); y[e] = (__bf16)(0.002f * e); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (heavy) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    } else {
      __builtin_amdgcn_s_sleep(16);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (long long)(t1 - t0) + (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f ? 1 : 0);
    out[2 * blockIdx.x + 1] = (long long)(w1 - w0);
  }
}
extern "C" int tssep_probe_clock(int64_t* out_, int nblocks, int iters, int heavy, void* stream) {
  long long* out = reinterpret_cast<long long*>(out_);
  if (!out) return TSSEP_E_NULL;
  if (nblocks <= 0 || iters <= 0) return TSSEP_E_SHAPE;
  hipLaunchKernelGGL(probe_clock_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, out,
                     iters, heavy);
  return tssep_launch_status();
}

// Store-flavour probe (round 4): does a line that is REWRITTEN while it sits in the XCD's L2 reach the
// memory side once or every time?  Every workgroup rewrites its own `bytes_per_wg` region `reps` times with 16-byte
// stores of one flavour (0 plain, 1 sc0, 2 sc1, 3 sc0 sc1, 4 nt -- the exchange granules of the W-stationary
// recurrences are sc0 stores) and, with `pressure` > 0, streams `pressure` bytes of a read-only buffer through the
// L2 between two rewrites (the recurrences' activation stream).  Read with rocprofv3 --pmc WRITE_SIZE:
// tools/probe_rewrite.py, profiles/r4_store_flavour_probe.json.
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
// RD: how a PEER workgroup of the same XCD (block b + 8) reads the region after every rewrite -- the gather of the
// recurrences' exchange: 0 = nobody reads, else the aux bits of the 16-byte buffer loads (16 = sc1, 2 = nt, 17 = sc0 sc1,
// 1 = sc0).  Does a peer's L1-bypassing read make the L2 write the dirty line back?
template <int AUX, int RD>
__global__ __launch_bounds__(256) void probe_rewrite_kernel(float* buf, int bytes_per_wg, int reps, const float* stream_src,
                                                            long long stream_bytes, int pressure, float* sink) {
  char* mine = reinterpret_cast<char*>(buf) + (size_t)blockIdx.x * bytes_per_wg;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, bytes_per_wg, 0x00020000);
  const unsigned peer = (blockIdx.x + 8u) % gridDim.x;
  const auto rp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(buf) + (size_t)peer * bytes_per_wg, 0, bytes_per_wg, 0x00020000);
  float acc = 0.f;
  const long long per_wg = pressure;
  for (int r = 0; r < reps; ++r) {
    const pu32x4 v = {(unsigned)r, threadIdx.x, blockIdx.x, 0x5eedu};
    for (int off = threadIdx.x * 16; off < bytes_per_wg; off += 256 * 16)
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
    if (pressure > 0) {
      const long long base = (((long long)blockIdx.x * reps + r) * per_wg) % (stream_bytes - per_wg);
      const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(stream_src) + (base & ~15ll));
      for (long long i = threadIdx.x; i < per_wg / 16; i += 256) { const f32x4 q = __builtin_nontemporal_load(src + i); acc += q[0] + q[3]; }
      // ... and writes two thirds as much back non-temporally (the d(gates) stream of the backward recurrence), into the
      // upper half of stream_src
      f32x4* dst = reinterpret_cast<f32x4*>(const_cast<char*>(reinterpret_cast<const char*>(stream_src)) + stream_bytes + (base & ~15ll));
      for (long long i = threadIdx.x; i < per_wg / 24; i += 256) __builtin_nontemporal_store(f32x4{acc, 1.f, 2.f, 3.f}, dst + i);
    }
    if (RD) {
      for (int off = threadIdx.x * 16; off < bytes_per_wg; off += 256 * 16) {
        const pu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, RD);
        acc += (float)(q[0] & 1u);
      }
    }
    __syncthreads();
  }
  if (acc == 12345.678f) sink[0] = acc;
}
extern "C" int tssep_probe_rewrite(float* buf, int nblocks, int bytes_per_wg, int reps, int flavour, int read_flavour,
                                   const float* stream_src, int64_t stream_bytes, int pressure, float* sink, void* stream) {
  if (!buf || !sink || (pressure > 0 && !stream_src)) return TSSEP_E_NULL;
  if (nblocks <= 0 || bytes_per_wg < 4096 || (bytes_per_wg & 4095) || reps <= 0 || flavour < 0 || flavour > 4 ||
      read_flavour < 0 || read_flavour > 4 || (pressure > 0 && ((pressure & 4095) || stream_bytes < 2 * (int64_t)pressure)))
    return TSSEP_E_SHAPE;
#define PR(AUX_, RD_) hipLaunchKernelGGL((probe_rewrite_kernel<AUX_, RD_>), dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, buf, \
                                         bytes_per_wg, reps, stream_src, (long long)stream_bytes, pressure, sink)
#define PRR(AUX_) switch (read_flavour) { case 0: PR(AUX_, 0); break; case 1: PR(AUX_, 16); break; case 2: PR(AUX_, 2); break; \
                                          case 3: PR(AUX_, 17); break; default: PR(AUX_, 1); break; }
  switch (flavour) {
    case 0: PRR(0); break;
    case 1: PRR(1); break;
    case 2: PRR(16); break;
    case 3: PRR(17); break;
    default: PRR(2); break;
  }
#undef PRR
#undef PR
  return tssep_launch_status();
}

extern "C" int tssep_abi_version(void) { return TSSEP_ABI_VERSION; }
extern "C" const char* tssep_arch(void) { return "gfx950"; }
