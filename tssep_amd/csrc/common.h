// Shared helpers for the gfx950 kernels of libtssep_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tssep_hip.h"

#define TSSEP_ABI_VERSION 4      // 3: + tssep_stft_plan (general FFT plans), tssep_*onchip16w*; 4: - tssep_*onchip16w* (experiment build only)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int tssep_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TSSEP_OK : TSSEP_E_LAUNCH;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

__device__ __forceinline__ float sigmoidf_acc(float x) {
  // 1/(1+exp(-x)) with the accurate expf (parity with torch.sigmoid to ~1 ulp)
  return 1.0f / (1.0f + expf(-x));
}
// mask-head sigmoid (net.py:983) on the hardware exp2 / reciprocal units: 6 instructions instead of ~25 for the
// accurate expf and the IEEE division (8 per lane and frame in the fused FFT kernels, a fifth of their vector
// instructions); relative error ~2^-22, the same value in the fused and the unfused mask-head kernels
__device__ __forceinline__ float sigmoidf_mask(float x) {
  // v_exp_f32 and v_rcp_f32 themselves (1 ulp each): __expf adds a range fix-up of compares and selects that the
  // sigmoid does not need (exp2 -> inf gives 0, -> 0 gives 1), and __frcp_rn expands to the ten-instruction IEEE
  // division -- together a tenth of the vector instructions of the fused FFT kernels, which are VALU-bound
  // (SQ_ACTIVE_INST_VALU x 3 waves per SIMD = 0.6-0.7, profiles/r3_sq_wave_states.jsonl)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- launch epoch of the W-stationary recurrences (lstm_cluster.hip, lstm_onchip.hip) -------------
// Granule tags are {launch epoch (15 bits, never 0) << 16 | step + 1}: a granule left in a cache or in
// memory by an EARLIER launch over the same buffer can never satisfy a later launch's poll (observed:
// with a second stream active, stale L2 lines of the previous launch carried tags that matched the same
// step of the next one; consumers ran ahead and the two-slot protocol broke).  The epoch lives in DEVICE
// memory (err[1], next to the error flag err[0]) and is advanced by the reset kernel that zeroes the
// exchange buffer in front of every launch, so a captured hipGraph replays with fresh epochs too (a
// host-side counter would be frozen into the captured kernel arguments).
int tssep_xbuf_reset(void* xbuf, size_t bytes, int* err, hipStream_t s);
__device__ __forceinline__ unsigned tssep_load_tagbase(const int* err) {
  const unsigned e = (unsigned)__hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(((e % 0x7fffu) + 1u) << 16));
}
