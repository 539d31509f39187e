// Shared helpers for the gfx950 kernels of libtssep_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tssep_hip.h"

#define TSSEP_ABI_VERSION 1

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
