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

extern "C" int tssep_abi_version(void) { return TSSEP_ABI_VERSION; }
extern "C" const char* tssep_arch(void) { return "gfx950"; }
