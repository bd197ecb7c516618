// MFMA issue-rate probe: N back-to-back v_mfma_f32_32x32x16_bf16 on 4 rotating accumulators per wave, W waves per workgroup, one
// workgroup per CU; reports shader-clock cycles (s_memtime) and 100 MHz ticks (s_memrealtime) per wave.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(int n, unsigned long long* out, float* sink) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f32x16 acc[4] = {};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 12345.f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}
extern "C" int mfma_probe(int grid, int threads, int n, void* out, void* sink, hipStream_t s) {
  hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, s, n, (unsigned long long*)out, (float*)sink);
  return (int)hipGetLastError();
}
