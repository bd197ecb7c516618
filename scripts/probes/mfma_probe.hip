// MFMA issue-rate probe: one wave per SIMD (256-thread workgroups, >256 registers), three accumulator chains, A operand from accumulator or vector registers.
// Prints shader cycles (s_memtime) and wall time (s_memrealtime, 100 MHz) per v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NCH, int RND>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink, int iters, unsigned seed) {
  bf16x8_t a[32], b0, b1;
  for (int i = 0; i < 32; ++i) {
    u32x4 v = {seed * (i + 1) * 2654435761u + threadIdx.x, seed ^ (i * 40503u + threadIdx.x * 7u), seed + i, seed * 3u + threadIdx.x};
    if (RND) { v &= 0x807f807fu; v |= 0x3f003f00u; } else v &= 0x3f803f80u;   // RND: random sign and mantissa, magnitude 0.5 .. 1
    a[i] = __builtin_bit_cast(bf16x8_t, v);
    if (MODE == 0) asm volatile("" : "+a"(a[i])); else asm volatile("" : "+v"(a[i]));
  }
  { u32x4 v = {0x3f803f80u, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u};
    if (RND) { v = u32x4{seed * 2246822519u + threadIdx.x * 3266489917u, seed * 668265263u ^ (threadIdx.x * 374761393u), seed + threadIdx.x * 2654435761u, (seed ^ threadIdx.x) * 40503u}; v &= 0x807f807fu; v |= 0x3f003f00u; }
    b0 = __builtin_bit_cast(bf16x8_t, v); v ^= RND ? 0x00550033u : 0u; b1 = __builtin_bit_cast(bf16x8_t, v); asm volatile("" : "+v"(b0)); asm volatile("" : "+v"(b1)); }
  f32x16 acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], (c & 1) ? b1 : b0, acc[c], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE, int NCH, int RND = 0> void run(const char* name, int grid) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 2 * grid * 8); hipMalloc(&sink, grid * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((probe<MODE, NCH, RND>), dim3(grid), dim3(256), 0, 0, out, sink, iters, 12345u + rep); hipEventRecord(e1); hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * 32 * NCH;
  printf("%-44s grid %4d: %.1f shader cycles / MFMA, %.2f ns / MFMA (realtime), clock %.2f GHz; kernel %.3f ms = %.1f TFLOP/s\n", name, grid, h[0] / n, h[1] * 10.0 / n,
         h[0] / (h[1] * 10.0), ms, grid * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(sink);
}
int main() {
  run<0, 3>("A in accumulator registers, 3 chains", 256);
  run<1, 3>("A in vector registers, 3 chains", 256);
  run<0, 1>("A in accumulator registers, 1 chain", 256);
  run<1, 4>("A in vector registers, 4 chains", 256);
  run<1, 3>("A in vector registers, 3 chains, 1 workgroup", 1);
  run<0, 3>("A in accumulator registers, 3 chains, 1 workgroup", 1);
  run<0, 3, 1>("random operands: A in accumulator registers, 3 chains", 256);
  run<1, 3, 1>("random operands: A in vector registers, 3 chains", 256);
  return 0;
}
