// What does a tiny dependent kernel after a big one cost, against doing its work in the big kernel's last workgroup to finish?
// big: a streaming pass over `bytes` that leaves per-workgroup partial sums (grid x C floats); small: sums the partials into C floats (the shape of the
// BatchNorm finalizes).  Chains of 20, timed with events:
//   0 big only | 1 big + small kernel | 2 big with the tail behind __threadfence() + ticket | 3 big + 2 small kernels
//   4 tail without fences: partials stored write-through (sc1), s_waitcnt vmcnt(0), relaxed device-scope ticket, the last workgroup loads them with sc1
// Every tail checks the sum it read against the value it must have (partials carry launch number + workgroup index) and counts mismatches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int C = 256;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
template <bool SC1>
__device__ __forceinline__ float colsum(const float* parts, int rows, int c) {   // 16 loads in flight
  const __amdgpu_buffer_rsrc_t rs = rsrc(parts, rows * C * 4);
  float t[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = 0.f;
  for (int r = 0; r < rows; r += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] += SC1 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ((r + j) * C + c) * 4, 0, 16)) : parts[(long)(r + j) * C + c];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += t[j];
  return s;
}
template <int TAIL>
__global__ __launch_bounds__(256) void big(const float4* x, float4* y, long n4, float* parts, float* out, unsigned* ticket, int it, unsigned* err) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = x[i]; s += v.x + v.y + v.z + v.w; v.x += 1.f; y[i] = v;
  }
  s += (float)(it + (int)blockIdx.x);
  if (TAIL == 2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rsrc(parts, gridDim.x * C * 4), (blockIdx.x * C + threadIdx.x) * 4, 0, 16);
  else parts[(long)blockIdx.x * C + threadIdx.x] = s;
  if (TAIL) {
    __shared__ unsigned last;
    if (TAIL == 1) __threadfence(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = TAIL == 1 ? atomicAdd(ticket, 1u) : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (last != gridDim.x - 1) return;
    if (TAIL == 1) __threadfence();
    const float t = TAIL == 1 ? colsum<false>(parts, gridDim.x, threadIdx.x) : colsum<true>(parts, gridDim.x, threadIdx.x);
    out[threadIdx.x] = t;
    const float want = (float)gridDim.x * it + 0.5f * gridDim.x * (gridDim.x - 1);
    if (t != want) atomicAdd(err, 1u);
    if (threadIdx.x == 0) *ticket = 0;
  }
}
__global__ __launch_bounds__(256) void empty(float* out) { if (out == nullptr) out[0] = 1.f; }
__global__ __launch_bounds__(1024) void small4(const float* parts, int rows, float* out) {   // 4 row groups side by side, LDS fold
  __shared__ float f[4][C];
  const int g = threadIdx.x >> 8, c = threadIdx.x & 255, per = rows / 4;
  f[g][c] = colsum<false>(parts + (long)g * per * C, per, c);
  __syncthreads();
  if (g == 0) out[c] = (f[0][c] + f[1][c]) + (f[2][c] + f[3][c]);
}
__global__ __launch_bounds__(256) void small(const float* parts, int rows, float* out) { out[threadIdx.x] = colsum<false>(parts, rows, threadIdx.x); }
int main(int argc, char** argv) {
  const long bytes = (argc > 1 ? atol(argv[1]) : 128) << 20;
  const int grid = argc > 2 ? atoi(argv[2]) : 256;
  float4 *x, *y; float *parts, *out, *out2; unsigned *ticket, *err;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&parts, (size_t)grid * C * 4)); CK(hipMalloc(&out, C * 4)); CK(hipMalloc(&out2, C * 4)); CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(x, 0, bytes)); CK(hipMemset(ticket, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long n4 = bytes / 16;
  const char* names[8] = {"big only", "big + small kernel", "big with tail (__threadfence + ticket)", "big + 2 small kernels", "big with tail (write-through partials, no fence)", "big + empty kernel", "big + small kernel over 16 rows", "big + small kernel, 1024 threads"};
  for (int mode = 0; mode < 8; ++mode) {
    float best = 1e9f; int it = 0;
    CK(hipMemset(err, 0, 4));
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 20; ++i, ++it) {
        if (mode == 2) hipLaunchKernelGGL(big<1>, dim3(grid), dim3(256), 0, s, x, y, n4, parts, out2, ticket, it, err);
        else if (mode == 4) hipLaunchKernelGGL(big<2>, dim3(grid), dim3(256), 0, s, x, y, n4, parts, out2, ticket, it, err);
        else hipLaunchKernelGGL(big<0>, dim3(grid), dim3(256), 0, s, x, y, n4, parts, out, ticket, it, err);
        if (mode == 1) hipLaunchKernelGGL(small, dim3(1), dim3(256), 0, s, parts, grid, out);
        if (mode == 5) hipLaunchKernelGGL(empty, dim3(1), dim3(256), 0, s, out);
        if (mode == 6) hipLaunchKernelGGL(small, dim3(1), dim3(256), 0, s, parts, 16, out);
        if (mode == 7) hipLaunchKernelGGL(small4, dim3(1), dim3(1024), 0, s, parts, grid, out);
        if (mode == 3) { hipLaunchKernelGGL(small, dim3(1), dim3(256), 0, s, parts, grid, out); hipLaunchKernelGGL(small, dim3(1), dim3(256), 0, s, parts, grid, out); }
      }
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%ld MiB, grid %d, %s: %.2f us per launch set", bytes >> 20, grid, names[mode], best * 1e3f / 20);
    if (mode == 2 || mode == 4) printf("  (%u wrong sums in %d tails)", herr, it);
    printf("\n");
  }
  return 0;
}
