// Measurement / test hooks -- NOT part of the product library: built into libcrnn_testhooks.so (include/crnn_testhooks.h), loaded only by
// bench.py's copy reference, scripts/ and tests/.  Nothing under crnn_mi355x/ or libcrnn_mi355x.so depends on it.
#include "common.h"
#include "crnn_testhooks.h"

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
}

// ---- measurement reference (bench.py depthwise_roofline "copy_reference"): what a plain copy of the same bytes achieves -----------------
// pattern 0: grid-stride, the resident workgroups sweep ONE window of the buffer together (16 KiB per workgroup and iteration);
// pattern 1: workgroup b copies its own contiguous 1/gridDim.x of the buffer front to back -- the access pattern of the row-stream kernels
//            (one image band per workgroup: reads and writes at gridDim.x places far apart).
namespace {
template <int BANDED>
__global__ __launch_bounds__(256) void debug_copy_kernel(const u32x4* __restrict__ s, u32x4* __restrict__ d, long n) {
  long lo, hi, stride;
  if (BANDED) { const long per = (n + gridDim.x - 1) / gridDim.x; lo = blockIdx.x * per; hi = lo + per < n ? lo + per : n; stride = 1024; }
  else { lo = (long)blockIdx.x * 1024; hi = n; stride = (long)gridDim.x * 1024; }
  for (long i = lo + threadIdx.x; i < hi; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) v[u] = s[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) d[i + u * 256] = v[u];
  }
}
}  // namespace
extern "C" int crnn_debug_copy(const void* src, void* dst, size_t bytes, int pattern, int workgroups, hipStream_t stream) {
  if (!src || !dst || (bytes & 15) || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) || workgroups < 1 || pattern < 0 || pattern > 1) return CRNN_ERR_ARG;
  const long n = (long)(bytes / 16);
  if (n == 0) return CRNN_OK;
  if (pattern) hipLaunchKernelGGL(debug_copy_kernel<1>, dim3(workgroups), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, n);
  else hipLaunchKernelGGL(debug_copy_kernel<0>, dim3(workgroups), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, n);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Test hook: workgroups that pin LDS and spin on the constant 100 MHz clock (bounded) -- tests/test_gpu_ops.py uses it to
// take the CUs away from a persistent recurrence and asserts that the give-up is reported instead of silently wrong numbers.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, int lds_bytes) {
  extern __shared__ unsigned char pin[];
  if (threadIdx.x == 0) pin[lds_bytes - 1] = 1;    // the allocation is what matters
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace
extern "C" int crnn_debug_occupy(int blocks, int lds_bytes, long microseconds, hipStream_t stream) {
  if (blocks < 1 || lds_bytes < 1 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 30L * 1000 * 1000) return CRNN_ERR_ARG;
  CRNN_LDS_ATTR(occupy_kernel, 160 * 1024);
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), lds_bytes, stream, (unsigned long long)microseconds * 100ull, lds_bytes);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
