// What does a plain copy of the depthwise forward's bytes achieve on this chip?  The six launches of a step read x and write d of the same
// size (bf16 [256][H][W][C]); this probe copies six buffer pairs of exactly those sizes back to back -- cold, like bench.py's
// depthwise_roofline: every buffer last touched five launches earlier, 1.7 GB working set against the 256 MB last-level cache -- with
// hipMemcpyDtoDAsync and with grid-stride 16-byte copy kernels.  The rate is bytes read + written per second, the roofline's convention.
// build: hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/copy_probe scripts/probes/copy_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ s, u32x4* __restrict__ d, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
  }
}
// the access pattern of a band-streaming kernel: workgroup b copies ITS OWN contiguous 1/gridDim.x of the buffer front to back (16 KiB per iteration),
// so the resident workgroups read and write at gridDim.x places far apart instead of sweeping one window together
__global__ __launch_bounds__(256) void copy_banded_kernel(const u32x4* __restrict__ s, u32x4* __restrict__ d, long n) {
  const long per = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long i = lo + threadIdx.x; i < hi; i += 1024) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) v[u] = s[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) d[i + u * 256] = v[u];
  }
}
int main() {
  const long B = 256;
  const long shp[6][3] = {{104, 36, 64}, {104, 36, 128}, {52, 18, 256}, {52, 18, 256}, {52, 9, 512}, {52, 9, 512}};
  void *src[6], *dst[6]; long bytes[6]; double total = 0;
  for (int i = 0; i < 6; ++i) {
    bytes[i] = B * shp[i][0] * shp[i][1] * shp[i][2] * 2;
    if (hipMalloc(&src[i], bytes[i]) != hipSuccess || hipMalloc(&dst[i], bytes[i]) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src[i], 1, bytes[i]); hipMemset(dst[i], 2, bytes[i]);
    total += 2.0 * bytes[i];
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int kind, int wgs) {
    std::vector<float> ts;
    for (int it = 0; it < 16; ++it) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 6; ++i) {
        const long n = bytes[i] / 16;
        if (kind == 0) hipMemcpyDtoDAsync(dst[i], src[i], bytes[i], 0);
        else if (kind == 1) hipLaunchKernelGGL(copy_kernel<0>, dim3(wgs), dim3(256), 0, 0, (const u32x4*)src[i], (u32x4*)dst[i], n);
        else if (kind == 3) hipLaunchKernelGGL(copy_banded_kernel, dim3(wgs), dim3(256), 0, 0, (const u32x4*)src[i], (u32x4*)dst[i], n);
        else hipLaunchKernelGGL(copy_kernel<1>, dim3(wgs), dim3(256), 0, 0, (const u32x4*)src[i], (u32x4*)dst[i], n);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double t = ts[ts.size() / 2] * 1e-3;
    printf("%-44s %7.1f us per set  %6.0f GB/s  %.3f of 8 TB/s\n", name, t * 1e6, total / t / 1e9, total / t / 8e12);
  };
  printf("six copies of the depthwise forward's tensors (%.0f MB read + written per set), median of 15 cold sets\n", total / 1e6);
  run("hipMemcpyDtoDAsync", 0, 0);
  char nm[64];
  for (int w : {256, 512, 1024, 2048, 4096, 16384}) { snprintf(nm, 64, "copy kernel, %d workgroups", w); run(nm, 1, w); }
  for (int w : {256, 1024, 4096}) { snprintf(nm, 64, "copy kernel, nontemporal, %d workgroups", w); run(nm, 2, w); }
  for (int w : {256, 512, 1024, 2048}) { snprintf(nm, 64, "copy kernel, own band per workgroup, %d workgroups", w); run(nm, 3, w); }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
