// Per-CU ingest probe: how fast can one workgroup per CU pull L2-/HBM-resident rows, (a) global_load_dwordx4 into registers,
// (b) global_load_lds b128 straight into LDS.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC ingest_probe.hip -o libingest.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// each workgroup reads `bytes_per_wg` bytes starting at base + wg * stride_wg (mod span), 16 B per lane per instruction
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(const unsigned char* base, long span, long stride_wg, long bytes_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const long mask = span - 1;   /* span is a power of two */
  const long start = (blockIdx.x * stride_wg) & mask;
  const long per_iter = 64L * WAVES * 16 * 8;     // 8 instructions per thread per iteration
  u32x4 acc = {0, 0, 0, 0};
  for (long off = 0; off < bytes_per_wg; off += per_iter) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      long a = (start + off + (long)u * 64 * WAVES * 16 + tid * 16) & mask;
      if (MODE == 0) { acc ^= *reinterpret_cast<const u32x4*>(base + a); }
      else { glds16(base + a, smem + ((u * WAVES + (tid >> 6)) * 1024)); }
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc.x = reinterpret_cast<unsigned*>(smem)[tid]; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
extern "C" int ingest_probe(int mode, int waves, const void* base, long span, long stride_wg, long bytes_per_wg, int grid, void* sink, hipStream_t s) {
  const int lds = 8 * 8 * 1024;
#define GO(M, W) hipLaunchKernelGGL((probe<M, W>), dim3(grid), dim3(64 * W), M ? lds : 0, s, (const unsigned char*)base, span, stride_wg, bytes_per_wg, (unsigned*)sink)
  if (mode == 0 && waves == 2) GO(0, 2); else if (mode == 0 && waves == 4) GO(0, 4); else if (mode == 0 && waves == 8) GO(0, 8);
  else if (mode == 1 && waves == 2) GO(1, 2); else if (mode == 1 && waves == 4) GO(1, 4); else if (mode == 1 && waves == 8) GO(1, 8);
  else return -1;
  return (int)hipGetLastError();
}
