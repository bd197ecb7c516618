// Ingest probe 2: LDS-DMA of 128-byte row segments (the GEMM loaders' pattern): stripes of 128 rows with row stride RS bytes, one
// 128-byte segment of every row per stage, RS/128 stages per stripe -- against fully contiguous reads of the same bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// grid workgroups x 2 waves; workgroup w handles stripes w, w + grid, ...; share = number of consecutive workgroups reading the SAME stripes
__global__ __launch_bounds__(128) void probe2(const unsigned char* base, long rs, int nstripes, int share, int contiguous, int swz, unsigned* sink) {
  const int bar = swz >> 1; swz &= 1;   // swz bit 1: s_barrier after every stage's wait
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nxcd = 8;
  const int x = blockIdx.x % nxcd, j = blockIdx.x / nxcd;
  const int q = j / share;
  const int Q = gridDim.x / nxcd / share;
  const int segs = (int)(rs / 128);
  int slot = 0;
  for (int st = q * nxcd + x; st < nstripes; st += Q * nxcd) {
    const unsigned char* sb = base + (long)st * 128 * rs;
    for (int sg = 0; sg < segs; ++sg) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jrow = wv * 8 + u;
        const int row = jrow * 8 + (lane >> 3);
        int c = lane & 7;
        if (swz) c ^= (4 * jrow + (lane >> 4)) & 7;
        const unsigned char* a = contiguous ? sb + ((long)sg * 128 + row) * 128 + c * 16 : sb + (long)row * rs + sg * 128 + c * 16;
        glds16(a, smem + slot * 16384 + jrow * 1024);
      }
      slot = (slot + 1) & 7;
      asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      if (bar) __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) sink[0] = 1;
}
extern "C" int ingest_probe2(const void* base, long rs, int nstripes, int share, int contiguous, int swz, int grid, void* sink, hipStream_t s) {
  static bool done = false;
  if (!done) { hipFuncSetAttribute((const void*)probe2, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); done = true; }
  hipLaunchKernelGGL(probe2, dim3(grid), dim3(128), 131072, s, (const unsigned char*)base, rs, nstripes, share, contiguous, swz, (unsigned*)sink);
  return (int)hipGetLastError();
}
