// probe of ds_read_b64_tr_b16 semantics (gfx950): see scripts/tr_probe.py
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* src, unsigned short* dst) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 160];
  for (int i = threadIdx.x; i < 64 * 160; i += 64) lds[i] = src[i];
  __syncthreads();
  const int l = threadIdx.x, grp = l >> 4, li = l & 15;
  const int k0 = 8 * (grp >> 1), c0 = 16 * (grp & 1);
  const unsigned short* p = lds + (k0 + li / 4) * 160 + c0 + (li % 4) * 4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 160));
  for (int j = 0; j < 4; ++j) { dst[l * 8 + j] = a[j]; dst[l * 8 + 4 + j] = b[j]; }
}
extern "C" int tr_probe(const unsigned short* src, unsigned short* dst, hipStream_t s) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, src, dst); return (int)hipGetLastError(); }
