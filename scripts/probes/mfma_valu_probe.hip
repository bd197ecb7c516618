// Do vector-ALU instructions of one wave issue beside the MFMAs of another wave of the same SIMD?  (round 3: the producer-wave
// three-plane GEMM runs at ~0.5 of its MFMA floor; this probe separates "issue slots are shared" from everything else.)
// One workgroup per CU (100 KB of LDS), 8 waves: waves 0-3 (one per SIMD) loop over independent v_mfma_f32_32x32x16_bf16, waves 4-7 loop
// over a VALU mix.  Timed three ways: MFMA waves only, VALU waves only, both.  both ~ max(...) => the pipes overlap; ~ sum => they do not.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_valu_probe scripts/probes/mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__device__ __forceinline__ void valu_body(float (&x)[8], unsigned (&w)[4]) {
  if constexpr (KIND == 0) {          // fp32 subtract chains (8 independent)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
  } else if constexpr (KIND == 1) {   // the split: cvt_pk + shifts/ands + subs (the 11-instruction pair split, twice)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned a, b, c; float r0, r1, s0, s1;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(x[4 * i]), "v"(x[4 * i + 1]));
      asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(r0) : "v"(a), "v"(x[4 * i]));
      asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(r1) : "v"(a), "v"(x[4 * i + 1]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b) : "v"(r0), "v"(r1));
      asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(s0) : "v"(b), "v"(r0));
      asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(s1) : "v"(b), "v"(r1));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c) : "v"(s0), "v"(s1));
      w[2 * i] ^= a ^ b; w[2 * i + 1] ^= c;
    }
  } else if constexpr (KIND == 3) {   // the LDS side of staging: 8-byte stores, 512 contiguous bytes per wave-instruction
    extern __shared__ unsigned char lds[];
    unsigned char* base = lds + 61440 + (threadIdx.x - 256) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)(base + i * 2048 - lds)), "v"(make_uint2(w[i & 3], w[(i + 1) & 3])) : "memory");
  } else if constexpr (KIND == 2) {   // integer ops
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_and_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2" : "+v"(w[i]) : "v"(w[(i + 1) & 3]), "v"(w[(i + 2) & 3]));
  }
}
// instructions per call of valu_body
static const int kValuInstr[4] = {8, 26, 8, 8};

template <int KIND>
__global__ __launch_bounds__(512) void probe(int mode, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    const unsigned char* rd = lds + (wave & 1) * 5120 + (lane & 31) * 80 + (lane >> 5) * 16;
    for (int it = 0; it < iters; ++it) {
      if (mode & 4) {                    // the fragment reads of a 16-k step of the three-plane kernel: 12 x ds_read_b128, 80-byte rows
        bf16x8 f[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) f[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile uint4*>(rd + r * 2560 + (it & 1) * 30720));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(u * 4 + i) % 12], f[(u * 4 + i + 5) % 12], acc[i], 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_readcyclecounter() - t0;
  } else {
    if (!(mode & 2)) return;
    float x[8]; unsigned w[4];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane * 8 + i) * 1.0009765625f;
    for (int i = 0; i < 4; ++i) w[i] = lane * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) valu_body<KIND>(x, w);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)(w[0] ^ w[1] ^ w[2] ^ w[3]);
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = __builtin_readcyclecounter() - t0;
  }
}

template <int KIND>
static void run(const char* name, float* out, unsigned long long* cyc, int rdflag = 0) {
  const int blocks = 256, iters = 4096, lds = 100 * 1024;
  hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      const int m = mode | ((mode & 1) ? rdflag : 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(512), lds, 0, m, iters, out, cyc);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
  }
  const double mf = 16.0 * iters, vi = 8.0 * kValuInstr[KIND] * iters;
  printf("%-22s mfma only %8.1f us (%5.1f ns/MFMA)   valu only %8.1f us (%5.2f ns/instr)   both %8.1f us   sum %8.1f   max %8.1f\n", name,
         ms[1] * 1e3, ms[1] * 1e6 / mf, ms[2] * 1e3, ms[2] * 1e6 / vi, ms[3] * 1e3, (ms[1] + ms[2]) * 1e3, (ms[1] > ms[2] ? ms[1] : ms[2]) * 1e3);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run<0>("v_sub_f32", out, cyc);
  run<1>("three-plane split", out, cyc);
  run<2>("v_and/v_xor", out, cyc);
  run<3>("ds_write_b64", out, cyc);
  run<2>("and/xor | mfma+ds_read", out, cyc, 4);
  run<1>("split | mfma+ds_read", out, cyc, 4);
  run<3>("ds_write | mfma+ds_read", out, cyc, 4);
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  return 0;
}
