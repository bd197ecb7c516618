// Shared device/host helpers for libcrnn_mi355x (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "crnn_mi355x.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CRNN_OK 0
#define CRNN_ERR_ARG (-2)
#define CRNN_ERR_UNSUPPORTED (-3)

// Every launcher returns 0 or a hipError_t (positive) / library code (negative).
#define CRNN_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

#define CRNN_TRY(expr)            \
  do {                            \
    int r__ = (expr);             \
    if (r__ != 0) return r__;     \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ float hard_sigmoid(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hs_grad_from_out(float a) { return (a > 0.f && a < 1.f) ? 0.2f : 0.f; }

// Counter-based dropout RNG: keep-decision for element `idx` of dropout site `layer` in step
// `seed`.  murmur3-style 64->32 finaliser; the backward pass recomputes it (no mask tensor).
__device__ __host__ __forceinline__ uint32_t crnn_hash(uint64_t seed, uint32_t layer, uint64_t idx) {
  uint64_t x = idx + 0x9E3779B97F4A7C15ull * (seed + 1) + ((uint64_t)layer << 56);
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return (uint32_t)x;
}
// returns the multiplier applied to the activation: 0 or 1/(1-rate); rate<=0 => 1
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t layer, uint64_t idx, float rate, float inv_keep) {
  if (rate <= 0.f) return 1.f;
  float u = (float)(crnn_hash(seed, layer, idx) >> 8) * (1.0f / 16777216.0f);
  return (u >= rate) ? inv_keep : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- storage types: fp32 or bf16 (bf16 = upper half of the fp32 bit pattern, round-to-nearest-even on store) ----
typedef unsigned short bf16_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define CRNN_F32 0
#define CRNN_BF16 1

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack2_bf16(v, 0.f) & 0xffffu); }
