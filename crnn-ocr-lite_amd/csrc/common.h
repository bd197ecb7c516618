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

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER DEVICE.  The latch is a
// per-call-site bit mask over device ordinals (atomic: launchers may be called from several host threads, one per GPU); it caches a
// device property, it is not state of the computation.
#include <atomic>
static inline int crnn_lds_attr(const void* fn, int bytes, std::atomic<unsigned long long>& latch) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (latch.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  latch.fetch_or(bit, std::memory_order_release);
  return 0;
}
#define CRNN_LDS_ATTR(fn, bytes)                                              \
  do {                                                                        \
    static std::atomic<unsigned long long> latch__{0};                        \
    CRNN_TRY(crnn_lds_attr((const void*)(fn), (bytes), latch__));             \
  } while (0)

// Tuning knobs of the micro-benchmark / ablation builds (scripts/, -DCRNN_EXPERIMENT_HOOKS): an environment variable overrides
// a default.  The product library is built WITHOUT the macro: the knob is the constant, nothing reads the environment and
// there is no static state (include/crnn_mi355x.h: "no global mutable state").
#ifdef CRNN_EXPERIMENT_HOOKS
#include <stdlib.h>
static inline int crnn_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int crnn_knob(const char*, int dflt) { return dflt; }
#endif

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }
__device__ __forceinline__ float hard_sigmoid(float z) { return fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hs_grad_from_out(float a) { return (a > 0.f && a < 1.f) ? 0.2f : 0.f; }
// BatchNorm backward, second pass, one element: dx = scale * (gy - c1 - xhat * c2) with c1 = mean(gy), c2 = mean(gy * xhat),
// xhat = (x - mean) * inv, evaluated as two fused multiply-adds on per-channel constants:
//     dx = fma(x, P, fma(gy, scale, Q)),   P = -scale * c2 * inv,   Q = scale * c2 * inv * mean - scale * c1.
// One spelling (explicit fma, no compiler contraction) shared by the stand-alone pass (conv.hip) and the kernel that applies it
// while it fills its tiles (conv_bwd_fused.hip), so both produce the same bits; 2 instead of 5 VALU operations per element is what
// the fused kernel (VALU-bound) cares about.
__device__ __forceinline__ void bn_bwd_pq(float sc, float c1, float c2, float mu, float inv, float& P, float& Q) {
#pragma clang fp contract(off)
  const float k = sc * c2 * inv;
  P = -k;
  Q = fmaf(k, mu, -(sc * c1));
}
__device__ __forceinline__ float bn_bwd_dx_pq(float x, float gy, float sc, float P, float Q) { return fmaf(x, P, fmaf(gy, sc, Q)); }

// Counter-based dropout RNG (round 4): three ChaCha quarter-rounds (add / xor / rotate only -- every operation full-rate on the
// vector ALU; the murmur3-style 64-bit finaliser it replaces spent 8 quarter-rate 32-bit multiplies per 4 elements, half the VALU
// work of the kernels that re-derive the mask) over the state (group ^ k0, k1, k2, rotl(group, 16) ^ c) with the key (k0, k1, k2)
// folded from (step seed, dropout site).  One evaluation serves a group of 8 consecutive elements: element idx takes 16-bit lane
// (idx & 7) of the 128-bit result for group idx >> 3 and is kept when lane >= floor(rate * 65536).  The backward pass recomputes it
// (no mask tensor).  Avalanche over group / seed / site bits, lane uniformity and lane / stride correlations measured at 2^20
// groups: indistinguishable from ideal at that sample size (3 quarter-rounds; 2 leave 0.11-0.19 avalanche bias).
struct crnn_rng_key { uint32_t k0, k1, k2; };
__device__ __host__ __forceinline__ crnn_rng_key crnn_rng_make_key(uint64_t seed, uint32_t layer) {
  crnn_rng_key k;
  k.k0 = (uint32_t)seed ^ 0x243F6A88u; k.k1 = (uint32_t)(seed >> 32) ^ 0x85A308D3u; k.k2 = (layer * 0x9E3779B9u) ^ 0x13198A2Eu;
  return k;
}
__device__ __host__ __forceinline__ uint32_t crnn_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
__device__ __host__ __forceinline__ void crnn_rng8(const crnn_rng_key& k, uint64_t group, uint32_t (&w)[4]) {
  const uint32_t glo = (uint32_t)group, ghi = (uint32_t)(group >> 32);
  uint32_t a = glo ^ k.k0, b = k.k1, c = ghi ^ k.k2, d = crnn_rotl(glo, 16) ^ 0x03707344u;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a += b; d = crnn_rotl(d ^ a, 16);
    c += d; b = crnn_rotl(b ^ c, 12);
    a += b; d = crnn_rotl(d ^ a, 8);
    c += d; b = crnn_rotl(b ^ c, 7);
  }
  w[0] = a; w[1] = b; w[2] = c; w[3] = d;
}
__device__ __host__ __forceinline__ uint32_t crnn_drop_threshold(float rate) { return (uint32_t)(rate * 65536.f); }
// bit e of the result: element 8 * group + e is KEPT
__device__ __forceinline__ uint32_t crnn_keep8(const crnn_rng_key& k, uint64_t group, uint32_t thr) {
  uint32_t w[4];
  crnn_rng8(k, group, w);
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    m |= ((w[q] & 0xffffu) >= thr ? 1u : 0u) << (2 * q);
    m |= ((w[q] >> 16) >= thr ? 1u : 0u) << (2 * q + 1);
  }
  return m;
}
// returns the multiplier applied to the activation: 0 or 1/(1-rate); rate<=0 => 1
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t layer, uint64_t idx, float rate, float inv_keep) {
  if (rate <= 0.f) return 1.f;
  uint32_t w[4];
  crnn_rng8(crnn_rng_make_key(seed, layer), idx >> 3, w);
  const uint32_t e = (uint32_t)idx & 7u;
  const uint32_t lane = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
  return (lane >= crnn_drop_threshold(rate)) ? inv_keep : 0.f;
}
// multipliers of N consecutive elements idx0 .. idx0+N-1; one evaluation per aligned group of 8 when idx0 % 8 == 0 (N % 8 == 0), half of
// one when idx0 % 4 == 0 (N == 4)
template <int N>
__device__ __forceinline__ void drop_scale_vec(uint64_t seed, uint32_t layer, uint64_t idx0, float rate, float inv_keep, float* out) {
  if (rate <= 0.f) {
#pragma unroll
    for (int e = 0; e < N; ++e) out[e] = 1.f;
    return;
  }
  const uint32_t thr = crnn_drop_threshold(rate);
  const crnn_rng_key key = crnn_rng_make_key(seed, layer);
  if (N % 8 == 0) {
#pragma unroll
    for (int gq = 0; gq < N / 8; ++gq) {
      uint32_t w[4];
      crnn_rng8(key, (idx0 >> 3) + gq, w);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        out[(8 * gq + 2 * q) % N] = ((w[q] & 0xffffu) >= thr) ? inv_keep : 0.f;
        out[(8 * gq + 2 * q + 1) % N] = ((w[q] >> 16) >= thr) ? inv_keep : 0.f;
      }
    }
  } else if (N == 4) {
    uint32_t w[4];
    crnn_rng8(key, idx0 >> 3, w);
    const bool hi = (idx0 >> 2) & 1;
    const uint32_t w0 = hi ? w[2] : w[0], w1 = hi ? w[3] : w[1];
    out[0] = ((w0 & 0xffffu) >= thr) ? inv_keep : 0.f;
    out[1 % N] = ((w0 >> 16) >= thr) ? inv_keep : 0.f;
    out[2 % N] = ((w1 & 0xffffu) >= thr) ? inv_keep : 0.f;
    out[3 % N] = ((w1 >> 16) >= thr) ? inv_keep : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) out[e] = drop_scale(seed, layer, idx0 + e, rate, inv_keep);
  }
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

// Scalar fp32 fma / add that the compiler cannot SLP-pack into v_pk_fma_f32 / v_pk_add_f32: next to MFMA waves on the same SIMD the
// packed fp32 forms contend with the matrix pipe (MI355X_MICROARCH.md: "an anti-lever beside MFMAs"); same IEEE results.
__device__ __forceinline__ float fma_unpacked(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float add_unpacked(float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float mul_unpacked(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float sub_unpacked(float a, float b) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

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
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round-to-nearest-even each (the differences are exact in
// fp32): the three bf16 planes of the parity mode's products (gemm_bf16.inc: gemm_x3p_kernel; gemm_wres3.hip).  One spelling shared by every
// kernel that forms planes, so that all of them produce the same words.  w0 / w1 / w2 = the packed pair (v0 low half, v1 high half) of each plane.
// (scalar v_sub_f32: SLP-packed into v_pk_add_f32 these four subtractions cost ~25 cycles each beside running MFMAs)
__device__ __forceinline__ void crnn_split3_pair(float v0, float v1, unsigned& w0, unsigned& w1, unsigned& w2) {
#pragma clang fp contract(off)
  w0 = pack2_bf16(v0, v1);
  const float r0 = sub_unpacked(v0, __uint_as_float(w0 << 16)), r1 = sub_unpacked(v1, __uint_as_float(w0 & 0xffff0000u));
  w1 = pack2_bf16(r0, r1);
  const float s0 = sub_unpacked(r0, __uint_as_float(w1 << 16)), s1 = sub_unpacked(r1, __uint_as_float(w1 & 0xffff0000u));
  w2 = pack2_bf16(s0, s1);
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
struct float8 { float4 lo, hi; };
__device__ __forceinline__ float8 ld8(const float* p) { float8 r; r.lo = ld4(p); r.hi = ld4(p + 4); return r; }
__device__ __forceinline__ float8 ld8(const bf16_t* p) {   // one 16-byte load
  uint4 u = *reinterpret_cast<const uint4*>(p);
  float8 r;
  r.lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  r.hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
  return r;
}
__device__ __forceinline__ void st8(float* p, const float8& v) { st4(p, v.lo); st4(p + 4, v.hi); }
__device__ __forceinline__ void st8(bf16_t* p, const float8& v) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2_bf16(v.lo.x, v.lo.y), pack2_bf16(v.lo.z, v.lo.w), pack2_bf16(v.hi.x, v.hi.y), pack2_bf16(v.hi.z, v.hi.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack2_bf16(v, 0.f) & 0xffffu); }
