// Pixel-streaming weight gradient of a pointwise (1x1) convolution in the PARITY mode (utils.py:44-49, training backward; fp32 tensors):
//     dW[K][N] = ReLU6(BN1(d))^T [K][M] . g[M][N]      d [M][K], g [M][N] fp32 (NHWC rows), dW fp32, K = channels in, N = channels out
// with two bf16 planes per operand and the three products hi*hi + hi*mid + mid*hi (16 significant bits per factor: the parity mode's default backward
// precision, crnn_gemm_f32x2 / include/crnn_mi355x.h).  The tile kernel (gemm_bf16.inc: gemm_x3p_kernel mode 2) runs this as 128 x 128 output tiles x ~32
// reduction ranges, two workgroups per CU, 16 pixels per stage and barrier: 202 us at K = N = 512 against 98 us of fp32 traffic and 75 us of MFMAs.  Here the
// bf16 mode's stream (gemm_wgrad.hip: pw_wgrad_stream_kernel) is carried over to plane operands:
//   * a workgroup (512 threads) owns one 128 x 128 output tile over a contiguous range of 32-pixel chunks; the 4 MFMA waves (2 x 2, 64 x 64 each: 4 accumulator
//     blocks) keep the tile in registers for the whole range; per 16-pixel step 12 MFMAs on 4 independent accumulators;
//   * the 4 IO waves load the fp32 d- and g-rows of the chunks kD3 chunks ahead into registers, apply ReLU6(d * scale + shift) (the arithmetic of
//     crnn_pwconv_bnrelu6_wgrad_f32x2's staging waves, bit for bit), split both operands into their planes (common.h crnn_split3_pair) and write them k-major
//     into an LDS ring of 3 stages (2 operands x 2 planes x 32 pixels x 160 bf16: 40 KiB per stage), conflict-free transposing fragment reads
//     (two ds_read_b64_tr_b16 per fragment); a chunk's planes are formed one stage interval before they are written (as gemm_wres3.hip);
//   * g given as planes (crnn_pwconv_bnrelu6_wgrad_planes_stream_gp: what the step runs): the IO waves handle d only, and FOUR loader waves bring the planes of g
//     by LDS-DMA into a ring of their own (6 stages of 2 planes x 32 pixel rows x 256 B, unpadded, 16-byte chunks swizzled on the source address) -- the IO side
//     is this kernel's bound (profiles/r06_wgrad3_ablate.txt);
//   * one raw s_barrier per chunk, branch-free steady state; all tiles of one reduction range sit on one XCD; the fp32 partial tiles go to scratch
//     [range][K][N] and a fixed-order second stage sums them (deterministic).
// Same planes and products as the tile kernel; the reduction is grouped differently (other range boundaries): equal to fp32 summation round-off.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

struct Wg3Params {
  const float* D; const float* G; float* part;        // d [M][lda], g [M][ldg], partial tiles [nsplit][K][N]
  const unsigned short* GP; long gps;                 // GPL: g as bf16 planes [2][M][ldg] (hi, mid: the words of crnn_split3_pair), element stride between them
  const float* scale; const float* shift;             // [K] or null (plain operand)
  int M, N, K;
  int chunks;        // M / 32
  int TI, TJ, nsplit, per, lda, ldg;
  int flat;          // more tiles than an XCD has CUs (dense1's weight gradient: 36): grid = tiles * nsplit, id -> (tile = id % tiles, range = id / tiles), as gemm_wgrad.hip
};

constexpr int kLd3 = 128 + 32;                // bf16 row stride of a k-major operand plane (320 B)
constexpr int kPl3 = 32 * kLd3 * 2;           // bytes of one plane of one operand stage: 32 pixels x 160 x 2 B = 10 KiB
constexpr int kStage3 = 4 * kPl3;             // A hi | A mid | B hi | B mid
#ifndef W3G_RING
#define W3G_RING 3
#define W3G_D 3
#define W3G_WGS 1
#endif
constexpr int kRing3 = W3G_RING;              // LDS stages (2: 80 KiB, two workgroups per CU)
constexpr int kD3 = W3G_D;                    // chunks in flight in the IO waves' registers
constexpr int kLead3 = kRing3 - 1;            // stage stored after barrier s: s + kLead3

// fragment = the 8 bf16 (k = 16 ks + 8 half .. +7) of tile row r0 + l31, from a k-major plane (as gemm_bf16.inc read_frag_h<true>)
__device__ __forceinline__ bf16x8_t wg3_frag(const unsigned char* Xs, int r0, int ks, int half, int l31) {
  const int li = l31 & 15;
  const unsigned short* X = reinterpret_cast<const unsigned short*>(Xs) + (ks * 16 + 8 * half + (li >> 2)) * kLd3 + r0 + (l31 & 16) + (li & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)X);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + 4 * kLd3));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

// ---- GPL (g as planes): the planes of g come by LDS-DMA issued by two more waves that do nothing else (as gemm_wres.hip's loader waves; issued by the MFMA waves
// themselves the 16 pieces of a chunk cost them as much as they saved the IO waves: 205 -> 188 us only, profiles/r06_wgrad3_ablate.txt), into a ring of
// its own: a stage = 2 planes x 32 pixel rows x 256 B (128 channels, unpadded -- the DMA's LDS image is lane-linear), 16-byte chunk c of row r at position
// c ^ ((r & 3) << 2) (on the source address): the eight rows a transposing fragment read touches fall two and two into the four 64-byte bank ranges.
constexpr int kPlB3 = 32 * 256;               // bytes of one g plane of a stage
constexpr int kStageB3 = 2 * kPlB3;           // hi | mid
constexpr int kRingB3 = 6;                    // stages of the g ring (5 in flight per CU: 80 KiB)
constexpr int kLoaders3 = 4;                  // loader waves (one sustains ~25 GB/s per CU -- MI355X_MICROARCH.md, measured here: 314 us with one)
constexpr int kStageA3 = 2 * kPl3;            // GPL: the IO waves' stages hold the planes of a only
__device__ __forceinline__ unsigned wg3_lds_addr(const void* p) { return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p); }
__device__ __forceinline__ void wg3_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// fragment of tile row r0 + l31 from a swizzled, unpadded plane (row = pixel, 128 channels)
__device__ __forceinline__ bf16x8_t wg3_frag_sw(const unsigned char* Xs, int r0, int ks, int half, int l31) {
  const int li = l31 & 15;
  const int row = ks * 16 + 8 * half + (li >> 2), col = r0 + (l31 & 16) + (li & 3) * 4;
  const unsigned short* X = reinterpret_cast<const unsigned short*>(Xs) + row * 128 + ((((col >> 3) ^ ((row & 3) << 2)) << 3) | (col & 7));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)X);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + 4 * 128));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

template <bool XF, bool GPL>   // XF: the A operand is ReLU6(d * scale + shift); GPL: g arrives split (written as planes by the kernel that produced it)
__global__ __launch_bounds__(GPL ? 512 + 64 * kLoaders3 : 512, GPL ? 1 : 2 * W3G_WGS) void pw_wgrad_planes_kernel(Wg3Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kRing3 stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (xcd, local) -> (range, tile): the tiles of a range are neighbours on one XCD
  const int wg = blockIdx.x, x = wg & 7, loc = wg >> 3;
  const int tiles = p.TI * p.TJ;
  const int lin = p.flat ? wg % tiles : loc % tiles, rloc = loc / tiles;
  const int split = p.flat ? wg / tiles : rloc * 8 + x;
  if (split >= p.nsplit) return;
  const int ti = lin / p.TJ, tj = lin % p.TJ;
  const int c0 = split * p.per;
  const int total = min(p.per, p.chunks - c0);                  // chunks of this range (> 0 by the host's choice of nsplit)

  if constexpr (GPL) {
    if (wave >= 8) {
      const int lw = wave - 8;                                    // this loader's share: pieces lw, lw + kLoaders3, ...
      // ---------------------------------------------------------------------- loader wave: the 16 1-KiB pieces of a stage's g planes -- piece pi -> plane pi >> 3,
      // rows 4 (pi & 7) + (lane >> 4), chunk position lane & 15 -- kRingB3 - 1 stages ahead, one counted wait and one barrier per chunk
      const unsigned bring = __builtin_amdgcn_readfirstlane(wg3_lds_addr(smem + kRing3 * kStageA3));
      const unsigned short* gl = p.GP + (long)(lane >> 4) * p.ldg + tj * 128 + 8 * ((lane & 15) ^ (((lane >> 4) & 3) << 2));
      auto issue_b = [&](int st, int sb) __attribute__((always_inline)) {
        const long uo = (long)(c0 + (st < total ? st : total - 1)) * 32 * p.ldg;   // (past the end: the last chunk again, into a slot whose stage has been consumed)
#pragma unroll
        for (int pi = lw; pi < 16; pi += kLoaders3)
          wg3_dma16(gl + uo + (long)(pi >> 3) * p.gps + (long)(4 * (pi & 7)) * p.ldg, bring + sb * kStageB3 + (pi >> 3) * kPlB3 + (pi & 7) * 1024);
      };
#pragma unroll
      for (int s0 = 0; s0 < kRingB3 - 1; ++s0) issue_b(s0, s0);
      int sb = kRingB3 - 1;                                       // slot of the stage issued next = (s - 1) % kRingB3 at iteration s
      for (int s = 0; s < total; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRingB3 - 2) * 16 / kLoaders3) : "memory");   // this wave's pieces of stage s have landed (the later stages' may be in flight)
        __builtin_amdgcn_s_barrier();
        issue_b(s + kRingB3 - 1, sb);
        sb = sb + 1 == kRingB3 ? 0 : sb + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the clamped re-reads past the end: nothing may land after the workgroup has left)
      __builtin_amdgcn_s_barrier();
      return;
    }
  }
  if (wave < 4) {
    // ------------------------------------------------------------------------ MFMA waves
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int slot = 0, slotb = 0;
    constexpr int STG = GPL ? kStageA3 : kStage3;
    for (int s = 0; s < total; ++s) {
      __builtin_amdgcn_s_barrier();                             // stage s is in LDS; stage s-1's slot is released
      const unsigned char* As = smem + slot * STG;
      const unsigned char* Bs = GPL ? smem + kRing3 * kStageA3 + slotb * kStageB3 : As + 2 * kPl3;
      slot = slot + 1 == kRing3 ? 0 : slot + 1;
      slotb = slotb + 1 == kRingB3 ? 0 : slotb + 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fa[2][2], fb[2][2];                            // [plane][block]
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[pl][i] = wg3_frag(As + pl * kPl3, wm * 64 + i * 32, ks, half, l31);
            if constexpr (GPL) fb[pl][i] = wg3_frag_sw(Bs + pl * kPlB3, wn * 64 + i * 32, ks, half, l31);
            else fb[pl][i] = wg3_frag(Bs + pl * kPl3, wn * 64 + i * 32, ks, half, l31);
          }
        // a_mid g_hi, a_hi g_mid, a_hi g_hi (the tile kernel's order: small terms first); consecutive MFMAs on different accumulators
        constexpr int PA[3] = {1, 0, 0}, PG[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[PG[t]][j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_barrier();                               // the IO waves' last barrier
    // partial tile -> scratch: lane = one column, register e = row 8 (e / 4) + 4 half + (e & 3) of the 32 x 32 block
    float* out = p.part + (long)split * p.K * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = ti * 128 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
          const int col = tj * 128 + wn * 64 + j * 32 + l31;
          out[(long)row * p.N + col] = acc[i][j][e];
        }
    return;
  }
  // -------------------------------------------------------------------------- IO waves: 256 lanes, per chunk and operand 32 pixels x 16 pieces of 8 channels
  const int w = wave - 4;
  const int c16 = lane & 15;                                    // 8-channel piece of the 128-channel tile row
  const int pxl = w * 4 + (lane >> 4);                          // pixels pxl, pxl + 16 of the chunk
  float sc[8], sh[8];
  if constexpr (XF) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = p.scale[ti * 128 + c16 * 8 + e]; sh[e] = p.shift[ti * 128 + c16 * 8 + e]; }
  }
  const float* dbase = p.D + (long)ti * 128 + c16 * 8;
  const float* gbase = GPL ? nullptr : p.G + (long)tj * 128 + c16 * 8;
  float4 ra[kD3][2][2], rg[kD3][2][2];                          // [buffer][pixel u][half]
  auto load = [&](int s, float4 (&xa)[2][2], float4 (&xg)[2][2]) {
    s = s < total ? s : total - 1;                              // past the end: a valid address, the data lands in a consumed slot
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long row = (long)(c0 + s) * 32 + pxl + 16 * u;
      const float* pa = dbase + row * p.lda;
      xa[u][0] = *reinterpret_cast<const float4*>(pa); xa[u][1] = *reinterpret_cast<const float4*>(pa + 4);
      if constexpr (GPL) {                                      // (the MFMA waves bring the planes of g by LDS-DMA)
        xg[u][0] = make_float4(0.f, 0.f, 0.f, 0.f); xg[u][1] = xg[u][0];
      } else {
        const float* pg = gbase + row * p.ldg;
        xg[u][0] = *reinterpret_cast<const float4*>(pg); xg[u][1] = *reinterpret_cast<const float4*>(pg + 4);
      }
    }
  };
  u32x4 pw[2][4];                                               // [pixel u][A hi, A mid, B hi, B mid] of the chunk formed last
  auto compute = [&](const float4 (&xa)[2][2], const float4 (&xg)[2][2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float va[8] = {xa[u][0].x, xa[u][0].y, xa[u][0].z, xa[u][0].w, xa[u][1].x, xa[u][1].y, xa[u][1].z, xa[u][1].w};
      const float vg[8] = {xg[u][0].x, xg[u][0].y, xg[u][0].z, xg[u][0].w, xg[u][1].x, xg[u][1].y, xg[u][1].z, xg[u][1].w};
      if constexpr (XF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) va[e] = relu6f(fmaf(va[e], sc[e], sh[e]));
      }
      unsigned wa[3][4], wb[3][4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        crnn_split3_pair(va[2 * pr], va[2 * pr + 1], wa[0][pr], wa[1][pr], wa[2][pr]);
        if constexpr (!GPL) crnn_split3_pair(vg[2 * pr], vg[2 * pr + 1], wb[0][pr], wb[1][pr], wb[2][pr]);
      }
      pw[u][0] = u32x4{wa[0][0], wa[0][1], wa[0][2], wa[0][3]}; pw[u][1] = u32x4{wa[1][0], wa[1][1], wa[1][2], wa[1][3]};
      if constexpr (GPL) { pw[u][2] = u32x4{0u, 0u, 0u, 0u}; pw[u][3] = pw[u][2]; }
      else { pw[u][2] = u32x4{wb[0][0], wb[0][1], wb[0][2], wb[0][3]}; pw[u][3] = u32x4{wb[1][0], wb[1][1], wb[1][2], wb[1][3]}; }
    }
  };
  auto store = [&](int s) {
    unsigned char* st = smem + (s % kRing3) * (GPL ? kStageA3 : kStage3) + c16 * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < (GPL ? 2 : 4); ++q) *reinterpret_cast<u32x4*>(st + q * kPl3 + (pxl + 16 * u) * (kLd3 * 2)) = pw[u][q];
  };
  // barrier s (s = 0 .. total): before it stage s is written; after it the slot of stage s-1 is free -> stage s + kLead3 goes there (its planes were formed before
  // the barrier), then the planes of stage s + kLead3 + 1 are formed and its buffer refilled
  auto step = [&](int s, float4 (&xa)[2][2], float4 (&xg)[2][2]) {   // xa/xg = buffer (s + kLead3 + 1) % kD3
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    store(s + kLead3);
    compute(xa, xg);
    load(s + kLead3 + 1 + kD3, xa, xg);
  };
#pragma unroll
  for (int k = 0; k < kD3; ++k) load(k, ra[k], rg[k]);
#pragma unroll
  for (int k = 0; k < kLead3; ++k) { compute(ra[k % kD3], rg[k % kD3]); store(k); load(kD3 + k, ra[k % kD3], rg[k % kD3]); }
  compute(ra[kLead3 % kD3], rg[kLead3 % kD3]); load(kD3 + kLead3, ra[kLead3 % kD3], rg[kLead3 % kD3]);
  int s = 0;
  for (; s + kD3 <= total; s += kD3) {
#pragma unroll
    for (int k = 0; k < kD3; ++k) step(s + k, ra[(k + kLead3 + 1) % kD3], rg[(k + kLead3 + 1) % kD3]);
  }
#pragma unroll
  for (int k = 0; k < kD3; ++k)
    if (s + k <= total) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + k < total) {
        store(s + k + kLead3); compute(ra[(k + kLead3 + 1) % kD3], rg[(k + kLead3 + 1) % kD3]);
        load(s + k + kLead3 + 1 + kD3, ra[(k + kLead3 + 1) % kD3], rg[(k + kLead3 + 1) % kD3]);
      }
    }
}

// out[i] = sum_s part[s][i], s ascending (deterministic), four partials in flight per thread
__global__ __launch_bounds__(256) void pw_wgrad3_sum_kernel(const float* __restrict__ part, int nsplit, long total, float* __restrict__ out, int N, int ldc) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    const float4 v0 = *reinterpret_cast<const float4*>(part + (long)s * total + i), v1 = *reinterpret_cast<const float4*>(part + (long)(s + 1) * total + i);
    const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(s + 2) * total + i), v3 = *reinterpret_cast<const float4*>(part + (long)(s + 3) * total + i);
    a.x = ((a.x + v0.x) + v1.x) + (v2.x + v3.x); a.y = ((a.y + v0.y) + v1.y) + (v2.y + v3.y);
    a.z = ((a.z + v0.z) + v1.z) + (v2.z + v3.z); a.w = ((a.w + v0.w) + v1.w) + (v2.w + v3.w);
  }
  for (; s < nsplit; ++s) { const float4 v = *reinterpret_cast<const float4*>(part + (long)s * total + i); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  *reinterpret_cast<float4*>(out + (i / N) * ldc + i % N) = a;
}

void wg3_geom(long M, int N, int K, Wg3Params& p, int& grid) {
  p.chunks = (int)(M / 32); p.TI = K / 128; p.TJ = N / 128;
  const int tiles = p.TI * p.TJ;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  int ns = (W3G_WGS * cus / tiles) & ~7;                                  // one workgroup per CU, a multiple of 8 ranges (8 XCDs)
  p.flat = tiles > cus / 8;
  if (p.flat) ns = cus / tiles < 1 ? 1 : cus / tiles;           // (round 6: dense1's 36 feature tiles -- cus / tiles ranges, one workgroup per CU)
  else if (ns < 8) ns = 8;
  while (ns > 8 && p.chunks / ns < 2 * kD3) ns -= 8;            // a range is at least two pipeline depths long
  if (ns > p.chunks) ns = p.chunks;
  p.per = cdiv(p.chunks, ns);
  p.nsplit = cdiv(p.chunks, p.per);                             // drop empty tail ranges
  grid = p.flat ? p.nsplit * tiles : 8 * cdiv(p.nsplit, 8) * tiles;
}

}  // namespace

// 0 if crnn_pwconv_bnrelu6_wgrad_planes_stream handles the shape (whole 32-pixel chunks, K a multiple of 128 up to 8192, N up to 1024, at most 64 tiles), else -3
extern "C" int crnn_pwconv_wgrad_planes_stream_supported(long M, int N, int K) {
  return (M >= 32 && M % 32 == 0 && N >= 128 && N % 128 == 0 && K >= 128 && K % 128 == 0 && N <= 1024 && K <= 8192 && (K / 128) * (N / 128) <= 64 &&
          M * (long)(K > N ? K : N) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" size_t crnn_pwconv_wgrad_planes_stream_scratch_bytes(long M, int N, int K) {
  if (crnn_pwconv_wgrad_planes_stream_supported(M, N, K) != CRNN_OK) return 0;
  Wg3Params p; int grid; wg3_geom(M, N, K, p, grid);
  return (size_t)p.nsplit * K * N * sizeof(float);
}
namespace {
template <bool XF, bool GPL>
int wg3_launch_kernel(const Wg3Params& p, int grid, int lds, hipStream_t stream) {
  CRNN_LDS_ATTR((pw_wgrad_planes_kernel<XF, GPL>), lds);
  hipLaunchKernelGGL((pw_wgrad_planes_kernel<XF, GPL>), dim3(grid), dim3(GPL ? 512 + 64 * kLoaders3 : 512), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
int wg3_run(const float* d, const float* in_bnstate, const float* g, const void* g_planes, long g_plane_stride, float* dw, long M, int N, int K, float* scratch,
            size_t scratch_bytes, hipStream_t stream, int lda = 0, int ldg = 0, int ldc = 0) {
  if (!d || (!g && !g_planes) || !dw || !scratch) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_pwconv_wgrad_planes_stream_supported(M, N, K));
  if ((((uintptr_t)d | (uintptr_t)g | (uintptr_t)g_planes | (uintptr_t)dw | (uintptr_t)scratch | (uintptr_t)in_bnstate) & 15) || (g_plane_stride & 7)) return CRNN_ERR_UNSUPPORTED;
  lda = lda ? lda : K; ldg = ldg ? ldg : N; ldc = ldc ? ldc : N;
  if (lda < K || ldg < N || ldc < N || ((lda | ldg | ldc) & 3) || M * (long)(lda > ldg ? lda : ldg) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  Wg3Params p;
  p.D = d; p.G = g; p.GP = reinterpret_cast<const unsigned short*>(g_planes); p.gps = g_plane_stride; p.part = scratch;
  p.scale = in_bnstate ? in_bnstate + 2L * K : nullptr; p.shift = in_bnstate ? in_bnstate + 3L * K : nullptr;
  p.M = (int)M; p.N = N; p.K = K; p.lda = lda; p.ldg = ldg;
  int grid; wg3_geom(M, N, K, p, grid);
  if ((size_t)p.nsplit * K * N * sizeof(float) > scratch_bytes) return CRNN_ERR_UNSUPPORTED;
  const int lds = g_planes ? kRing3 * kStageA3 + kRingB3 * kStageB3 : kRing3 * kStage3;
  if (in_bnstate) CRNN_TRY(g_planes ? (wg3_launch_kernel<true, true>(p, grid, lds, stream)) : (wg3_launch_kernel<true, false>(p, grid, lds, stream)));
  else CRNN_TRY(g_planes ? (wg3_launch_kernel<false, true>(p, grid, lds, stream)) : (wg3_launch_kernel<false, false>(p, grid, lds, stream)));
  const long total = (long)K * N;
  hipLaunchKernelGGL(pw_wgrad3_sum_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, stream, scratch, p.nsplit, total, dw, N, ldc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
}  // namespace
// dw[K][N] (fp32, row stride N) = ReLU6(d * scale + shift)^T [K][M] . g[M][N]   (two bf16 planes per operand; d [M][K], g [M][N] fp32;
// in_bnstate = [mean|var|scale|shift] x K of the BatchNorm on d, NULL: dw = d^T . g)
extern "C" int crnn_pwconv_bnrelu6_wgrad_planes_stream(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K,
                                                       float* scratch, size_t scratch_bytes, hipStream_t stream) {
  return wg3_run(d, in_bnstate, g, nullptr, 0, dw, M, N, K, scratch, scratch_bytes, stream);
}
// ... with g given as its two planes (g_planes: hi plane [M][N] bf16 words, the mid plane g_plane_stride elements behind it -- crnn_bn_bwd_planes_ex's output):
// the IO waves copy them to LDS instead of splitting fp32 values; the same words, the same result bit for bit
extern "C" int crnn_pwconv_bnrelu6_wgrad_planes_stream_gp(const float* d, const float* in_bnstate, const void* g_planes, long g_plane_stride, float* dw, long M, int N,
                                                          int K, float* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (!g_planes) return CRNN_ERR_ARG;
  return wg3_run(d, in_bnstate, nullptr, g_planes, g_plane_stride, dw, M, N, K, scratch, scratch_bytes, stream);
}
// The same stream without a transform and with leading dimensions (round 6): C[M][N] (fp32, row stride ldc) = A^T . B over the K rows of A [K][lda >= M] and
// B [K][ldb >= N], both fp32, two bf16 planes per operand (hi*hi + hi*mid + mid*hi: the precision of crnn_gemm_f32x2) -- the recurrent layers' weight
// gradients dW = x^T dz, dU = h_prev^T dz of the parity mode (utils.py:77-82 backwards), which the tile kernel ran as 8-16 tiles x 32-64 K ranges plus a
// 64-row reduction.  Supported (else -3): crnn_pwconv_wgrad_planes_stream_supported(K, N, M), leading dimensions multiples of 4, 16-byte aligned pointers;
// scratch: crnn_pwconv_wgrad_planes_stream_scratch_bytes(K, N, M).  Same products as crnn_gemm_f32x2 mode 2, another summation order.
extern "C" int crnn_gemm_tn_planes_stream(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                                          size_t scratch_bytes, hipStream_t stream) {
  return wg3_run(A, nullptr, B, nullptr, 0, C, K, N, M, scratch, scratch_bytes, stream, lda, ldb, ldc);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Input gradient of a Bidirectional recurrent layer's input projections in the parity mode (utils.py:77-82 backwards; round 6):
//     dX[M][N] (fp32) = dZf[M][K] . Wf[N][K]^T + dZb[M][K] . Wb[N][K]^T      M = T*B rows, K = 4u (3u) gate columns, N = 128 | 256
// with two bf16 planes per operand (hi*hi + hi*mid + mid*hi: the precision of crnn_gemm_f32x2) on the schedule of gemm_wgrad.hip's stripe stream
// (gemm_nt_f32_stream_kernel): one workgroup per (64-row stripe, 128-column slab) keeps its [64][128] result in the MFMA waves' registers over the whole
// reduction of both pairs; eight IO waves stage every 64-k chunk of the dZ rows and of the weight rows through registers, split each fp32 value into its
// hi and mid words (crnn_split3_pair: the words every plane kernel forms) and leave both planes in an LDS ring (swizzled 128-byte rows per plane).  The tile
// kernel ran this as two launches (the second accumulating) of 104-208 tiles with a serial chunk chain each: 41-47 us per launch at batch 256.
// Same products as crnn_gemm_f32x2 mode 1, summed small terms first and in the stripe's (rotated) chunk order: fp32 round-off apart.
namespace {
struct Nts2Params {
  const float* A[2]; const float* W[2];     // the (dZ, W) pairs: A [M][lda], W [N][ldw] fp32
  float* Y;
  int M, K, lda, ldw, ldy, npairs, skew;    // 128 columns per workgroup: blockIdx.y selects the slab
};
constexpr int kN2Ring = 3;
__global__ __launch_bounds__(768) void gemm_nt_f32x2_stream_kernel(Nts2Params p) {
  constexpr int N = 128, kX = 64 * 128, kW = N * 128, kPl = kX + kW, kSt = 2 * kPl;   // a stage: plane hi [X rows | W rows], plane mid [X rows | W rows], 64 k each
  constexpr int WP = N * 8 / 512, ND = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * N;
  const int kch = p.K / 64, total = kch * p.npairs;

  if (wave < 4) {
    const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    int slot = 0;
    for (int s = 0; s < total; ++s) {
      __builtin_amdgcn_s_barrier();
      const unsigned char* Xs = smem + slot * kSt + l31 * 128;
      const unsigned char* Ws = smem + slot * kSt + kX + (wave * 32 + l31) * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = ((2 * ks + half) ^ sw) * 16;
        bf16x8_t fx[2][2], fw[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int b = 0; b < 2; ++b) fx[b][pl] = *reinterpret_cast<const bf16x8_t*>(Xs + pl * kPl + b * 32 * 128 + off);
          fw[pl] = *reinterpret_cast<const bf16x8_t*>(Ws + pl * kPl + off);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[0], fx[b][1], acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[1], fx[b][0], acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[0], fx[b][0], acc[b], 0, 0, 0);
        }
      }
      slot = slot + 1 == kN2Ring ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();
    // lane = one row; register group g of a block = columns 8 g + 4 half + 0..3
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* yrow = p.Y + (long)(m0 + 32 * b + l31) * p.ldy + n0 + wave * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(yrow + 8 * g) = make_float4(acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]);
    }
    return;
  }
  // ---------------------------------------------------------------------------- IO waves (8): lane il = 0..511
  const int il = tid - 256;
  const int xr = il >> 3, xc = il & 7;                          // dZ piece: row xr (0..63), 8-k piece xc; weight pieces: rows xr, xr + 64
  const int kc0 = p.skew ? (int)(((long)p.skew * blockIdx.x) % kch) : 0;   // (see gemm_nt_f32_stream_kernel: every workgroup starts its walk at another chunk)
  u32x4 rx[ND][2], rw[ND][WP][2];
  auto load = [&](int s, u32x4 (&ax)[2], u32x4 (&aw)[WP][2]) {
    s = s < total ? s : total - 1;
    const int pr = s / kch;
    int kc = s - pr * kch + kc0; kc = kc >= kch ? kc - kch : kc;
    const float* a = p.A[pr] + (long)(m0 + xr) * p.lda + kc * 64 + xc * 8;
    ax[0] = *reinterpret_cast<const u32x4*>(a); ax[1] = *reinterpret_cast<const u32x4*>(a + 4);
    const float* w = p.W[pr] + kc * 64 + xc * 8;
#pragma unroll
    for (int u = 0; u < WP; ++u) {
      const float* wr = w + (long)(n0 + xr + 64 * u) * p.ldw;
      aw[u][0] = *reinterpret_cast<const u32x4*>(wr); aw[u][1] = *reinterpret_cast<const u32x4*>(wr + 4);
    }
  };
  auto split8 = [](const u32x4& lo4, const u32x4& hi4, u32x4& h, u32x4& m) __attribute__((always_inline)) {
    unsigned a[4], b[4], w2;
    crnn_split3_pair(__uint_as_float(lo4.x), __uint_as_float(lo4.y), a[0], b[0], w2); crnn_split3_pair(__uint_as_float(lo4.z), __uint_as_float(lo4.w), a[1], b[1], w2);
    crnn_split3_pair(__uint_as_float(hi4.x), __uint_as_float(hi4.y), a[2], b[2], w2); crnn_split3_pair(__uint_as_float(hi4.z), __uint_as_float(hi4.w), a[3], b[3], w2);
    h = u32x4{a[0], a[1], a[2], a[3]}; m = u32x4{b[0], b[1], b[2], b[3]};
  };
  auto write = [&](int s, const u32x4 (&ax)[2], const u32x4 (&aw)[WP][2]) {
    unsigned char* st = smem + (s % kN2Ring) * kSt;
    u32x4 h, m;
    split8(ax[0], ax[1], h, m);
    const int xo = xr * 128 + ((xc ^ ((xr >> 1) & 7)) * 16);
    *reinterpret_cast<u32x4*>(st + xo) = h; *reinterpret_cast<u32x4*>(st + kPl + xo) = m;
#pragma unroll
    for (int u = 0; u < WP; ++u) {
      const int r = xr + 64 * u;
      split8(aw[u][0], aw[u][1], h, m);
      const int wo = kX + r * 128 + ((xc ^ ((r >> 1) & 7)) * 16);
      *reinterpret_cast<u32x4*>(st + wo) = h; *reinterpret_cast<u32x4*>(st + kPl + wo) = m;
    }
  };
  auto step = [&](int s, u32x4 (&ax)[2], u32x4 (&aw)[WP][2]) {     // buffer (s + 1) % ND
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write(s + 1, ax, aw); load(s + 1 + ND, ax, aw);
  };
  // barrier s: stage s is written before it; after it the slot of stage s-1 is free; stage s+1 goes into slot (s+1) % 3, which held stage s-2 -- released at barrier s-1
#pragma unroll
  for (int k = 0; k < ND; ++k) load(k, rx[k], rw[k]);
  write(0, rx[0], rw[0]); load(ND, rx[0], rw[0]);
  int s = 0;
  for (; s + ND <= total; s += ND) {
#pragma unroll
    for (int k = 0; k < ND; ++k) step(s + k, rx[(k + 1) % ND], rw[(k + 1) % ND]);
  }
#pragma unroll
  for (int k = 0; k < ND; ++k)
    if (s + k <= total) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + k < total) { write(s + k + 1, rx[(k + 1) % ND], rw[(k + 1) % ND]); load(s + k + 1 + ND, rx[(k + 1) % ND], rw[(k + 1) % ND]); }
    }
}
}  // namespace
// Y[M][N] (fp32, row stride ldy) = A0[M][K] . W0[N][K]^T (+ A1 . W1^T when A1 != NULL), everything fp32, two bf16 planes per operand.  Supported (else -3):
// M % 64 == 0, N % 128 == 0, K % 64 == 0, leading dimensions multiples of 4, 16-byte aligned pointers.
extern "C" int crnn_gemm_nt_f32x2_stream(const float* A0, const float* W0, const float* A1, const float* W1, float* Y, int M, int N, int K, int lda, int ldw,
                                         int ldy, hipStream_t stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !A0 || !W0 || !Y || ((A1 != nullptr) != (W1 != nullptr))) return CRNN_ERR_ARG;
  if (M % 64 || N % 128 || K % 64 || ((lda | ldw | ldy) & 3) || lda < K || ldw < K || ldy < N) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)Y) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)M * (lda > ldy ? lda : ldy) >= (1L << 31) || N > 65535 * 128) return CRNN_ERR_UNSUPPORTED;
  Nts2Params p;
  p.A[0] = A0; p.A[1] = A1 ? A1 : A0; p.W[0] = W0; p.W[1] = W1 ? W1 : W0; p.Y = Y;
  p.M = M; p.K = K; p.lda = lda; p.ldw = ldw; p.ldy = ldy; p.npairs = A1 ? 2 : 1; p.skew = 3;
  const int lds = kN2Ring * 2 * (64 * 128 + 128 * 128);
  CRNN_LDS_ATTR(gemm_nt_f32x2_stream_kernel, lds);
  hipLaunchKernelGGL(gemm_nt_f32x2_stream_kernel, dim3(M / 64, N / 128), dim3(768), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
