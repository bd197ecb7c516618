// Pixel-streaming weight gradient of a pointwise (1x1) convolution whose input is the PRE-BatchNorm depthwise output
// (utils.py:44-49, training backward):
//     dW[K][N] = ReLU6(BN1(d))^T [K][M] . g[M][N]        d, g bf16 (NHWC rows), dW fp32, K = channels in, N = channels out
// The reduction runs over the M = B*H*W pixels (10^5..10^6), the result is at most 512 x 512.  The tile kernel (gemm_bf16.inc,
// mode 2) runs this as output tiles x ~48 reduction ranges with one register-staged k-chunk in flight per workgroup: every
// 64-pixel chunk costs it a full load round trip (2.8 us per chunk for 0.25 us of MFMAs) -- it ingests ~13 B/clk/CU where the
// L2 can deliver three times that.  Here the same decomposition is organised as a stream:
//   * a workgroup (512 threads) owns one 128 x 128 output tile over a contiguous range of 64-pixel chunks; the 4 MFMA waves
//     (2 x 2, 64 x 64 each: 4 accumulator blocks of v_mfma_f32_32x32x16_bf16) keep the tile in registers for the whole range;
//   * the 4 IO waves load the d- and g-rows of the chunks three chunks ahead into registers (96 KiB per CU in flight), apply
//     ReLU6(d * scale + shift) -> bf16 (a lane keeps the scale / shift of its 8 channels for the whole launch; the arithmetic of
//     crnn_pwconv_bnrelu6_wgrad's prologue bit for bit) and write both operands k-major into an LDS ring of 3 stages, row stride
//     128 + 32 bf16 so that the transposing fragment reads (two ds_read_b64_tr_b16 per fragment) are conflict-free;
//   * one raw s_barrier per chunk; the steady-state IO loop is branch-free so the wait-count insertion counts the outstanding
//     loads exactly instead of draining them;
//   * all tiles of one reduction range sit on one XCD (they read the same pixel rows: HBM sees them once); ranges are
//     contiguous, the fp32 partial tiles go to scratch [range][K][N] and a fixed-order second stage sums them (deterministic).
// Same products as the tile kernel (bf16 operands, fp32 accumulation); the reduction is grouped differently (other range
// boundaries), so the result agrees to fp32 summation round-off, not bit for bit.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef CRNN_WG_EXP
#define CRNN_WG_EXP 0     // compile-time ablation mask (scripts/wgrad_bench.py --ablate): 1 no transform, 2 no fragment reads / MFMAs, 4 no global loads
#endif

namespace {

struct WgParams {
  const bf16_t* D; const bf16_t* G; float* part;     // d [M][K], g [M][N], partial tiles [nsplit][K][N]
  const float* scale; const float* shift;             // [K]
  int M, N, K;
  int chunks;        // M / 64
  int TI, TJ;        // output tiles along K (rows) and N (columns)
  int nsplit;        // reduction ranges; grid = 8 * ceil(TI * TJ * nsplit / 8) workgroups, id -> (xcd, tile, range) below
  int per;           // chunks per range (the last range may be shorter)
  int lda, ldg;      // row strides of D and G in elements (bf16 convs: K and N)
  int flat;          // more tiles than an XCD has CUs (round 5, dense1: 36): grid = tiles * nsplit, id -> (tile = id % tiles, range = id / tiles) -- a range's
                     // tiles spread over the XCDs (one XCD would run 32 of them and then the other 4: twice the time, measured)
#ifdef CRNN_WG_TRACE
  unsigned long long* trace;   // [2][64][4] s_memrealtime stamps of workgroup 8: IO wave 4, MFMA wave 0
#endif
};

constexpr int kLd = 128 + 32;                 // bf16 row stride of a k-major operand stage (320 B)
constexpr int kOp = 64 * kLd * 2;             // bytes of one operand stage: 64 pixels x 160 x 2 B = 20 KiB
constexpr int kRing = 3;
constexpr int kD = 3;                         // chunks in flight in the IO waves' registers

__device__ __forceinline__ unsigned wg_bnrelu6_pair(unsigned w, f32x2_t s, f32x2_t t) {
  f32x2_t v;
  v[0] = fma_unpacked(__uint_as_float(w << 16), s[0], t[0]);             // (not v_pk_fma_f32: see common.h)
  v[1] = fma_unpacked(__uint_as_float(w & 0xffff0000u), s[1], t[1]);
  v[0] = __builtin_amdgcn_fmed3f(v[0], 0.f, 6.f); v[1] = __builtin_amdgcn_fmed3f(v[1], 0.f, 6.f);
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// fragment = the 8 bf16 (k = 16 ks + 8 half .. +7) of tile row r0 + l31, from a k-major stage (as gemm_bf16.inc read_frag_h<true>)
__device__ __forceinline__ bf16x8_t wg_frag(const unsigned char* Xs, int r0, int ks, int half, int l31) {
  const int li = l31 & 15;
  const unsigned short* X = reinterpret_cast<const unsigned short*>(Xs) + (ks * 16 + 8 * half + (li >> 2)) * kLd + r0 + (l31 & 16) + (li & 3) * 4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)X);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + 4 * kLd));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

// ---- GDMA (round 6, bf16 operands): g is only copied -- its stages come by LDS-DMA from four loader waves into a ring of their own instead of through the IO waves'
// registers (as gemm_wgrad3.hip, where taking g off the IO waves was worth 11 %): a stage = 64 pixel rows x 256 B (128 channels, unpadded: the DMA's LDS image is
// lane-linear), 16-byte chunk c of row r at position c ^ ((r & 3) << 2) on the source address, so that the transposing fragment reads stay conflict-free.
constexpr int kOpB = 64 * 256;                // bytes of one g stage
#ifndef WG_RINGB
#define WG_RINGB 6
#endif
#ifndef WG_LOADERS
#define WG_LOADERS 4
#endif
#ifndef WG_GDMA
#define WG_GDMA 1
#endif
constexpr int kRingB = WG_RINGB;                     // stages of the g ring (5 in flight per CU: 80 KiB)
constexpr int kLoaders = WG_LOADERS;                   // loader waves (one sustains ~25 GB/s per CU)
__device__ __forceinline__ unsigned wg_lds_addr(const void* p) { return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p); }
__device__ __forceinline__ void wg_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ bf16x8_t wg_frag_sw(const unsigned char* Xs, int r0, int ks, int half, int l31) {
  const int li = l31 & 15;
  const int row = ks * 16 + 8 * half + (li >> 2), col = r0 + (l31 & 16) + (li & 3) * 4;
  const unsigned short* X = reinterpret_cast<const unsigned short*>(Xs) + row * 128 + ((((col >> 3) ^ ((row & 3) << 2)) << 3) | (col & 7));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)X);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + 4 * 128));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

// F32: both operands are fp32 tensors rounded to bf16 on the way in (v_cvt_pk_bf16_f32, RNE -- what the tile GEMM does while it stages
// them), no BatchNorm transform: the weight gradients of the recurrent layers, dW = X^T dZ and dU = H^T dZ over the T*B rows
// XF = false (bf16 operands): no BatchNorm transform either -- round 5, dense1's weight gradient dW1 = x7^T gbm over bf16 tensors
template <int NIO, bool F32, bool XF = true, bool GDMA = false>   // IO waves (4 or 8): 16 / NIO 8-channel pieces of each operand per lane and chunk; GDMA: g by loader waves
__global__ __launch_bounds__(256 + 64 * NIO + (GDMA ? 64 * kLoaders : 0)) void pw_wgrad_stream_kernel(WgParams p) {
  static_assert(!GDMA || !F32, "the loader waves copy bf16 rows");
  constexpr int STA = GDMA ? kOp : 2 * kOp;                     // bytes of an IO-wave stage (GDMA: the a operand only)
  constexpr int NP = 16 / NIO, RW = F32 ? 2 : 1;              // 16-byte registers per piece
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kRing x (A stage | B stage)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (xcd, local) -> (range, tile): the tiles of a range are neighbours on one XCD
  const int wg = blockIdx.x, x = wg & 7, loc = wg >> 3;
  const int tiles = p.TI * p.TJ;
  const int lin = p.flat ? wg % tiles : loc % tiles, rloc = loc / tiles;   // tile, range index within this XCD
  const int split = p.flat ? wg / tiles : rloc * 8 + x;
  if (split >= p.nsplit) return;
  const int ti = lin / p.TJ, tj = lin % p.TJ;
  const int c0 = split * p.per;
  const int total = min(p.per, p.chunks - c0);                  // chunks of this range (> 0 by the host's choice of nsplit)

  if constexpr (GDMA) {
    if (wave >= 4 + NIO) {
      // ---------------------------------------------------------------------- loader waves: the 16 1-KiB pieces of a g stage (piece pi = rows 4 pi + (lane >> 4), chunk
      // position lane & 15), a quarter each, kRingB - 1 stages ahead; one counted wait and one barrier per chunk
      const int lw = wave - (4 + NIO);
      const unsigned bring = __builtin_amdgcn_readfirstlane(wg_lds_addr(smem + kRing * kOp));
      const bf16_t* gl = p.G + (long)(lane >> 4) * p.ldg + tj * 128 + 8 * ((lane & 15) ^ (((lane >> 4) & 3) << 2));
      auto issue_b = [&](int st, int sb) __attribute__((always_inline)) {
        const long uo = (long)(c0 + (st < total ? st : total - 1)) * 64 * p.ldg;   // (past the end: the last chunk again, into a slot whose stage has been consumed)
#pragma unroll
        for (int q = 0; q < 16 / kLoaders; ++q) wg_dma16(gl + uo + (long)(4 * (lw + q * kLoaders)) * p.ldg, bring + sb * kOpB + (lw + q * kLoaders) * 1024);
      };
#pragma unroll
      for (int s0 = 0; s0 < kRingB - 1; ++s0) issue_b(s0, s0);
      int sb = kRingB - 1;
      for (int s = 0; s < total; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRingB - 2) * 16 / kLoaders) : "memory");   // this wave's pieces of stage s have landed
        __builtin_amdgcn_s_barrier();
        issue_b(s + kRingB - 1, sb);
        sb = sb + 1 == kRingB ? 0 : sb + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing may land after the workgroup has left
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
    for (int s = 0; s < total; ++s) {
#ifdef CRNN_WG_TRACE
      const bool tr = p.trace && wg == 8 && wave == 0 && lane == 0 && s < 64;
      if (tr) p.trace[256 + s * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
      __builtin_amdgcn_s_barrier();                             // stage s is in LDS; stage s-1's slot is released
#ifdef CRNN_WG_TRACE
      if (tr) p.trace[256 + s * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
      const unsigned char* As = smem + slot * STA;
      const unsigned char* Bs = GDMA ? smem + kRing * kOp + slotb * kOpB : As + kOp;
      slot = slot + 1 == kRing ? 0 : slot + 1;
      slotb = slotb + 1 == kRingB ? 0 : slotb + 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (CRNN_WG_EXP & 2) continue;
        bf16x8_t fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = wg_frag(As, wm * 64 + i * 32, ks, half, l31);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = GDMA ? wg_frag_sw(Bs, wn * 64 + j * 32, ks, half, l31) : wg_frag(Bs, wn * 64 + j * 32, ks, half, l31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
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
  // -------------------------------------------------------------------------- IO waves
  const int w = wave - 4;
  const int c16 = lane & 15;                                    // 16-byte piece of a 256-byte tile row: channels 8 c16 .. +7
  const int pxl = w * 4 + (lane >> 4);                          // pixel row within a group of 4 NIO (chunk = NP groups)
  f32x2_t sc[4], sh[4];
  if constexpr (!F32 && XF) {
    const float* s = p.scale + ti * 128 + c16 * 8; const float* t = p.shift + ti * 128 + c16 * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = f32x2_t{s[2 * e], s[2 * e + 1]}; sh[e] = f32x2_t{t[2 * e], t[2 * e + 1]}; }
  }
  const bf16_t* dbase = p.D + (F32 ? 2L : 1L) * ((long)ti * 128 + c16 * 8);    // (fp32 operands: the same pointers, 4-byte elements)
  const bf16_t* gbase = p.G + (F32 ? 2L : 1L) * ((long)tj * 128 + c16 * 8);
  u32x4 ra[kD][NP * RW], rg[kD][NP * RW];
  auto load1 = [&](int s, int u, u32x4* xa, u32x4* xg) {       // RW registers each
    s = s < total ? s : total - 1;                              // past the end: a valid address, the data lands in a consumed slot
    const long row = (long)(c0 + s) * 64 + pxl + 4 * NIO * u;
    if (CRNN_WG_EXP & 4) { xa[0] = u32x4{(unsigned)s, 0u, 0u, 0u}; xg[0] = xa[0]; if (F32) { xa[RW - 1] = xa[0]; xg[RW - 1] = xa[0]; } return; }
    if constexpr (F32) {
      const float* pa = reinterpret_cast<const float*>(dbase) + row * p.lda; const float* pg = reinterpret_cast<const float*>(gbase) + row * p.ldg;
      xa[0] = *reinterpret_cast<const u32x4*>(pa); xa[1] = *reinterpret_cast<const u32x4*>(pa + 4);
      xg[0] = *reinterpret_cast<const u32x4*>(pg); xg[1] = *reinterpret_cast<const u32x4*>(pg + 4);
    } else {
      xa[0] = *reinterpret_cast<const u32x4*>(dbase + row * p.lda);
      if constexpr (GDMA) xg[0] = u32x4{0u, 0u, 0u, 0u}; else xg[0] = *reinterpret_cast<const u32x4*>(gbase + row * p.ldg);
    }
  };
  auto load = [&](int s, u32x4 (&xa)[NP * RW], u32x4 (&xg)[NP * RW]) {
#pragma unroll
    for (int u = 0; u < NP; ++u) load1(s, u, &xa[u * RW], &xg[u * RW]);
  };
  auto round8 = [](const u32x4& lo, const u32x4& hi) {          // 8 fp32 -> 8 bf16
    return u32x4{pack2_bf16(__uint_as_float(lo.x), __uint_as_float(lo.y)), pack2_bf16(__uint_as_float(lo.z), __uint_as_float(lo.w)),
                 pack2_bf16(__uint_as_float(hi.x), __uint_as_float(hi.y)), pack2_bf16(__uint_as_float(hi.z), __uint_as_float(hi.w))};
  };
  auto write1 = [&](int s, int u, const u32x4* xa, const u32x4* xg) {
    unsigned char* As = smem + (s % kRing) * STA;
    unsigned char* Bs = As + kOp;
    const int px = pxl + 4 * NIO * u;
    u32x4 o, og;
    if constexpr (F32) { o = round8(xa[0], xa[1]); og = round8(xg[0], xg[1]); }
    else {
      og = xg[0];
      if ((CRNN_WG_EXP & 1) || !XF) o = xa[0]; else {
        o.x = wg_bnrelu6_pair(xa[0].x, sc[0], sh[0]); o.y = wg_bnrelu6_pair(xa[0].y, sc[1], sh[1]);
        o.z = wg_bnrelu6_pair(xa[0].z, sc[2], sh[2]); o.w = wg_bnrelu6_pair(xa[0].w, sc[3], sh[3]);
      }
    }
    *reinterpret_cast<u32x4*>(As + px * (kLd * 2) + c16 * 16) = o;
    if constexpr (!GDMA) *reinterpret_cast<u32x4*>(Bs + px * (kLd * 2) + c16 * 16) = og;
  };
  auto write = [&](int s, const u32x4 (&xa)[NP * RW], const u32x4 (&xg)[NP * RW]) {
#pragma unroll
    for (int u = 0; u < NP; ++u) write1(s, u, &xa[u * RW], &xg[u * RW]);
  };
  // one chunk piece at a time: transform + stage it, then refill its registers -- the two loads of a piece go out between the
  // vector work of two pieces instead of eight in a burst (a burst stalls the wave at the texture-address queue for ~0.4 us)
  auto write_load = [&](int sw, int sl, u32x4 (&xa)[NP * RW], u32x4 (&xg)[NP * RW]) {
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      write1(sw, u, &xa[u * RW], &xg[u * RW]);
      __builtin_amdgcn_sched_barrier(0);
      load1(sl, u, &xa[u * RW], &xg[u * RW]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // barrier s (s = 0 .. total): before it stage s is written; after it the slot of stage s-1 is free -> stage s+2 goes there
  auto step = [&](int s, u32x4 (&xa)[NP * RW], u32x4 (&xg)[NP * RW]) {   // xa/xg = buffer (s + 2) % kD
#ifdef CRNN_WG_TRACE
    const bool tr = p.trace && wg == 8 && w == 0 && lane == 0 && s < 64;
    if (tr) p.trace[s * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef CRNN_WG_TRACE
    if (tr) p.trace[s * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef CRNN_WG_TRACE
    if (tr) p.trace[s * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    write_load(s + 2, s + 2 + kD, xa, xg);
#ifdef CRNN_WG_TRACE
    if (tr) p.trace[s * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
  };
#pragma unroll
  for (int k = 0; k < kD; ++k) load(k, ra[k], rg[k]);
  write(0, ra[0], rg[0]); load(kD, ra[0], rg[0]);
  write(1, ra[1], rg[1]); load(kD + 1, ra[1], rg[1]);
  int s = 0;
  for (; s + kD <= total; s += kD) {
#pragma unroll
    for (int k = 0; k < kD; ++k) step(s + k, ra[(k + 2) % kD], rg[(k + 2) % kD]);
  }
#pragma unroll
  for (int k = 0; k < kD; ++k)
    if (s + k <= total) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + k < total) { write(s + k + 2, ra[(k + 2) % kD], rg[(k + 2) % kD]); load(s + k + 2 + kD, ra[(k + 2) % kD], rg[(k + 2) % kD]); }
    }
}

// out[i] = sum_s part[s][i] in a fixed grouping (deterministic): 8 split-lanes per output quad (lane l takes s = l, l + 8, ... with four
// loads in flight), combined through LDS in lane order.  32 quads per workgroup: 256 workgroups for a 128 x 256 gradient -- the partials are
// ingested by the whole chip instead of by 32 CUs.
__device__ __forceinline__ void pw_wgrad_sum_block(const float* __restrict__ part, int nsplit, long total, float* __restrict__ out, int N, int ldc, int blk) {
  __shared__ float4 red[8][32];
  const int q = threadIdx.x & 31, l = threadIdx.x >> 5;
  const long i = ((long)blk * 32 + q) * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (i < total) {
    int s = l;
    for (; s + 24 < nsplit; s += 32) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (long)s * total + i), v1 = *reinterpret_cast<const float4*>(part + (long)(s + 8) * total + i);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(s + 16) * total + i), v3 = *reinterpret_cast<const float4*>(part + (long)(s + 24) * total + i);
      a0.x += v0.x + v2.x; a0.y += v0.y + v2.y; a0.z += v0.z + v2.z; a0.w += v0.w + v2.w;
      a1.x += v1.x + v3.x; a1.y += v1.y + v3.y; a1.z += v1.z + v3.z; a1.w += v1.w + v3.w;
    }
    for (; s < nsplit; s += 8) { const float4 v0 = *reinterpret_cast<const float4*>(part + (long)s * total + i); a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
  }
  red[l][q] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (l == 0 && i < total) {
    float4 r = red[0][q];
#pragma unroll
    for (int k = 1; k < 8; ++k) { const float4 v = red[k][q]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    *reinterpret_cast<float4*>(out + (i / N) * ldc + i % N) = r;
  }
}
__global__ __launch_bounds__(256) void pw_wgrad_sum_kernel(const float* __restrict__ part, int nsplit, long total, float* __restrict__ out, int N, int ldc) {
  pw_wgrad_sum_block(part, nsplit, total, out, N, ldc, blockIdx.x);
}
// the second stages of several weight-gradient streams in ONE launch (crnn_wgrad_sum_batch): a workgroup finds its job by its block
// range and runs the very code above on it -- same sums, same order, 5 us of dependent launch per job less
struct SumBatch { crnn_sum_job job[CRNN_SUM_BATCH_MAX]; int first[CRNN_SUM_BATCH_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void pw_wgrad_sum_batch_kernel(SumBatch b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
  const crnn_sum_job& jb = b.job[j];
  pw_wgrad_sum_block(jb.partials, jb.nsplit, jb.total, jb.out, jb.N, jb.ldc, blockIdx.x - b.first[j]);
}

void wg_geom(long M, int N, int K, WgParams& p, int& grid) {
  p.chunks = (int)(M / 64); p.TI = K / 128; p.TJ = N / 128;
  const int tiles = p.TI * p.TJ;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  int ns = (cus / tiles) & ~7;                                  // one workgroup per CU, a multiple of 8 ranges (8 XCDs)
  p.flat = tiles > cus / 8;
  if (p.flat) ns = cus / tiles < 1 ? 1 : cus / tiles;           // more tiles than an XCD has CUs (dense1: 36): cus / tiles ranges, one workgroup per CU
  else if (ns < 8) ns = 8;
  while (ns > 8 && p.chunks / ns < 2 * kD) ns -= 8;             // a range is at least two pipeline depths long
  if (ns > p.chunks) ns = p.chunks;                             // (tiny inputs: ranges of one chunk; some of the 8 XCD lanes stay empty)
  p.nsplit = ns;
  p.per = cdiv(p.chunks, ns);
  p.nsplit = cdiv(p.chunks, p.per);                             // drop empty tail ranges
  grid = p.flat ? p.nsplit * tiles : 8 * cdiv(p.nsplit, 8) * tiles;
}

}  // namespace

// 0 if crnn_pwconv_bnrelu6_wgrad_stream handles the shape (whole 64-pixel chunks, K and N multiples of 128 up to 1024), else -3
extern "C" int crnn_pwconv_wgrad_stream_supported(long M, int N, int K) {
  return (M >= 64 && M % 64 == 0 && N >= 128 && N % 128 == 0 && K >= 128 && K % 128 == 0 && N <= 1024 && K <= 1024 &&
          M * (long)(K > N ? K : N) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
// bytes of scratch the call needs (the partial tiles)
extern "C" size_t crnn_pwconv_wgrad_stream_scratch_bytes(long M, int N, int K) {
  if (crnn_pwconv_wgrad_stream_supported(M, N, K) != CRNN_OK) return 0;
  WgParams p; int grid; wg_geom(M, N, K, p, grid);
  return (size_t)p.nsplit * K * N * sizeof(float);
}
namespace {
template <bool F32, bool XF = true>
int wg_launch(WgParams& p, long M, float* out, int ldc, float* scratch, size_t scratch_bytes, hipStream_t stream, crnn_sum_job* defer = nullptr) {
  int grid;
  wg_geom(M, p.N, p.K, p, grid);
#ifdef CRNN_WG_TRACE
  { const char* e = getenv("CRNN_WG_TRACE"); p.trace = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  if ((size_t)p.nsplit * p.K * p.N * sizeof(float) > scratch_bytes) return CRNN_ERR_UNSUPPORTED;
  const int lds = kRing * 2 * kOp;
  bool done = false;
  if constexpr (!F32 && XF) {   // the pointwise convolutions' form (bf16 d with the BatchNorm transform, bf16 g): g by LDS-DMA from loader waves
    if (crnn_knob("CRNN_WG_GDMA", WG_GDMA) && (p.ldg & 7) == 0 && (((uintptr_t)p.G) & 15) == 0) {
      const int ldsd = kRing * kOp + kRingB * kOpB;
      CRNN_LDS_ATTR((pw_wgrad_stream_kernel<8, false, true, true>), ldsd);
      hipLaunchKernelGGL((pw_wgrad_stream_kernel<8, false, true, true>), dim3(grid), dim3(256 + 64 * 8 + 64 * kLoaders), ldsd, stream, p);
      done = true;
    }
  }
  if (!done) {
  CRNN_LDS_ATTR((pw_wgrad_stream_kernel<4, F32, XF>), lds);
  CRNN_LDS_ATTR((pw_wgrad_stream_kernel<8, F32, XF>), lds);
  if (crnn_knob("CRNN_WG_NIO", 8) == 4) hipLaunchKernelGGL((pw_wgrad_stream_kernel<4, F32, XF>), dim3(grid), dim3(512), lds, stream, p);
  else hipLaunchKernelGGL((pw_wgrad_stream_kernel<8, F32, XF>), dim3(grid), dim3(768), lds, stream, p);
  }
  CRNN_LAUNCH_CHECK();
  const long total = (long)p.K * p.N;
  if (defer) {   // the caller batches the second stage (crnn_wgrad_sum_batch); `scratch` must stay untouched until then
    defer->partials = scratch; defer->nsplit = p.nsplit; defer->total = total; defer->out = out; defer->N = p.N; defer->ldc = ldc;
    return CRNN_OK;
  }
  hipLaunchKernelGGL(pw_wgrad_sum_kernel, dim3(cdiv(total, 128)), dim3(256), 0, stream, scratch, p.nsplit, total, out, p.N, ldc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
}  // namespace

// Second stages of up to CRNN_SUM_BATCH_MAX deferred weight-gradient streams (crnn_pwconv_bnrelu6_wgrad_stream_defer / crnn_gemm_tn_stream_defer)
// in one launch; results bit-identical to the immediate forms.
extern "C" int crnn_wgrad_sum_batch(const crnn_sum_job* jobs, int n, hipStream_t stream) {
  if (n <= 0) return CRNN_OK;
  if (!jobs || n > CRNN_SUM_BATCH_MAX) return CRNN_ERR_ARG;
  SumBatch b; b.n = n; int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!jobs[i].partials || !jobs[i].out || jobs[i].total <= 0 || (jobs[i].total & 3) || jobs[i].nsplit < 1) return CRNN_ERR_ARG;
    b.job[i] = jobs[i]; b.first[i] = blocks; blocks += cdiv(jobs[i].total, 128);
  }
  for (int i = n; i < CRNN_SUM_BATCH_MAX; ++i) { b.job[i] = jobs[0]; b.first[i] = blocks; }
  b.first[CRNN_SUM_BATCH_MAX] = blocks;
  hipLaunchKernelGGL(pw_wgrad_sum_batch_kernel, dim3(blocks), dim3(256), 0, stream, b);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// dw[K][N] (fp32) = ReLU6(d * scale + shift)^T [K][M] . g[M][N]; d, g bf16; in_bnstate = [mean|var|scale|shift] of the BatchNorm on d
// `defer` (may be NULL): only the first stage runs, *defer describes the second one for crnn_wgrad_sum_batch (scratch stays live until then)
extern "C" int crnn_pwconv_bnrelu6_wgrad_stream_defer(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K,
                                                      float* scratch, size_t scratch_bytes, crnn_sum_job* defer, hipStream_t stream) {
  if (!in_bnstate || !scratch) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_pwconv_wgrad_stream_supported(M, N, K));
  if ((((uintptr_t)d | (uintptr_t)g | (uintptr_t)dw | (uintptr_t)scratch) & 15)) return CRNN_ERR_UNSUPPORTED;
  WgParams p;
  p.D = (const bf16_t*)d; p.G = (const bf16_t*)g; p.part = scratch; p.scale = in_bnstate + 2L * K; p.shift = in_bnstate + 3L * K;
  p.M = (int)M; p.N = N; p.K = K; p.lda = K; p.ldg = N;
  return wg_launch<false>(p, M, dw, N, scratch, scratch_bytes, stream, defer);
}
extern "C" int crnn_pwconv_bnrelu6_wgrad_stream(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K,
                                                float* scratch, size_t scratch_bytes, hipStream_t stream) {
  return crnn_pwconv_bnrelu6_wgrad_stream_defer(d, in_bnstate, g, dw, M, N, K, scratch, scratch_bytes, nullptr, stream);
}

// The same stream for fp32 operands (rounded to bf16 on the way in, like crnn_gemm_bf16_ex mode 2 with fp32 A and B):
//     C[M][N] (fp32, row stride ldc) = A^T . B,   A [K][lda >= M] fp32, B [K][ldb >= N] fp32, K = the reduction over rows.
// The weight gradients of the recurrent layers (dW = X^T dZ, dU = H^T dZ over T*B rows).  Supported (else -3): M, N multiples of
// 128 up to 1024, K a multiple of 64, lda / ldb / ldc multiples of 4, 16-byte aligned pointers; scratch as above.
extern "C" int crnn_gemm_tn_stream_defer(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                                         size_t scratch_bytes, crnn_sum_job* defer, hipStream_t stream) {
  if (!scratch) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_pwconv_wgrad_stream_supported(K, N, M));
  if (lda < M || ldb < N || ldc < N || ((lda | ldb | ldc) & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)scratch) & 15)) return CRNN_ERR_UNSUPPORTED;
  if (K * (long)(lda > ldb ? lda : ldb) >= (1L << 30)) return CRNN_ERR_UNSUPPORTED;
  WgParams p;
  p.D = (const bf16_t*)A; p.G = (const bf16_t*)B; p.part = scratch; p.scale = nullptr; p.shift = nullptr;
  p.M = (int)K; p.N = N; p.K = M; p.lda = lda; p.ldg = ldb;
  return wg_launch<true>(p, K, C, ldc, scratch, scratch_bytes, stream, defer);
}
extern "C" int crnn_gemm_tn_stream(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                                   size_t scratch_bytes, hipStream_t stream) {
  return crnn_gemm_tn_stream_defer(A, lda, B, ldb, C, ldc, M, N, K, scratch, scratch_bytes, nullptr, stream);
}

// The same stream for bf16 operands without a transform (round 5): C[M][N] (fp32, row stride ldc) = A^T . B, A [K][lda >= M] bf16, B [K][ldb >= N]
// bf16, the reduction over the K rows -- dense1's weight gradient dW1 [feat][tds] = x7^T . gbm over T*B rows (M = feat = 4608: 36 tiles of 128
// features, cus / 36 = 7 row ranges).  Supported (else -3): M % 128 == 0 up to 8192, N % 128 == 0 up to 1024, K % 64 == 0, leading dimensions
// multiples of 8, 16-byte aligned pointers; scratch: crnn_gemm_tn_bf16_stream_scratch_bytes.
extern "C" int crnn_gemm_tn_bf16_stream_supported(int M, int N, long K) {
  return (K >= 64 && K % 64 == 0 && N >= 128 && N % 128 == 0 && M >= 128 && M % 128 == 0 && N <= 1024 && M <= 8192 &&
          K * (long)(M > N ? M : N) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" size_t crnn_gemm_tn_bf16_stream_scratch_bytes(int M, int N, long K) {
  if (crnn_gemm_tn_bf16_stream_supported(M, N, K) != CRNN_OK) return 0;
  WgParams p; int grid; wg_geom(K, N, M, p, grid);
  return (size_t)p.nsplit * M * N * sizeof(float);
}
extern "C" int crnn_gemm_tn_bf16_stream(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, long K, float* scratch,
                                        size_t scratch_bytes, hipStream_t stream) {
  if (!A || !B || !C || !scratch) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_gemm_tn_bf16_stream_supported(M, N, K));
  if (lda < M || ldb < N || ldc < N || ((lda | ldb) & 7) || (ldc & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)scratch) & 15)) return CRNN_ERR_UNSUPPORTED;
  if (K * (long)(lda > ldb ? lda : ldb) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  WgParams p;
  p.D = (const bf16_t*)A; p.G = (const bf16_t*)B; p.part = scratch; p.scale = nullptr; p.shift = nullptr;
  p.M = (int)K; p.N = N; p.K = M; p.lda = lda; p.ldg = ldb;
  return wg_launch<false, false>(p, K, C, ldc, scratch, scratch_bytes, stream, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Input gradient of a Bidirectional recurrent layer's input projections (utils.py:77-82, backward):
//     dX[M][N] (fp32) = dZf[M][K] . Wf[N][K]^T + dZb[M][K] . Wb[N][K]^T      M = T*B rows, K = 4u (3u) gate columns, N = 128 | 256
// dZ are fp32 tensors (rounded to bf16 on the way in, as the tile GEMM does), W the bf16 weight shadows.  The tile GEMM runs this as
// two launches (the second accumulating into the first's result) of 104-208 tiles with a serial k-chunk chain each (38-40 us per
// launch against an 11 us HBM floor).  Here one workgroup per 64-row stripe keeps its [64][N] result in the MFMA waves' registers over
// the whole 2K reduction; 8 IO waves stage every 64-k chunk of the dZ rows (fp32 -> bf16) and of the weight rows (bf16) two chunks
// ahead through registers into an LDS ring (swizzled 128-byte rows, the weights-resident kernels' fragment layout), branch-free
// steady state; weights are the MFMA A operand so a lane ends up with 4 consecutive output columns of one row: float4 stores.
namespace {

struct NtsParams {
  const float* A[2]; const bf16_t* W[2];    // the (dZ, W) pairs
  float* Y;
  const float* bias;                        // optional [N total]: added to the finished sums (fp32), as the tile GEMM's epilogue does
  int M, N, K, lda, ldw, ldy, npairs;       // N = columns per workgroup (128 CB); blockIdx.y selects the column slab
  // dense1's epilogue (ABF instantiation, round 5): ReLU, the rows written in permuted order out_row = (m % permP) * (M / permP) + m / permP (batch-major
  // rows to time-major, as crnn_gemm_f32's permP; 0: none), Dropout of site `layer` over the compact [M][ldy] index space of the OUTPUT rows
  int relu = 0, permP = 0; float drop_rate = 0.f; uint64_t seed = 0; uint32_t layer = 0;
  // skew != 0: workgroup i walks the 64-k chunks of the reduction starting at chunk (skew * i) % (K / 64) -- see the kernel
  int skew = 0;
};
constexpr int kNtsRing = 3, kNtsD = 2;

// ABF: the A operand is a bf16 tensor (p.A reinterpreted; no rounding on the way in) and the epilogue is dense1's (bias, ReLU, row permutation, Dropout)
template <int CB, bool ABF = false, int KC = 1>   // 32-channel blocks per MFMA wave: N = 128 CB; KC 64-k chunks per stage (and barrier)
__global__ __launch_bounds__(768) void gemm_nt_f32_stream_kernel(NtsParams p) {
  constexpr int N = 128 * CB;
  constexpr int kX = 64 * 128, kW = N * 128, kCh = kX + kW, kSt = KC * kCh;   // bytes of a chunk: 64 rows x 64 k bf16 | N weight rows x 64 k bf16; a stage = KC chunks
  constexpr int WP = N * 8 / 512;                              // 16-byte weight pieces per IO lane and chunk (N rows x 8 pieces over 512 lanes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * N;          // row stripe, column slab (weight rows n0 .. n0 + N - 1)
  const int kch = p.K / 64, total = kch * p.npairs / KC;        // stages (the host picks KC = 2 only for an even number of chunks)
  // stages a lane has in flight.  (dense1, 72 chunks per stripe: depth 4 instead of 2 changed nothing -- the chunk loop is bound by the barrier round
  // trip of its twelve waves, 0.5-0.6 us per chunk whatever the number of busy CUs; two chunks per barrier, KC = 2, is what shortens it.)
  constexpr int ND = KC > 1 ? 2 : (ABF ? 4 : kNtsD);

  if (wave < 4) {
    const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
    f32x16 acc[2][CB];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][cb][e] = 0.f;
    int slot = 0;
    for (int s = 0; s < total; ++s) {
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const unsigned char* Xs = smem + slot * kSt + j * kCh + l31 * 128;
        const unsigned char* Ws = smem + slot * kSt + j * kCh + kX + (wave * 32 * CB + l31) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int off = ((2 * ks + half) ^ sw) * 16;
          bf16x8_t fx[2], fw[CB];
#pragma unroll
          for (int b = 0; b < 2; ++b) fx[b] = *reinterpret_cast<const bf16x8_t*>(Xs + b * 32 * 128 + off);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) fw[cb] = *reinterpret_cast<const bf16x8_t*>(Ws + cb * 32 * 128 + off);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[b][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cb], fx[b], acc[b][cb], 0, 0, 0);
        }
      }
      slot = slot + 1 == kNtsRing ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();
    // lane = one row; register group g of a block = columns 8 g + 4 half + 0..3
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col0 = n0 + wave * 32 * CB + 4 * half;
      long orow = m0 + 32 * b + l31;
      if (ABF && p.permP) orow = (orow % p.permP) * (p.M / p.permP) + orow / p.permP;
      float* yrow = p.Y + orow * p.ldy + col0;
      const float inv_keep = (ABF && p.drop_rate > 0.f) ? 1.f / (1.f - p.drop_rate) : 1.f;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(acc[b][cb][4 * g], acc[b][cb][4 * g + 1], acc[b][cb][4 * g + 2], acc[b][cb][4 * g + 3]);
          if (p.bias) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + col0 + 32 * cb + 8 * g); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
          if constexpr (ABF) {
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (p.drop_rate > 0.f) {          // the multipliers crnn_dropout applies to elements orow * ldy + col .. + 3 (ldy == the row length there)
              float dm[4];
              drop_scale_vec<4>(p.seed, p.layer, (uint64_t)(orow * p.ldy + col0 + 32 * cb + 8 * g), p.drop_rate, inv_keep, dm);
              v.x *= dm[0]; v.y *= dm[1]; v.z *= dm[2]; v.w *= dm[3];
            }
          }
          *reinterpret_cast<float4*>(yrow + 32 * cb + 8 * g) = v;
        }
    }
    return;
  }
  // ---------------------------------------------------------------------------- IO waves (8): lane il = 0..511
  const int il = tid - 256;
  const int xr = il >> 3, xc = il & 7;                          // dZ piece: row xr (0..63), 8-k piece xc
  // All workgroups start together and advance in step, and the rows of both operands are a multiple of 256 bytes apart (x7: 9216, W1^T: 9216, the
  // recurrent weights: 2048): at any moment every workgroup asks the SAME few memory channels for chunk kc of its rows (0.6 us per chunk whatever the
  // ring depth, the chunks per barrier or the number of busy CUs).  With a skew each workgroup starts its walk over the reduction at another chunk:
  // the chip reads all chunks of the rows at once.  (The sums of a stripe then run in a rotated k order: deterministic, fp32 round-off apart.)
  const int kc0 = p.skew ? (int)(((long)p.skew * blockIdx.x) % kch) : 0;
  u32x4 rx[ND][KC][2], rw[ND][KC][WP];
  auto load = [&](int s, u32x4 (&ax)[KC][2], u32x4 (&aw)[KC][WP]) {
    s = s < total ? s : total - 1;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int g = s * KC + j, pr = g / kch;
      int kc = g - pr * kch + kc0; kc = kc >= kch ? kc - kch : kc;
      if constexpr (ABF) {
        ax[j][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.A[pr]) + (long)(m0 + xr) * p.lda + kc * 64 + xc * 8));
        ax[j][1] = ax[j][0];
      } else {
        const float* a = p.A[pr] + (long)(m0 + xr) * p.lda + kc * 64 + xc * 8;
        ax[j][0] = *reinterpret_cast<const u32x4*>(a); ax[j][1] = *reinterpret_cast<const u32x4*>(a + 4);
      }
      const bf16_t* w = p.W[pr] + kc * 64 + xc * 8;
#pragma unroll
      for (int u = 0; u < WP; ++u) aw[j][u] = *reinterpret_cast<const u32x4*>(w + (long)(n0 + xr + 64 * u) * p.ldw);
    }
  };
  auto write = [&](int s, const u32x4 (&ax)[KC][2], const u32x4 (&aw)[KC][WP]) {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      unsigned char* st = smem + (s % kNtsRing) * kSt + j * kCh;
      const u32x4 o = ABF ? ax[j][0] : u32x4{pack2_bf16(__uint_as_float(ax[j][0].x), __uint_as_float(ax[j][0].y)), pack2_bf16(__uint_as_float(ax[j][0].z), __uint_as_float(ax[j][0].w)),
                       pack2_bf16(__uint_as_float(ax[j][1].x), __uint_as_float(ax[j][1].y)), pack2_bf16(__uint_as_float(ax[j][1].z), __uint_as_float(ax[j][1].w))};
      *reinterpret_cast<u32x4*>(st + xr * 128 + ((xc ^ ((xr >> 1) & 7)) * 16)) = o;
#pragma unroll
      for (int u = 0; u < WP; ++u) {
        const int r = xr + 64 * u;
        *reinterpret_cast<u32x4*>(st + kX + r * 128 + ((xc ^ ((r >> 1) & 7)) * 16)) = aw[j][u];
      }
    }
  };
  auto step = [&](int s, u32x4 (&ax)[KC][2], u32x4 (&aw)[KC][WP]) {     // buffer (s + 1) % ND
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write(s + 1, ax, aw); load(s + 1 + ND, ax, aw);
  };
  // barrier s: stage s is written before it; after it the slot of stage s-1 is free; stage s+1 goes into slot (s+1) % 3, which held
  // stage s-2 -- released at barrier s-1
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

// ---------------------------------------------------------------------------------------------------------------------------
// Input projections of a Bidirectional recurrent layer, forward (utils.py:77-82: x W + b of both directions, hoisted out of the recurrences):
//     Yd[M][N] (fp32) = X[M][K] . Wd[N][K]^T + bd[N]      d = forward | backward direction, M = T*B rows, K = 128 | 256, N = 4u (3u)
// The stripe kernel above runs this as one workgroup per (64-row stripe, 256-column slab) and direction: with K this short a workgroup's life is one load
// round trip, it re-stages its 64-128 KiB weight slab for 2-4 MFMA k-chunks, and 120 KiB of LDS leave one workgroup per CU -- 28-37 us per launch for
// 61 MB of traffic (round 5 timeline), four launches per step.  Here BOTH directions are one launch of persistent workgroups: a workgroup owns a column slab
// of one direction, keeps the slab's weights [256][K] in LDS for its whole life and walks the row stripes blockIdx.x, + gridDim.x, ...; the IO waves stream
// only the X chunks (fp32 -> bf16) two stages ahead through the ring, the MFMA waves store a finished stripe while the next one's chunks are already landing.
// Same products in the same order as the stripe kernel (bf16 operands, k ascending, bias added last): bit-identical results.
namespace {
struct NtpParams {
  const float* X; const bf16_t* W[2]; float* Y[2]; const float* bias[2];
  int M, N, K, lda, ldw, ldy;              // N = gate columns of ONE direction (a multiple of 256)
};
__global__ __launch_bounds__(768) void gemm_nt_f32_proj_kernel(NtpParams p) {
  constexpr int CB = 2, NS = 128 * CB;                         // 256-column slabs
  constexpr int kX = 64 * 128, kW = NS * 128;                  // bytes of an X stage (64 rows x 64 k bf16), of one 64-k chunk of the weight slab
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // weight slab [kch][NS rows][64 k] | ring of kNtsRing X stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kch = p.K / 64;
  const int slabs = p.N / NS;                                  // per direction
  const int dir = (int)blockIdx.y / slabs, n0 = ((int)blockIdx.y - dir * slabs) * NS;
  const int stripes = p.M / 64;
  const int mine = ((int)blockIdx.x < stripes) ? (stripes - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = mine * kch;
  unsigned char* const ring = smem + kch * kW;
  if (mine <= 0) return;

  if (wave < 4) {
    const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
    f32x16 acc[2][CB];
    float* const Y = p.Y[dir]; const float* const bias = p.bias[dir];
    int slot = 0, s = 0;
    for (int it = 0; it < mine; ++it) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[b][cb][e] = 0.f;
      for (int kc = 0; kc < kch; ++kc, ++s) {
        __builtin_amdgcn_s_barrier();
        const unsigned char* Xs = ring + slot * kX + l31 * 128;
        const unsigned char* Ws = smem + kc * kW + (wave * 32 * CB + l31) * 128;
        slot = slot + 1 == kNtsRing ? 0 : slot + 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int off = ((2 * ks + half) ^ sw) * 16;
          bf16x8_t fx[2], fw[CB];
#pragma unroll
          for (int b = 0; b < 2; ++b) fx[b] = *reinterpret_cast<const bf16x8_t*>(Xs + b * 32 * 128 + off);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) fw[cb] = *reinterpret_cast<const bf16x8_t*>(Ws + cb * 32 * 128 + off);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[b][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[cb], fx[b], acc[b][cb], 0, 0, 0);
        }
      }
      // the stripe is finished: lane = one row; register group g of a block = columns 8 g + 4 half + 0..3 (the IO waves are already two stages into the next one)
      const int m0 = ((int)blockIdx.x + it * (int)gridDim.x) * 64;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col0 = n0 + wave * 32 * CB + 4 * half;
        float* yrow = Y + (long)(m0 + 32 * b + l31) * p.ldy + col0;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(acc[b][cb][4 * g], acc[b][cb][4 * g + 1], acc[b][cb][4 * g + 2], acc[b][cb][4 * g + 3]);
            if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + col0 + 32 * cb + 8 * g); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
            *reinterpret_cast<float4*>(yrow + 32 * cb + 8 * g) = v;
          }
      }
    }
    __builtin_amdgcn_s_barrier();
    return;
  }
  // ---------------------------------------------------------------------------- IO waves (8): lane il = 0..511
  const int il = tid - 256;
  const int xr = il >> 3, xc = il & 7;                          // piece: row xr (0..63), 8-k piece xc
  {  // the weight slab, once: NS rows x kch chunks x 8 pieces of 16 bytes, swizzled like a stage
    const bf16_t* W = p.W[dir];
    for (int kc = 0; kc < kch; ++kc)
#pragma unroll
      for (int u = 0; u < NS / 64; ++u) {
        const int r = xr + 64 * u;
        const u32x4 v = *reinterpret_cast<const u32x4*>(W + (long)(n0 + r) * p.ldw + kc * 64 + xc * 8);
        *reinterpret_cast<u32x4*>(smem + kc * kW + r * 128 + ((xc ^ ((r >> 1) & 7)) * 16)) = v;
      }
  }
  u32x4 rx[kNtsD][2];
  auto load = [&](int s, u32x4 (&ax)[2]) {
    s = s < total ? s : total - 1;
    const int it = s / kch, kc = s - it * kch;
    const float* a = p.X + (long)(((int)blockIdx.x + it * (int)gridDim.x) * 64 + xr) * p.lda + kc * 64 + xc * 8;
    ax[0] = *reinterpret_cast<const u32x4*>(a); ax[1] = *reinterpret_cast<const u32x4*>(a + 4);
  };
  auto write = [&](int s, const u32x4 (&ax)[2]) {
    unsigned char* st = ring + (s % kNtsRing) * kX;
    const u32x4 o = {pack2_bf16(__uint_as_float(ax[0].x), __uint_as_float(ax[0].y)), pack2_bf16(__uint_as_float(ax[0].z), __uint_as_float(ax[0].w)),
                     pack2_bf16(__uint_as_float(ax[1].x), __uint_as_float(ax[1].y)), pack2_bf16(__uint_as_float(ax[1].z), __uint_as_float(ax[1].w))};
    *reinterpret_cast<u32x4*>(st + xr * 128 + ((xc ^ ((xr >> 1) & 7)) * 16)) = o;
  };
  // barrier s: stage s (and, at s = 0, the weight slab) is written before it; after it the slot of stage s-1 is free; stage s+1 goes into slot
  // (s+1) % 3, which held stage s-2 -- released at barrier s-1
  load(0, rx[0]); load(1, rx[1]);
  write(0, rx[0]); load(2, rx[0]);
  for (int s = 0; s < total; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < total) {
      if ((s + 1) & 1) { write(s + 1, rx[1]); load(s + 3, rx[1]); } else { write(s + 1, rx[0]); load(s + 3, rx[0]); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
}  // namespace

// Both directions' input projections of a Bidirectional layer in one launch: Yf = X . Wf^T + bf, Yb = X . Wb^T + bb (fp32 X [M][K] row stride lda, bf16
// W [N][K] row stride ldw, fp32 Y [M][N] row stride ldy; bias may be NULL for both).  Bit-identical to two crnn_gemm_nt_f32_stream_bias calls.
// Supported (else -3): M % 64 == 0, N % 256 == 0, K = 64 | 128 | 192 | 256 (the weight slab stays in LDS), leading dimensions multiples of 8, 16-byte aligned pointers.
extern "C" int crnn_rnn_input_proj_supported(int M, int N, int K) {
  return (M > 0 && M % 64 == 0 && N > 0 && N % 256 == 0 && K >= 64 && K <= 256 && K % 64 == 0 && 2 * (N / 256) <= 65535) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_rnn_input_proj(const float* X, const void* Wf, const void* Wb, const float* bias_f, const float* bias_b, float* Yf, float* Yb, int M, int N,
                                   int K, int lda, int ldw, int ldy, hipStream_t stream) {
  if (!X || !Wf || !Wb || !Yf || !Yb || ((bias_f != nullptr) != (bias_b != nullptr))) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_rnn_input_proj_supported(M, N, K));
  if (((lda | ldw | ldy) & 7) || lda < K || ldw < K || ldy < N) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)Wf | (uintptr_t)Wb | (uintptr_t)Yf | (uintptr_t)Yb | (uintptr_t)bias_f | (uintptr_t)bias_b) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)M * (lda > ldy ? lda : ldy) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  NtpParams p;
  p.X = X; p.W[0] = (const bf16_t*)Wf; p.W[1] = (const bf16_t*)Wb; p.Y[0] = Yf; p.Y[1] = Yb; p.bias[0] = bias_f; p.bias[1] = bias_b;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldy = ldy;
  const int slabs2 = 2 * (N / 256);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  int groups = cus / slabs2; if (groups < 1) groups = 1;       // one workgroup per CU: (row groups) x (slabs of both directions)
  if (groups > M / 64) groups = M / 64;
  const int lds = (K / 64) * 256 * 128 + kNtsRing * 64 * 128;
  CRNN_LDS_ATTR(gemm_nt_f32_proj_kernel, 4 * 256 * 128 + kNtsRing * 64 * 128);
  hipLaunchKernelGGL(gemm_nt_f32_proj_kernel, dim3(groups, slabs2), dim3(768), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// Y[M][N] (fp32, row stride ldy) = A0[M][K] . W0[N][K]^T (+ A1 . W1^T when A1 != NULL) (+ bias[N]); A fp32 (row stride lda), W bf16 (row stride
// ldw).  Supported (else -3): M % 64 == 0, N % 128 == 0 (column slabs of 256, or of 128 when N % 256 != 0), K % 64 == 0, leading dimensions
// multiples of 8, 16-byte aligned pointers.
static int nts_launch(const float* A0, const void* W0, const float* A1, const void* W1, float* Y, const float* bias, int M, int N,
                      int K, int lda, int ldw, int ldy, int skew, hipStream_t stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !A0 || !W0 || !Y || ((A1 != nullptr) != (W1 != nullptr))) return CRNN_ERR_ARG;
  if (M % 64 || N % 128 || K % 64 || ((lda | ldw | ldy) & 7) || lda < K || ldw < K || ldy < N) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)A1 | (uintptr_t)W1 | (uintptr_t)Y | (uintptr_t)bias) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)M * (lda > ldy ? lda : ldy) >= (1L << 31) || N > 65535 * 128) return CRNN_ERR_UNSUPPORTED;
  const int slab = (N % 256 == 0) ? 256 : 128;
  NtsParams p;
  p.A[0] = A0; p.A[1] = A1 ? A1 : A0; p.W[0] = (const bf16_t*)W0; p.W[1] = (const bf16_t*)(W1 ? W1 : W0); p.Y = Y; p.bias = bias;
  p.M = M; p.N = slab; p.K = K; p.lda = lda; p.ldw = ldw; p.ldy = ldy; p.npairs = A1 ? 2 : 1; p.skew = skew;
  const int lds = kNtsRing * (64 * 128 + slab * 128);
  CRNN_LDS_ATTR(gemm_nt_f32_stream_kernel<1>, kNtsRing * (64 * 128 + 128 * 128));
  CRNN_LDS_ATTR(gemm_nt_f32_stream_kernel<2>, kNtsRing * (64 * 128 + 256 * 128));
  CRNN_LDS_ATTR((gemm_nt_f32_stream_kernel<1, false, 2>), 2 * kNtsRing * (64 * 128 + 128 * 128));
  // 128-column slabs with an even number of chunks: two chunks per stage and barrier (144 KiB of ring; the 256-column slab's would not fit)
  if (slab == 128 && ((K / 64) * p.npairs) % 2 == 0 && crnn_knob("CRNN_NTS_KC", 2) == 2)
    hipLaunchKernelGGL((gemm_nt_f32_stream_kernel<1, false, 2>), dim3(M / 64, N / slab), dim3(768), 2 * lds, stream, p);
  else if (slab == 128) hipLaunchKernelGGL(gemm_nt_f32_stream_kernel<1>, dim3(M / 64, N / slab), dim3(768), lds, stream, p);
  else hipLaunchKernelGGL(gemm_nt_f32_stream_kernel<2>, dim3(M / 64, N / slab), dim3(768), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// dense1's forward (round 5; utils.py:72-75): Y[perm(m)][N] (fp32) = Dropout(ReLU(X[M][K] . W[N][K]^T + bias)) with X bf16 (block 7's output rows, batch-major),
// W the bf16 W^T copy, one workgroup per 64-row stripe over the whole K = 4608 reduction (the tile GEMM ran it as 8 reduction ranges + a second stage + a
// dropout pass: 63 us for 122 MB).  permP / drop_rate / relu as in NtsParams.  Supported (else -3): M % 64 == 0 (and % permP), N = 128 | 256 = ldy,
// K % 64 == 0, leading dimensions multiples of 8, 16-byte aligned pointers.
extern "C" int crnn_dense_fwd_stream_supported(long M, int N, long K) {
  return (M > 0 && M % 64 == 0 && (N == 128 || N == 256) && K >= 64 && K % 64 == 0 && M * (K > N ? K : (long)N) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dense_fwd_stream(const void* X, const void* WT, const float* bias, float* Y, long M, int N, long K, int lda, int ldw, int relu, int permP,
                                     float drop_rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  if (!X || !WT || !Y || permP < 0 || drop_rate < 0.f || drop_rate >= 1.f) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_dense_fwd_stream_supported(M, N, K));
  if (((lda | ldw) & 7) || lda < K || ldw < K || (permP && M % permP)) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)X | (uintptr_t)WT | (uintptr_t)Y | (uintptr_t)bias) & 15)) return CRNN_ERR_UNSUPPORTED;
  NtsParams p;
  p.A[0] = p.A[1] = reinterpret_cast<const float*>(X); p.W[0] = p.W[1] = (const bf16_t*)WT; p.Y = Y; p.bias = bias;
  p.M = (int)M; p.N = N; p.K = (int)K; p.lda = lda; p.ldw = ldw; p.ldy = N; p.npairs = 1;
  // The rotated start of the reduction walk (p.skew, see the kernel) makes a row's fp32 sum depend on which 64-row stripe it sits in.  Training launches
  // (drop_rate > 0: the dropout decision depends on the row index anyway) take it; inference launches walk the chunks in ascending order in every workgroup,
  // so that a sample's activations do not depend on its position in the batch or on the batch size (round 6, ADVICE; test_dense1_stream_inference_rows_do_
  // not_depend_on_their_position).
  p.relu = relu; p.permP = permP; p.drop_rate = drop_rate; p.seed = seed; p.layer = layer; p.skew = drop_rate > 0.f ? crnn_knob("CRNN_NTS_SKEW", 3) : 0;
  const int lds = kNtsRing * (64 * 128 + N * 128);
  if (N == 128 && (K / 64) % 2 == 0 && crnn_knob("CRNN_NTS_KC", 2) == 2) {
    CRNN_LDS_ATTR((gemm_nt_f32_stream_kernel<1, true, 2>), 2 * lds);
    hipLaunchKernelGGL((gemm_nt_f32_stream_kernel<1, true, 2>), dim3((unsigned)(M / 64), 1), dim3(768), 2 * lds, stream, p);
  } else if (N == 128) { CRNN_LDS_ATTR((gemm_nt_f32_stream_kernel<1, true>), lds); hipLaunchKernelGGL((gemm_nt_f32_stream_kernel<1, true>), dim3((unsigned)(M / 64), 1), dim3(768), lds, stream, p); }
  else { CRNN_LDS_ATTR((gemm_nt_f32_stream_kernel<2, true>), lds); hipLaunchKernelGGL((gemm_nt_f32_stream_kernel<2, true>), dim3((unsigned)(M / 64), 1), dim3(768), lds, stream, p); }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// (the forward input projections and dense2: chunks in ascending order in every workgroup -- the persistent projection kernel's order, bit for bit)
extern "C" int crnn_gemm_nt_f32_stream_bias(const float* A0, const void* W0, const float* A1, const void* W1, float* Y, const float* bias, int M, int N,
                                            int K, int lda, int ldw, int ldy, hipStream_t stream) {
  return nts_launch(A0, W0, A1, W1, Y, bias, M, N, K, lda, ldw, ldy, 0, stream);
}
// (the recurrent layers' input gradients: each workgroup starts its walk over the reduction three chunks after its neighbour's -- see the kernel)
extern "C" int crnn_gemm_nt_f32_stream(const float* A0, const void* W0, const float* A1, const void* W1, float* Y, int M, int N, int K, int lda,
                                       int ldw, int ldy, hipStream_t stream) {
  if (N != 128 && N != 256) return (M <= 0 || K <= 0) ? CRNN_ERR_ARG : CRNN_ERR_UNSUPPORTED;
  return nts_launch(A0, W0, A1, W1, Y, nullptr, M, N, K, lda, ldw, ldy, 3, stream);
}
