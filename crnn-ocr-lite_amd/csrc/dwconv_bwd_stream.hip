// Backward of one depthwise stage of a conv block (bf16 storage, training) as a ROW STREAM -- the arithmetic of dw_bwd_fused_kernel
// (conv_bwd_fused.hip: BatchNorm-backward pass 2 + depthwise weight gradient + depthwise data gradient, reference utils.py:44-46) on
// the schedule of dw_fwd_stream_kernel (dwconv_stream.hip):
//     dd = scale * (gy - c1 - xhat * c2),  gy = da * [0 < BN(d) < 6]           (rounded to bf16 as the stored tensor would be)
//     dx[y][x] = sum_ij k[8 - (3i+j)] dd[y+i-1][x+j-1]                           dk[3i+j] = sum_yx xin[y+i-1][x+j-1] dd[y][x]
// A workgroup walks a band of rows of one image and one channel range (288 16-byte columns = pixel x 8 channels); per step the LOADER
// wave brings one row each of d, da and xin (the latter one row behind) into an LDS ring with global_load_lds, and two groups of
// compute waves work on it:
//   * the five DK waves own a column each: they form dd of the arriving d/da row (own column only), leave it as bf16 in a two-row LDS
//     buffer for the other group, and add the arriving xin row (x-1, x, x+1 from LDS) times the last three dd values of their column
//     to the 72 weight-gradient sums they keep in registers (dk[0..2] with dd[r+1], dk[3..5] with dd[r], dk[6..8] with dd[r-1]);
//   * the five DX waves read the previous dd row (x-1, x, x+1) and run the forward stream kernel's three running output rows with the
//     mirrored taps: every dx value is the same fp32 fma chain as in the tile kernels (bit-identical).
// The two groups have different register sets (72 sums + 32 BatchNorm constants vs 72 taps + 24 running sums) -- one wave holding both
// would need 260 registers -- and unequal work, so the roles are dealt over the wave slots such that every SIMD carries about the same
// (wave w runs on SIMD w mod 4).  Nothing is re-read: 4 tensor passes of HBM traffic (+ 2 halo rows per band), one barrier per row.
//
// PROLOGUE form (round 4): `xin` is the previous block's pointwise output q -- the block output x = Dropout(ReLU6(BatchNorm-2(q)))
// (utils.py:48-56) is not kept by the forward (dwconv_stream.hip) and is re-formed here, bit for bit, one row ahead into a two-row LDS
// buffer the DK waves take their weight-gradient operand from; one more stage in flight.  The re-forming (~65 VALU operations per
// 16-byte chunk; the dropout decisions come as keep bytes, crnn_dropout_keep_bytes, which the loader wave brings along) is done by
// the DX waves, each for its own column: spread over five waves it fits next to their ~70 operations per step, where one dedicated
// wave for all five column groups became the critical path (profiles/r04_*: + 23 % kernel time).
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BN_EPS_F 1e-3f

#ifndef CRNN_DBS_EXP
#define CRNN_DBS_EXP 0      // experiment builds: 1 = no DMA, 2 = no stores, 4 = no dk fmas, 8 = no dx fmas
#endif
#ifndef CRNN_DBS_D
#define CRNN_DBS_D 3
#endif
#ifndef CRNN_DBS_F32_WGS
#define CRNN_DBS_F32_WGS 1  // workgroups per CU the fp32 form is compiled for (2 = 6 waves per SIMD = 80 registers: the DK waves spill 81 -- not used)
#endif

struct DbsParams {
  const unsigned char *d, *da, *xin; unsigned char* dx; const float *bnstate, *coef, *k; float* partials;
  int H, W, C, HB, nwgb, nsplit, cols, cppw, rowbytes;
  // prologue form: xin = q of the previous block, its BatchNorm-2 state [mean|var|scale|shift] and dropout site
  const float* pro_bn; const unsigned char* keep; float rate; float* bn2_partials;
};

#ifndef CRNN_DBS_NT
#define CRNN_DBS_NT 1      // round 5: d, da, xin are read here for the last time in the step -> nontemporal LDS-DMA (they do not push the kernel's own output,
#endif                     // which the next BatchNorm backward reads at once, out of the last-level cache): -0.07 ms per step, same bits (profiles/r05_nt_loads_ab.txt)
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, CRNN_DBS_NT ? 2 : 0);
}
__device__ __forceinline__ void widen8(const u32x4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// element forms of a 16-byte chunk: 8 bf16 (EPC = 8) or 4 fp32 (EPC = 4) -- the fp32 form runs the same schedule on half as many channels
// per lane and twice as many channel ranges per row (round 4: the parity mode's depthwise-stage backward)
template <int EPC> __device__ __forceinline__ void widenE(const u32x4& u, float (&f)[EPC]);
template <> __device__ __forceinline__ void widenE<8>(const u32x4& u, float (&f)[8]) { widen8(u, f); }
template <> __device__ __forceinline__ void widenE<4>(const u32x4& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <int EPC> __device__ __forceinline__ u32x4 packE(const float (&f)[EPC]);
template <> __device__ __forceinline__ u32x4 packE<8>(const float (&f)[8]) {
  u32x4 o; o.x = pack2_bf16(f[0], f[1]); o.y = pack2_bf16(f[2], f[3]); o.z = pack2_bf16(f[4], f[5]); o.w = pack2_bf16(f[6], f[7]); return o;
}
template <> __device__ __forceinline__ u32x4 packE<4>(const float (&f)[4]) {
  u32x4 o; o.x = __float_as_uint(f[0]); o.y = __float_as_uint(f[1]); o.z = __float_as_uint(f[2]); o.w = __float_as_uint(f[3]); return o;
}

constexpr int kCW = 5;                       // compute waves per group (320 columns)
constexpr int kSub = 5 * 1024;               // one row of one tensor in a ring stage (5 DMA instructions)
constexpr int kStageB = 3 * kSub;            // d | da | xin
constexpr int kNI = 15;                      // DMA instructions per stage
// LDS plan for KD stages in flight (prologue form: + two rows of re-formed x)
// STATS: statistics form; DXS: the statistics are taken by the DX waves themselves (fp32 tensors) instead of by a twelfth wave
template <int KD, bool STATS = false, bool DXS = false>
struct DbsLds {
  static constexpr int NR = KD + (STATS ? (DXS ? 2 : 3) : 1);   // (the statistics form reads a stage's raw xin row once more, two | three steps after it landed)
  static constexpr int DdOff = NR * kStageB;        // two dd rows
  static constexpr int ZOff = DdOff + 2 * kSub;     // 16 zero bytes
  static constexpr int CstOff = ZOff + 64;          // BatchNorm constants of the workgroup's channels: scale | shift | P | Q, [4][<= 256] floats
  static constexpr int XtOff = CstOff + 4 * 256 * 4;   // prologue form: two rows of x = Dropout(ReLU6(BatchNorm-2(q)))
  static constexpr int PcOff = XtOff + 2 * kSub;       // prologue form: BatchNorm-2 scale | shift of the workgroup's channels, [2][<= 256] floats
  static constexpr int KeepOff = PcOff + 2 * 256 * 4;  // prologue form: NR x 512 bytes of keep bytes (one per 16-byte chunk of the stage's xin row)
  static constexpr int DxrOff = KeepOff + NR * 512;    // statistics form: two rows of dx (bf16, as stored) for the statistics wave
  static constexpr int Total = DxrOff + ((STATS && !DXS) ? 2 * kSub : 0);
  static constexpr int SredOff = 64 * 1024;            // DXS: the DX waves' statistics records [5][2][<= 256] after the ring is free (behind the DK waves' 45 KiB)
  static_assert(!DXS || NR * kStageB >= SredOff + 5 * 2 * 256 * 4, "statistics records inside the ring");
  static_assert((KD - 1) * kNI <= 63, "vmcnt is a 6-bit counter");
};
constexpr int kD = CRNN_DBS_D;

// role of wave slot w (11 waves; SIMD = w mod 4): DK work is about twice DX work per step
//   SIMD 3: w3 w7 = DK DK;  SIMD 0: w0 w4 w8 = DK DX DX;  SIMD 1: w1 w5 w9 = DK DX DX;  SIMD 2: w2 w6 w10 = DK DX loader
__device__ __forceinline__ int role_of(int w, int& idx) {   // 0 = DK, 1 = DX, 2 = loader; idx = index inside the group
  if (w == 10) { idx = 0; return 2; }
  if (w < 4) { idx = w; return 0; }
  if (w == 7) { idx = 4; return 0; }
  idx = w < 7 ? w - 4 : w - 5;                                // 4 5 6 8 9 -> 0 1 2 3 4
  return 1;
}

// statistics form (12 waves): DK ~145 operations per step, DX ~135 (with the re-forming of x), the statistics wave ~400 --
//   SIMD 0: w0 w4 w8 = DK DK DX;  SIMD 1: w1 w5 w9 = DK DK DX;  SIMD 2: w2 w6 w10 = DK DX DX;  SIMD 3: w3 w7 w11 = DX loader S
__device__ __forceinline__ int role_of_stats(int w, int& idx) {   // 0 = DK, 1 = DX, 2 = loader, 3 = statistics
  switch (w) {
    case 0: idx = 0; return 0;  case 4: idx = 1; return 0;  case 1: idx = 2; return 0;  case 5: idx = 3; return 0;  case 2: idx = 4; return 0;
    case 8: idx = 0; return 1;  case 9: idx = 1; return 1;  case 6: idx = 2; return 1;  case 10: idx = 3; return 1;  case 3: idx = 4; return 1;
    case 7: idx = 0; return 2;
    default: idx = 0; return 3;   // w11
  }
}

// STATS (prologue form only, opt-in: CRNN_FLAG_BN2_STATS_FUSION): a twelfth wave takes the statistics pass of the producer's BatchNorm-2
// backward -- per channel sum gy and sum gy * xhat with gy = dx * dropout * [0 < q * scale + shift < 6] (what bn_bwd_kernel<1> reads q and dx
// again for) -- from the dx rows the DX waves leave in LDS beside their global stores and the raw q rows still in the ring (three stages
// deeper); one partial row [2][C] per workgroup band.  Measured (profiles/r04_*, four launches at batch 256): the pass it replaces 0.28 ms;
// this kernel + 0.24 ms (the one wave's ~400 operations per step are the critical path); done by the DX waves themselves + 0.65 (36
// spilled registers); gating in the DX waves and sums in the twelfth + 0.35 -- so it is not the default schedule.
template <int KD, bool PRO, bool DROP, bool STATS = false, bool F32 = false>
__global__ __launch_bounds__((STATS && !F32) ? 768 : 704, (F32 && CRNN_DBS_F32_WGS > 1) ? 3 * CRNN_DBS_F32_WGS : 1) void dw_bwd_stream_kernel(DbsParams p) {
  static_assert(!STATS || PRO, "the statistics form is a prologue form");
  // fp32 tensors: half the elements per step leave the DX waves the issue slots and registers to take the statistics themselves (their dx values are
  // in registers: no LDS round trip, no twelfth wave whose ~250 operations per step were the critical path: + 0.25 ms over four launches at batch 256)
  constexpr bool SW = STATS && !F32;                 // statistics wave present
  constexpr bool DXS = STATS && F32;                 // statistics in the DX waves
  constexpr int EPC = F32 ? 4 : 8;                   // elements per 16-byte chunk
  constexpr int ES = F32 ? 4 : 2;                    // bytes per element
  typedef DbsLds<KD, STATS, DXS> LP;
  constexpr int kNR = LP::NR, kDdOff = LP::DdOff, kZOff = LP::ZOff, kCstOff = LP::CstOff, kXtOff = LP::XtOff, kPcOff = LP::PcOff, kKeepOff = LP::KeepOff, kDxrOff = LP::DxrOff;
  constexpr int kNIT = kNI + (DROP ? 2 : 0);           // DMA instructions per stage
  static_assert((KD - 1) * kNIT <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int gidx; const int role = SW ? role_of_stats(wave, gidx) : role_of(wave, gidx);
  int bid = blockIdx.x;
  const int split = bid % p.nsplit; bid /= p.nsplit;
  const int wb = bid % p.nwgb, img = bid / p.nwgb;
  const int r0 = wb * p.HB;                         // first row of the band
  const int nsteps = p.HB + 3;                      // step s: d/da row r0-1+s (s <= HB+1), xin row r0-2+s (s >= 1), dx row r0+s-3 (s >= 3)
  const int c0 = split * (p.C / p.nsplit);          // first channel
  if (tid < 4) reinterpret_cast<unsigned*>(lds + kZOff)[tid] = 0u;
  const long imgoff = (long)img * p.H * p.rowbytes;

  if (role == 2) {
    // ------------------------------------------------------------------ loader wave
    int goff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int f = i * 64 + lane; f = f < p.cols ? f : p.cols - 1;
      const int px = f / p.cppw, o = f - px * p.cppw;
      goff[i] = (px * p.C + c0) * ES + o * 16;
    }
    const unsigned char* gd = p.d + imgoff; const unsigned char* gg = p.da + imgoff; const unsigned char* gxx = p.xin + imgoff;
    // prologue form with dropout: the keep bytes of the stage's xin row for this workgroup's columns, 4 bytes (4 columns of one pixel) per lane
    // (one keep byte per 8 elements: per column for bf16 tensors, per pair of columns for fp32 ones)
    const int cpp = p.C >> 3, rowcols = p.W * cpp;
    int koff[2];
    const unsigned char* gk = DROP ? p.keep + (long)img * p.H * rowcols : nullptr;
    if (DROP) {
      const int kpp = p.cppw * EPC / 8, kcols = p.cols * EPC / 8;   // keep bytes per pixel of the channel range, per stage row
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int c4 = (i * 64 + lane) * 4; if (c4 >= kcols) c4 = kcols - 4;
        const int px = c4 / kpp, o = c4 - px * kpp;
        koff[i] = px * cpp + (c0 >> 3) + o;
      }
    }
    auto issue = [&](int s, int slot) {
      s = s < nsteps ? s : nsteps - 1;
      int rd = r0 - 1 + s; rd = rd < 0 ? 0 : (rd >= p.H ? p.H - 1 : rd);
      int rx = r0 - 2 + s; rx = rx < 0 ? 0 : (rx >= p.H ? p.H - 1 : rx);
      const long od = (long)rd * p.rowbytes, ox = (long)rx * p.rowbytes;
      unsigned char* dst = lds + slot * kStageB;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (!(CRNN_DBS_EXP & 1)) {
          glds16(gd + od + goff[i], dst + i * 1024);
          glds16(gg + od + goff[i], dst + kSub + i * 1024);
          glds16(gxx + ox + goff[i], dst + 2 * kSub + i * 1024);
        }
      }
      if (DROP) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gk + (long)rx * rowcols + koff[i]),
                                           (__attribute__((address_space(3))) void*)(lds + kKeepOff + slot * 512 + i * 256), 4, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < KD; ++s) issue(s, s);
    int slot = KD;
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KD - (PRO ? 2 : 1)) * kNIT) : "memory");   // stage s (prologue form: s + 1) has landed
      __builtin_amdgcn_s_barrier();
      issue(s + KD, slot);
      slot = slot + 1 == kNR ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (SW) __builtin_amdgcn_s_barrier();            // (the statistics wave takes its last row before the ring is reused)
    __builtin_amdgcn_s_barrier();
    return;
  }
  if (SW && role == 3) {
    // ------------------------------------------------------------------ statistics wave: all five column groups, one step behind the DX waves
    int offS[kCW]; bool actS[kCW];
#pragma unroll
    for (int g = 0; g < kCW; ++g) { const int c = g * 64 + lane; actS[g] = c < p.cols; offS[g] = actS[g] ? c : p.cols - 1; }
    // 64 % cppw == 0 (cppw is a power of two <= 32): every group of a lane holds the same channel octet -- one set of constants and sums
    const int chS = c0 + (lane % p.cppw) * EPC;
    constexpr int HP = EPC / 2;                                    // channel pairs per column
    f32x2_t sc[HP], sh[HP], iv[HP], nm[HP], st_s[HP], st_q[HP];    // xhat = q * inv - mean * inv
#pragma unroll
    for (int e = 0; e < HP; ++e) {
      sc[e] = (f32x2_t){p.pro_bn[2 * p.C + chS + 2 * e], p.pro_bn[2 * p.C + chS + 2 * e + 1]};
      sh[e] = (f32x2_t){p.pro_bn[3 * p.C + chS + 2 * e], p.pro_bn[3 * p.C + chS + 2 * e + 1]};
      iv[e] = (f32x2_t){1.0f / sqrtf(p.pro_bn[p.C + chS + 2 * e] + BN_EPS_F), 1.0f / sqrtf(p.pro_bn[p.C + chS + 2 * e + 1] + BN_EPS_F)};
      nm[e] = (f32x2_t){-p.pro_bn[chS + 2 * e] * iv[e].x, -p.pro_bn[chS + 2 * e + 1] * iv[e].y};
      st_s[e] = (f32x2_t){0.f, 0.f}; st_q[e] = (f32x2_t){0.f, 0.f};
    }
    int qslot = 2;                                   // ring slot of stage b - 2 at barrier index b = 4
    // barrier index b: dx row b - 4 (image row r0 + b - 4, left in LDS by the DX waves in their step b - 2) and the raw q row of stage b - 2:
    // gy = the stored bf16 dx where the element was kept and 0 < q * scale + shift < 6 (bn_bwd_kernel's gy without the 1 / (1 - rate) factor)
    auto take = [&](int b) {
      const unsigned char* dr = lds + kDxrOff + (b & 1) * kSub;
      const unsigned char* qr = lds + qslot * kStageB + 2 * kSub;
      const unsigned char* kr = lds + kKeepOff + qslot * 512;
      qslot = qslot + 1 == kNR ? 0 : qslot + 1;
      u32x4 vg[kCW], vq[kCW]; uint32_t kq[kCW];
#pragma unroll
      for (int g = 0; g < kCW; ++g) {
        vg[g] = *reinterpret_cast<const u32x4*>(dr + offS[g] * 16); vq[g] = *reinterpret_cast<const u32x4*>(qr + offS[g] * 16);
        kq[g] = DROP ? (F32 ? ((uint32_t)kr[offS[g] >> 1] >> ((offS[g] & 1) * 4)) : (uint32_t)kr[offS[g]]) : 0xffu;
      }
#pragma unroll
      for (int g = 0; g < kCW; ++g) {
        if (g * 64 >= p.cols) continue;              // (uniform)
        if (F32) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const f32x2_t x2 = (f32x2_t){__uint_as_float(vq[g][2 * q]), __uint_as_float(vq[g][2 * q + 1])};
            const f32x2_t t = __builtin_elementwise_fma(x2, sc[q], sh[q]);
            f32x2_t gy = actS[g] ? (f32x2_t){__uint_as_float(vg[g][2 * q]), __uint_as_float(vg[g][2 * q + 1])} : (f32x2_t){0.f, 0.f};
            gy = (f32x2_t){(((kq[g] >> (2 * q)) & 1u) && t.x > 0.f && t.x < 6.f) ? gy.x : 0.f,
                           (((kq[g] >> (2 * q + 1)) & 1u) && t.y > 0.f && t.y < 6.f) ? gy.y : 0.f};
            st_s[q] += gy;
            st_q[q] = __builtin_elementwise_fma(gy, __builtin_elementwise_fma(x2, iv[q], nm[q]), st_q[q]);
          }
          continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2_t x2 = (f32x2_t){__uint_as_float(vq[g][q] << 16), __uint_as_float(vq[g][q] & 0xffff0000u)};
          const f32x2_t t = __builtin_elementwise_fma(x2, sc[q % HP], sh[q % HP]);
          uint32_t gm = actS[g] ? vg[g][q] : 0u;
          if (DROP) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)kq[g], 2 * q, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)kq[g], 2 * q + 1, 1);
            gm &= __builtin_amdgcn_perm(hi, lo, 0x07060100u);
          }
          f32x2_t gy = (f32x2_t){__uint_as_float(gm << 16), __uint_as_float(gm & 0xffff0000u)};
          gy = (f32x2_t){(t.x > 0.f && t.x < 6.f) ? gy.x : 0.f, (t.y > 0.f && t.y < 6.f) ? gy.y : 0.f};
          st_s[q % HP] += gy;
          st_q[q % HP] = __builtin_elementwise_fma(gy, __builtin_elementwise_fma(x2, iv[q % HP], nm[q % HP]), st_q[q % HP]);
        }
      }
    };
    for (int b = 0; b < nsteps; ++b) {
      __builtin_amdgcn_s_barrier();
      if (b >= 4) take(b);
    }
    __builtin_amdgcn_s_barrier();                    // ring free for everybody else ...
    take(nsteps);                                    // ... after the last row (dx row HB - 1): barrier index HB + 3
    __builtin_amdgcn_s_barrier();
    // lanes l, l + cppw, ... hold the same channels -> xor shuffles; this one wave has seen every column of the workgroup
    for (int o = p.cppw; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < HP; ++e) {
        st_s[e].x += __shfl_xor(st_s[e].x, o, 64); st_s[e].y += __shfl_xor(st_s[e].y, o, 64);
        st_q[e].x += __shfl_xor(st_q[e].x, o, 64); st_q[e].y += __shfl_xor(st_q[e].y, o, 64);
      }
    }
    if (lane < p.cppw) {
      const float ik = DROP ? 1.f / (1.f - p.rate) : 1.f;        // gy carries the dropout's 1 / (1 - rate): applied once here
      float* prow = p.bn2_partials + (long)(blockIdx.x / p.nsplit) * 2 * p.C + chS;
#pragma unroll
      for (int e = 0; e < HP; ++e) {
        prow[2 * e] = st_s[e].x * ik; prow[2 * e + 1] = st_s[e].y * ik;
        prow[p.C + 2 * e] = st_q[e].x * ik; prow[p.C + 2 * e + 1] = st_q[e].y * ik;
      }
    }
    __builtin_amdgcn_s_barrier();
    return;
  }
  const int col = gidx * 64 + lane;
  const bool act = col < p.cols;
  const int ccol = act ? col : p.cols - 1;
  const int px = ccol / p.cppw, oct = ccol - px * p.cppw;
  const int offC = ccol * 16;
  const bool hasL = px > 0, hasR = px < p.W - 1;
  const int pitch = p.cppw * 16;                    // LDS bytes between horizontally adjacent pixels
  const int ch0 = c0 + oct * EPC;

  if (role == 1) {
    // ------------------------------------------------------------------ DX waves: dx = correlation of dd with the mirrored taps
    float kw[9][EPC];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* kp = p.k + (long)(8 - t) * p.C + ch0;
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) {
        const float4 a = *reinterpret_cast<const float4*>(kp + 4 * h);
        kw[t][4 * h] = a.x; kw[t][4 * h + 1] = a.y; kw[t][4 * h + 2] = a.z; kw[t][4 * h + 3] = a.w;
      }
    }
    float X0[EPC], X1[EPC], X2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) X0[e] = X1[e] = X2[e] = 0.f;
    // (spelled as two products: with the factored form the register allocator of the bf16 instantiation spills a 16-byte value inside the DK waves'
    // row loop -- the kernel ran 22 % slower; check with scripts/check_loop_spills.sh after touching this file)
    // (round 6) the lane's byte offset inside a row as ONE opaque 32-bit value beside a wave-uniform base: left as an expression of px and ch0 the allocator kept
    // both alive across the row loop for this one address and spilled them (8 bytes of scratch per lane)
    int ooff = px * p.C * ES + ch0 * ES;
    if constexpr (!PRO) asm volatile("" : "+v"(ooff));     // (the prologue forms hold other values across the loop: opaque there costs them a register more)
    unsigned char* const obase = p.dx + imgoff + (long)r0 * p.rowbytes;
#define orow (obase + ooff)
    // prologue form: re-form x = Dropout(ReLU6(q * scale + shift)) of the stage that arrived one step ahead (own column) into the two-row
    // buffer the DK waves read; bn_act_pool_drop_kernel's arithmetic bit for bit.  (The 16 BatchNorm-2 constants of the lane's channels
    // sit in LDS and are read per step: kw and the running rows fill the registers.)
    const int pcw = p.cppw * EPC;                     // channels of this workgroup
    float* pct = reinterpret_cast<float*>(lds + kPcOff);
    if (PRO)
      for (int i = gidx * 64 + lane; i < pcw; i += kCW * 64) {
        pct[i] = p.pro_bn[2 * p.C + c0 + i]; pct[pcw + i] = p.pro_bn[3 * p.C + c0 + i];
      }
    const float* pcl = pct + oct * EPC;
    const float pik = DROP ? 1.f / (1.f - p.rate) : 1.f;
    int xslot = 1;                                    // ring slot of the next stage to re-form (stage 1 first)
    auto xform = [&](int st) {                        // stage st: xin row r0 - 2 + st  ->  xt[st & 1]
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + xslot * kStageB + 2 * kSub + offC);
      uint32_t kc = 0xffu;
      if (DROP) kc = F32 ? ((uint32_t)lds[kKeepOff + xslot * 512 + (ccol >> 1)] >> ((ccol & 1) * 4)) : (uint32_t)lds[kKeepOff + xslot * 512 + ccol];
      xslot = xslot + 1 == kNR ? 0 : xslot + 1;
      u32x4 o;
      if (F32) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2_t x2 = (f32x2_t){__uint_as_float(v[2 * q]), __uint_as_float(v[2 * q + 1])};
          const f32x2_t psc = *reinterpret_cast<const f32x2_t*>(pcl + 2 * q), psh = *reinterpret_cast<const f32x2_t*>(pcl + pcw + 2 * q);
          f32x2_t y = __builtin_elementwise_fma(x2, psc, psh);
          y = (f32x2_t){relu6f(y.x), relu6f(y.y)};
          if (DROP) {
            y = y * (f32x2_t){pik, pik};                // y >= 0: a dropped element is +0 like y * 0
            y = (f32x2_t){(kc >> (2 * q)) & 1u ? y.x : 0.f, (kc >> (2 * q + 1)) & 1u ? y.y : 0.f};
          }
          o[2 * q] = __float_as_uint(y.x); o[2 * q + 1] = __float_as_uint(y.y);
        }
      } else
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2_t x2 = (f32x2_t){__uint_as_float(v[q] << 16), __uint_as_float(v[q] & 0xffff0000u)};
        const f32x2_t psc = *reinterpret_cast<const f32x2_t*>(pcl + (2 * q) % EPC), psh = *reinterpret_cast<const f32x2_t*>(pcl + pcw + (2 * q) % EPC);
        f32x2_t y = __builtin_elementwise_fma(x2, psc, psh);
        y = (f32x2_t){relu6f(y.x), relu6f(y.y)};
        if (DROP) {
          y = y * (f32x2_t){pik, pik};
          // dropped elements as an AND mask on the packed pair (y >= 0: y * 0 and 0 are the same bits)
          const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)kc, 2 * q, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)kc, 2 * q + 1, 1);
          o[q] = pack2_bf16(y.x, y.y) & __builtin_amdgcn_perm(hi, lo, 0x07060100u);
        } else {
          o[q] = pack2_bf16(y.x, y.y);
        }
      }
      if (act) *reinterpret_cast<u32x4*>(lds + kXtOff + (st & 1) * kSub + offC) = o;
      __builtin_amdgcn_sched_barrier(0);              // its temporaries are dead before the correlation's operands are loaded
    };
    // DXS: the statistics pass of the producer's BatchNorm-2 backward on the finished dx row (A, image row r0 + a - 2) and the raw q row of the same
    // image row (stage a, two stages behind the one being re-formed): gy = dx where the element was kept and 0 < q * scale + shift < 6
    f32x2_t ds_s[EPC / 2], ds_q[EPC / 2], dsc[EPC / 2], dsh[EPC / 2], div_[EPC / 2], dnm[EPC / 2];
    int sslot = 2;                                    // ring slot of stage a at the first statistics step (a = 2)
    if (DXS) {
#pragma unroll
      for (int e = 0; e < EPC / 2; ++e) {
        dsc[e] = (f32x2_t){p.pro_bn[2 * p.C + ch0 + 2 * e], p.pro_bn[2 * p.C + ch0 + 2 * e + 1]};
        dsh[e] = (f32x2_t){p.pro_bn[3 * p.C + ch0 + 2 * e], p.pro_bn[3 * p.C + ch0 + 2 * e + 1]};
        div_[e] = (f32x2_t){1.0f / sqrtf(p.pro_bn[p.C + ch0 + 2 * e] + BN_EPS_F), 1.0f / sqrtf(p.pro_bn[p.C + ch0 + 2 * e + 1] + BN_EPS_F)};
        dnm[e] = (f32x2_t){-p.pro_bn[ch0 + 2 * e] * div_[e].x, -p.pro_bn[ch0 + 2 * e + 1] * div_[e].y};
        ds_s[e] = (f32x2_t){0.f, 0.f}; ds_q[e] = (f32x2_t){0.f, 0.f};
      }
    }
    auto dxstats = [&](const float (&A)[EPC]) {
      if constexpr (DXS) {
        const u32x4 vq = *reinterpret_cast<const u32x4*>(lds + sslot * kStageB + 2 * kSub + offC);
        uint32_t kq = 0xfu;
        if (DROP) kq = (uint32_t)lds[kKeepOff + sslot * 512 + (ccol >> 1)] >> ((ccol & 1) * 4);
#pragma unroll
        for (int q = 0; q < EPC / 2; ++q) {
          const f32x2_t x2 = (f32x2_t){__uint_as_float(vq[2 * q]), __uint_as_float(vq[2 * q + 1])};
          const f32x2_t t = __builtin_elementwise_fma(x2, dsc[q], dsh[q]);
          const f32x2_t gy = (f32x2_t){(((kq >> (2 * q)) & 1u) && t.x > 0.f && t.x < 6.f) ? A[2 * q] : 0.f,
                                       (((kq >> (2 * q + 1)) & 1u) && t.y > 0.f && t.y < 6.f) ? A[2 * q + 1] : 0.f};
          ds_s[q] += gy;
          ds_q[q] = __builtin_elementwise_fma(gy, __builtin_elementwise_fma(x2, div_[q], dnm[q]), ds_q[q]);
        }
      }
    };
    // a = arriving dd row (relative: image row r0-1+a), read one step after the DK waves wrote it
    auto step = [&](int a, float (&A)[EPC], float (&Bc)[EPC], float (&Cn)[EPC]) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // barrier of stage a + 1
      if (PRO && a + 2 < nsteps) xform(a + 2);
      const unsigned char* sb = lds + kDdOff + (a & 1) * kSub;
      const u32x4 vL = *reinterpret_cast<const u32x4*>(hasL ? sb + offC - pitch : lds + kZOff);
      const u32x4 vC = *reinterpret_cast<const u32x4*>(sb + offC);
      const u32x4 vR = *reinterpret_cast<const u32x4*>(hasR ? sb + offC + pitch : lds + kZOff);
      if (!(CRNN_DBS_EXP & 8)) {
        float f[EPC];
        widenE<EPC>(vL, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { A[e] = fmaf(f[e], kw[6][e], A[e]); Bc[e] = fmaf(f[e], kw[3][e], Bc[e]); Cn[e] = fmaf(f[e], kw[0][e], 0.f); }
        widenE<EPC>(vC, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { A[e] = fmaf(f[e], kw[7][e], A[e]); Bc[e] = fmaf(f[e], kw[4][e], Bc[e]); Cn[e] = fmaf(f[e], kw[1][e], Cn[e]); }
        widenE<EPC>(vR, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { A[e] = fmaf(f[e], kw[8][e], A[e]); Bc[e] = fmaf(f[e], kw[5][e], Bc[e]); Cn[e] = fmaf(f[e], kw[2][e], Cn[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { A[e] += __uint_as_float(vL[e]); A[(e + 4) % EPC] += __uint_as_float(vC[e]); Bc[e] += __uint_as_float(vR[e]); }
      }
      if (a >= 2 && act) {
        const u32x4 o = packE<EPC>(A);
        if (!(CRNN_DBS_EXP & 2)) *reinterpret_cast<u32x4*>(orow + (long)(a - 2) * p.rowbytes) = o;
#undef orow
        if (SW) *reinterpret_cast<u32x4*>(lds + kDxrOff + (a & 1) * kSub + offC) = o;   // (for the statistics wave, one step later)
        if (DXS) dxstats(A);
      }
      if (DXS && a >= 2) sslot = sslot + 1 == kNR ? 0 : sslot + 1;
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // step 0: the first dd row is being formed
    if (PRO) xform(1);
    step(0, X1, X2, X0);
    step(1, X2, X0, X1);
    const int last = p.HB + 1;                     // arriving rows 0 .. HB+1
    int a = 2;
    for (; a + 2 <= last; a += 3) {
      step(a, X0, X1, X2);
      step(a + 1, X1, X2, X0);
      step(a + 2, X2, X0, X1);
    }
    for (int r = 0; a <= last; ++a, ++r) {
      if (r == 0) step(a, X0, X1, X2);
      else step(a, X1, X2, X0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // ring free
    if (SW) __builtin_amdgcn_s_barrier();
    if (DXS) {
      // lanes l, l + cppw, ... of a wave hold the same channels -> xor shuffles; one record [2][channels] per DX wave in LDS
      for (int o = p.cppw; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < EPC / 2; ++e) {
          ds_s[e].x += __shfl_xor(ds_s[e].x, o, 64); ds_s[e].y += __shfl_xor(ds_s[e].y, o, 64);
          ds_q[e].x += __shfl_xor(ds_q[e].x, o, 64); ds_q[e].y += __shfl_xor(ds_q[e].y, o, 64);
        }
      }
      float* sred = reinterpret_cast<float*>(lds + LP::SredOff);      // [5 waves][2][pcw]
      if (lane < p.cppw) {
#pragma unroll
        for (int e = 0; e < EPC / 2; ++e) {
          *reinterpret_cast<f32x2_t*>(sred + (gidx * 2) * pcw + lane * EPC + 2 * e) = ds_s[e];
          *reinterpret_cast<f32x2_t*>(sred + (gidx * 2 + 1) * pcw + lane * EPC + 2 * e) = ds_q[e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (DXS) {
      const float ik = DROP ? 1.f / (1.f - p.rate) : 1.f;          // gy carries the dropout's 1 / (1 - rate): applied once here
      const float* sred = reinterpret_cast<const float*>(lds + LP::SredOff);
      float* prow = p.bn2_partials + (long)(blockIdx.x / p.nsplit) * 2 * p.C + c0;
      for (int i = gidx * 64 + lane; i < 2 * pcw; i += kCW * 64) {   // i < pcw: sum gy, else sum gy * xhat; the five waves in a fixed order
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < kCW; ++w) a += sred[w * 2 * pcw + i];
        prow[i < pcw ? i : p.C + i - pcw] = a * ik;
      }
    }
    return;
  }

  // -------------------------------------------------------------------- DK waves: dd of the arriving row, weight-gradient sums
  // (the 72 sums leave no room for the 32 BatchNorm constants of the lane's channels: they sit in LDS and are read per step; the dd
  // history is kept as packed bf16)
  const int cw = p.cppw * EPC;                      // channels of this workgroup
  float* cst = reinterpret_cast<float*>(lds + kCstOff);
  for (int i = gidx * 64 + lane; i < cw; i += kCW * 64) {
    const int ch = c0 + i;
    const float scv = p.bnstate[2 * p.C + ch];
    float P, Q;
    bn_bwd_pq(scv, p.coef[ch], p.coef[p.C + ch], p.bnstate[ch], 1.0f / sqrtf(p.bnstate[p.C + ch] + BN_EPS_F), P, Q);
    cst[i] = scv; cst[cw + i] = p.bnstate[3 * p.C + ch]; cst[2 * cw + i] = P; cst[3 * cw + i] = Q;
  }
  const float* cl = cst + oct * EPC;
  float dk[9][EPC];
  u32x4 H0 = (u32x4)(0u), H1 = (u32x4)(0u), H2 = (u32x4)(0u);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) dk[t][e] = 0.f;
  int slot = 0;
  // step s: Dn <- dd of row s (relative: image row r0-1+s); xin row s-1 (image row r0-2+s) meets Dn (taps 0..2), Dm1 (3..5), Dm2 (6..8)
  auto step = [&](int s, u32x4& Dn, const u32x4& Dm1, const u32x4& Dm2, bool edge) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* sb = lds + slot * kStageB;
    slot = slot + 1 == kNR ? 0 : slot + 1;
    if (!edge || s <= p.HB + 1) {
      const u32x4 vd = *reinterpret_cast<const u32x4*>(sb + offC);
      const u32x4 vg = *reinterpret_cast<const u32x4*>(sb + kSub + offC);
      float xv[EPC], gv[EPC], r[EPC];
      widenE<EPC>(vd, xv); widenE<EPC>(vg, gv);
#pragma unroll
      for (int hq = 0; hq < EPC / 4; ++hq) {
        const float4 s4 = *reinterpret_cast<const float4*>(cl + 4 * hq), h4 = *reinterpret_cast<const float4*>(cl + cw + 4 * hq);
        const float4 p4 = *reinterpret_cast<const float4*>(cl + 2 * cw + 4 * hq), q4 = *reinterpret_cast<const float4*>(cl + 3 * cw + 4 * hq);
        const float scq[4] = {s4.x, s4.y, s4.z, s4.w}, shq[4] = {h4.x, h4.y, h4.z, h4.w}, pq[4] = {p4.x, p4.y, p4.z, p4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tv = fmaf(xv[4 * hq + e], scq[e], shq[e]);
          const float gy = (tv > 0.f && tv < 6.f) ? gv[4 * hq + e] : 0.f;
          r[4 * hq + e] = bn_bwd_dx_pq(xv[4 * hq + e], gy, scq[e], pq[e], qq[e]);
        }
      }
      u32x4 w = packE<EPC>(r);                               // (bf16: rounded as the stored tensor would be)
      bool band = act;
      if (edge) {
        const int g = r0 - 1 + s;
        if (g < 0 || g >= p.H) w = (u32x4)(0u);
        band = act && s >= 1 && s <= p.HB;                     // halo rows feed dx only: their weight-gradient terms belong to the neighbour band
      }
      if (act) *reinterpret_cast<u32x4*>(lds + kDdOff + (s & 1) * kSub + offC) = w;
      Dn = band ? w : (u32x4)(0u);
    } else {
      Dn = (u32x4)(0u);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!edge || s >= 1) {
      const unsigned char* xb = PRO ? lds + kXtOff + (s & 1) * kSub : sb + 2 * kSub;
      u32x4 vL = *reinterpret_cast<const u32x4*>(hasL ? xb + offC - pitch : lds + kZOff);
      u32x4 vC = *reinterpret_cast<const u32x4*>(xb + offC);
      u32x4 vR = *reinterpret_cast<const u32x4*>(hasR ? xb + offC + pitch : lds + kZOff);
      if (edge) {
        const int g = r0 - 2 + s;
        if (g < 0 || g >= p.H) { vL = (u32x4)(0u); vC = (u32x4)(0u); vR = (u32x4)(0u); }
      }
      if (!(CRNN_DBS_EXP & 4)) {
        float f[EPC], dn[EPC], dm1[EPC], dm2[EPC];
        widenE<EPC>(Dn, dn); widenE<EPC>(Dm1, dm1); widenE<EPC>(Dm2, dm2);
        widenE<EPC>(vL, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { dk[0][e] = fmaf(f[e], dn[e], dk[0][e]); dk[3][e] = fmaf(f[e], dm1[e], dk[3][e]); dk[6][e] = fmaf(f[e], dm2[e], dk[6][e]); }
        widenE<EPC>(vC, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { dk[1][e] = fmaf(f[e], dn[e], dk[1][e]); dk[4][e] = fmaf(f[e], dm1[e], dk[4][e]); dk[7][e] = fmaf(f[e], dm2[e], dk[7][e]); }
        widenE<EPC>(vR, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { dk[2][e] = fmaf(f[e], dn[e], dk[2][e]); dk[5][e] = fmaf(f[e], dm1[e], dk[5][e]); dk[8][e] = fmaf(f[e], dm2[e], dk[8][e]); }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { dk[0][e] += __uint_as_float(vL[e] + Dn[e]); dk[1][e] += __uint_as_float(vC[e] + Dm1[e]); dk[2][e] += __uint_as_float(vR[e] + Dm2[e]); }
      }
    }
  };
  step(0, H0, H2, H1, true);
  step(1, H1, H0, H2, true);
  int s = 2;
  for (; s + 2 <= p.HB; s += 3) {                   // rows 2 .. HB need no edge handling
    step(s, H2, H1, H0, false);
    step(s + 1, H0, H2, H1, false);
    step(s + 2, H1, H0, H2, false);
  }
  for (; s < nsteps; ++s) {
    const int m = s % 3;
    if (m == 2) step(s, H2, H1, H0, true);
    else if (m == 0) step(s, H0, H2, H1, true);
    else step(s, H1, H0, H2, true);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // ring free
  if (SW) __builtin_amdgcn_s_barrier();            // (... once the statistics wave has taken its last row out of it)
  // weight-gradient partials of the workgroup: lanes l, l + cppw, ... of a wave hold the same channels -> xor shuffles; the five waves
  // through LDS in a fixed order
  float* red = reinterpret_cast<float*>(lds);       // [5 waves][9][cppw * 8]
  for (int o = p.cppw; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < EPC; ++e) dk[t][e] += __shfl_xor(dk[t][e], o, 64);
  }
  // the lane index is formed again HERE (round 6): kept live from the top of the kernel across the row loop -- whose 72 sums, taps and running rows sit at the
  // 168-register ceiling -- it was one of three values the allocator spilled to scratch (16 bytes per lane) and reloaded for these few lines
  // (the prologue forms keep other values across the loop; re-forming the index costs them a register more: measured with -Rpass-analysis)
  int lane_e = lane;
  if constexpr (!PRO) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
  if (lane_e < p.cppw) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float* dst = red + (gidx * 9 + t) * cw + lane_e * EPC;
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) *reinterpret_cast<float4*>(dst + 4 * h) = make_float4(dk[t][4 * h], dk[t][4 * h + 1], dk[t][4 * h + 2], dk[t][4 * h + 3]);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    // the DK waves' threads, numbered 0 .. 319 by (group index, lane)
    const int t5 = gidx * 64 + lane_e;
    const long prow = (long)(blockIdx.x / p.nsplit) * 9 * p.C;
    for (int i = t5; i < 9 * cw; i += kCW * 64) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < kCW; ++w) a += red[w * 9 * cw + i];
      const int t = i / cw, c = i - t * cw;
      p.partials[prow + (long)t * p.C + c0 + c] = a;
    }
  }
}

struct DbsGeom { int nsplit, nwgb, HB, cols, cppw; bool ok; };
DbsGeom dbs_geom(int B, int H, int W, int C, int epc = 8) {   // epc: elements per 16-byte chunk (8 bf16 | 4 fp32)
  DbsGeom g; g.ok = false; g.nsplit = g.nwgb = g.HB = g.cols = g.cppw = 0;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return g;
  const long cols1 = (long)W * C / epc;
  int ns = 0;
  for (int n = 1; n <= C / epc; ++n) {
    if ((C / epc) % n) continue;
    if (cols1 / n <= kCW * 64) { ns = n; break; }
  }
  if (!ns) return g;
  const int cols = (int)(cols1 / ns), cppw = C / epc / ns;
  // more than two of the five waves of a group busy (round 5: was more than four -- image width 48 gives ranges of 208 columns, which used to send the
  // whole depthwise stage back to the three-kernel tile path; 65 % of the lanes busy is still the faster schedule); power-of-two columns per pixel
  // below a wave (shuffle reduction)
  if (cols <= 2 * 64 || (cppw & (cppw - 1)) || cppw > 32) return g;
  int nwgb = 1;
#ifndef CRNN_DBS_WGS
#define CRNN_DBS_WGS 256
#endif
  for (int n = 1; n <= H; ++n) {
    if (H % n || H / n < 8) continue;
    nwgb = n;
    if ((long)B * n * ns >= CRNN_DBS_WGS) break;
  }
  if (H % nwgb) return g;
  g.nsplit = ns; g.nwgb = nwgb; g.HB = H / nwgb; g.cols = cols; g.cppw = cppw; g.ok = true;
  return g;
}

}  // namespace

// CRNN_OK when crnn_dwconv3x3_bwd_stream takes the shape (W * C / 8 sixteen-byte columns split over whole channel octets into
// workgroups of 129..320 columns: every block of the CRNN at image widths 32, 48, 64), else CRNN_ERR_UNSUPPORTED -> crnn_dwconv3x3_bwd_fused.
extern "C" int crnn_dwconv_bwd_stream_supported(int B, int H, int W, int C) { return dbs_geom(B, H, W, C).ok ? CRNN_OK : CRNN_ERR_UNSUPPORTED; }
// rows of [9][C] weight-gradient partials the launch writes (scratch = rows * 9 * C floats)
extern "C" int crnn_dwconv_bwd_stream_rows(int B, int H, int W, int C) { DbsGeom g = dbs_geom(B, H, W, C); return g.ok ? B * g.nwgb : 0; }
// Same contract as crnn_dwconv3x3_bwd_fused (d, da, xin, dx: bf16 [B,H,W,C]; bnstate = [mean|var|scale|shift]; coef = [c1|c2];
// k [9][C]; dk [9][C] out); dx bit-identical, dk to the summation order of its partial sums.
extern "C" int crnn_dwconv3x3_bwd_stream(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k,
                                         void* dx, float* dk, float* scratch, int B, int H, int W, int C, hipStream_t stream) {
  if (!d || !da || !bnstate || !coef || !xin || !k || !dx || !dk || !scratch) return CRNN_ERR_ARG;
  const DbsGeom g = dbs_geom(B, H, W, C);
  if (!g.ok) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)d | (uintptr_t)da | (uintptr_t)xin | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)k) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 2 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DbsParams p;
  p.d = (const unsigned char*)d; p.da = (const unsigned char*)da; p.xin = (const unsigned char*)xin; p.dx = (unsigned char*)dx;
  p.bnstate = bnstate; p.coef = coef; p.k = k; p.partials = scratch;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.nwgb = g.nwgb; p.nsplit = g.nsplit; p.cols = g.cols; p.cppw = g.cppw; p.rowbytes = W * C * 2;
  p.pro_bn = nullptr; p.keep = nullptr; p.rate = 0.f; p.bn2_partials = nullptr;
  CRNN_LDS_ATTR((dw_bwd_stream_kernel<kD, false, false>), DbsLds<kD>::XtOff);
  hipLaunchKernelGGL((dw_bwd_stream_kernel<kD, false, false>), dim3(B * g.nwgb * g.nsplit), dim3(704), DbsLds<kD>::XtOff, stream, p);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, B * g.nwgb, 9 * C, dk, 1.f, stream);
}
// The same stage on fp32 tensors (dtype CRNN_F32; CRNN_BF16 = the entry points above): four channels per lane, W * C / 4 columns split into
// workgroups of 129..320 -- the parity mode's crnn_bn_bwd_apply_ex + crnn_dwconv3x3_wgrad_ex + crnn_dwconv3x3_fwd_ex(flip) in one pass over
// d, da, xin (4 tensor passes instead of 7); dx bit-identical to that sequence, dk to the summation order of its partial sums.
extern "C" int crnn_dwconv_bwd_stream_supported_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_bwd_stream_supported(B, H, W, C);
  if (dtype != CRNN_F32) return CRNN_ERR_ARG;
  return dbs_geom(B, H, W, C, 4).ok ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dwconv_bwd_stream_rows_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_bwd_stream_rows(B, H, W, C);
  DbsGeom g = dbs_geom(B, H, W, C, 4); return (dtype == CRNN_F32 && g.ok) ? B * g.nwgb : 0;
}
extern "C" int crnn_dwconv3x3_bwd_stream_ex(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k,
                                            void* dx, float* dk, float* scratch, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  if (dtype == CRNN_BF16) return crnn_dwconv3x3_bwd_stream(d, da, bnstate, coef, xin, k, dx, dk, scratch, B, H, W, C, stream);
  if (dtype != CRNN_F32 || !d || !da || !bnstate || !coef || !xin || !k || !dx || !dk || !scratch) return CRNN_ERR_ARG;
  const DbsGeom g = dbs_geom(B, H, W, C, 4);
  if (!g.ok) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)d | (uintptr_t)da | (uintptr_t)xin | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)k) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 4 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DbsParams p;
  p.d = (const unsigned char*)d; p.da = (const unsigned char*)da; p.xin = (const unsigned char*)xin; p.dx = (unsigned char*)dx;
  p.bnstate = bnstate; p.coef = coef; p.k = k; p.partials = scratch;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.nwgb = g.nwgb; p.nsplit = g.nsplit; p.cols = g.cols; p.cppw = g.cppw; p.rowbytes = W * C * 4;
  p.pro_bn = nullptr; p.keep = nullptr; p.rate = 0.f; p.bn2_partials = nullptr;
  CRNN_LDS_ATTR((dw_bwd_stream_kernel<kD, false, false, false, true>), DbsLds<kD>::XtOff);
  hipLaunchKernelGGL((dw_bwd_stream_kernel<kD, false, false, false, true>), dim3(B * g.nwgb * g.nsplit), dim3(704), DbsLds<kD>::XtOff, stream, p);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, B * g.nwgb, 9 * C, dk, 1.f, stream);
}
// Prologue form: xin = q of the previous block ([B,H,W,C] bf16), pro_bnstate = its BatchNorm-2 state, keep = the keep bytes of the dropout site of
// its output (crnn_dropout_keep_bytes; NULL allowed when rate == 0): x = Dropout(ReLU6(q * scale + shift)) is re-formed in LDS instead of read.  dx / dk bit-identical to
// crnn_dwconv3x3_bwd_stream on the materialised x.  Same shape rule + fewer than 2^32 dropout groups (B*H*W*C/8).
extern "C" int crnn_dwconv_bwd_stream_pro_supported(int B, int H, int W, int C) {
  const DbsGeom g = dbs_geom(B, H, W, C);       // (+ the keep bytes of a stage travel as whole dwords: 4 columns of one pixel)
  return (g.ok && g.cppw % 4 == 0 && g.cols % 4 == 0 && (W * (C / 8)) % 4 == 0 && (long)B * H * W * (C / 8) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
// bn2_stat_partials != NULL: [crnn_dwconv_bwd_stream_rows][2][C] partial sums (sum gy | sum gy * xhat) of the PRODUCER's BatchNorm-2 backward, gy = dx routed
// through the dropout mask and the ReLU6 gate of q -- the statistics pass of crnn_bn_bwd_ex(q, dx, pro_bnstate, ..., rate, seed, layer) (finish with
// crnn_bn_bwd_finalize, then crnn_bn_bwd_apply_ex); the same sums in another order.
extern "C" int crnn_dwconv3x3_bwd_stream_pro(const void* d, const void* da, const float* bnstate, const float* coef, const void* q, const float* pro_bnstate,
                                             float rate, const void* keep, const float* k, void* dx, float* dk, float* scratch, float* bn2_stat_partials,
                                             int B, int H, int W, int C, hipStream_t stream) {
  if (!d || !da || !bnstate || !coef || !q || !pro_bnstate || !k || !dx || !dk || !scratch || rate < 0.f || rate >= 1.f || (rate > 0.f && !keep)) return CRNN_ERR_ARG;
  const DbsGeom g = dbs_geom(B, H, W, C);
  if (crnn_dwconv_bwd_stream_pro_supported(B, H, W, C) != CRNN_OK) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)d | (uintptr_t)da | (uintptr_t)q | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)k | (uintptr_t)pro_bnstate) & 15) || ((uintptr_t)keep & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 2 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DbsParams p;
  p.d = (const unsigned char*)d; p.da = (const unsigned char*)da; p.xin = (const unsigned char*)q; p.dx = (unsigned char*)dx;
  p.bnstate = bnstate; p.coef = coef; p.k = k; p.partials = scratch;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.nwgb = g.nwgb; p.nsplit = g.nsplit; p.cols = g.cols; p.cppw = g.cppw; p.rowbytes = W * C * 2;
  p.pro_bn = pro_bnstate; p.keep = (const unsigned char*)keep; p.rate = rate; p.bn2_partials = bn2_stat_partials;
  constexpr int KD = kD + 1;
  const dim3 grid(B * g.nwgb * g.nsplit);
#define DBS_PRO_LAUNCH(DROP, STATS)                                                                                     \
  do {                                                                                                                  \
    CRNN_LDS_ATTR((dw_bwd_stream_kernel<KD, true, DROP, STATS>), (DbsLds<KD, STATS>::Total));                           \
    hipLaunchKernelGGL((dw_bwd_stream_kernel<KD, true, DROP, STATS>), grid, dim3(STATS ? 768 : 704), (DbsLds<KD, STATS>::Total), stream, p); \
  } while (0)
  if (bn2_stat_partials) { if (rate > 0.f) DBS_PRO_LAUNCH(true, true); else DBS_PRO_LAUNCH(false, true); }
  else { if (rate > 0.f) DBS_PRO_LAUNCH(true, false); else DBS_PRO_LAUNCH(false, false); }
#undef DBS_PRO_LAUNCH
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, B * g.nwgb, 9 * C, dk, 1.f, stream);
}
// The prologue form by storage type (CRNN_BF16 = the entry points above; CRNN_F32, round 4 -- the parity mode): d, da, q, dx fp32, four channels per
// lane, the keep bytes (one per 8 elements) as nibbles.  dx / dk / the statistics partials as for the bf16 form, against the fp32 sequence
// crnn_bn_act_pool_drop_ex(q -> x) + crnn_dwconv3x3_bwd_stream_ex(x) [+ crnn_bn_bwd_ex's statistics pass].
extern "C" int crnn_dwconv_bwd_stream_pro_supported_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_bwd_stream_pro_supported(B, H, W, C);
  if (dtype != CRNN_F32) return CRNN_ERR_ARG;
  const DbsGeom g = dbs_geom(B, H, W, C, 4);   // (+ the keep bytes of a stage travel as whole dwords: 8 columns of one pixel)
  return (g.ok && g.cppw % 8 == 0 && g.cols % 8 == 0 && (long)B * H * W * (C / 8) < (1L << 31)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dwconv3x3_bwd_stream_pro_ex(const void* d, const void* da, const float* bnstate, const float* coef, const void* q, const float* pro_bnstate,
                                                float rate, const void* keep, const float* k, void* dx, float* dk, float* scratch, float* bn2_stat_partials,
                                                int B, int H, int W, int C, int dtype, hipStream_t stream) {
  if (dtype == CRNN_BF16) return crnn_dwconv3x3_bwd_stream_pro(d, da, bnstate, coef, q, pro_bnstate, rate, keep, k, dx, dk, scratch, bn2_stat_partials, B, H, W, C, stream);
  if (dtype != CRNN_F32 || !d || !da || !bnstate || !coef || !q || !pro_bnstate || !k || !dx || !dk || !scratch || rate < 0.f || rate >= 1.f || (rate > 0.f && !keep)) return CRNN_ERR_ARG;
  const DbsGeom g = dbs_geom(B, H, W, C, 4);
  if (crnn_dwconv_bwd_stream_pro_supported_ex(B, H, W, C, CRNN_F32) != CRNN_OK) return CRNN_ERR_UNSUPPORTED;
  if ((((uintptr_t)d | (uintptr_t)da | (uintptr_t)q | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)k | (uintptr_t)pro_bnstate) & 15) || ((uintptr_t)keep & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 4 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DbsParams p;
  p.d = (const unsigned char*)d; p.da = (const unsigned char*)da; p.xin = (const unsigned char*)q; p.dx = (unsigned char*)dx;
  p.bnstate = bnstate; p.coef = coef; p.k = k; p.partials = scratch;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.nwgb = g.nwgb; p.nsplit = g.nsplit; p.cols = g.cols; p.cppw = g.cppw; p.rowbytes = W * C * 4;
  p.pro_bn = pro_bnstate; p.keep = (const unsigned char*)keep; p.rate = rate; p.bn2_partials = bn2_stat_partials;
  constexpr int KD = kD + 1;
  const dim3 grid(B * g.nwgb * g.nsplit);
#define DBS_PRO_LAUNCH(DROP, STATS)                                                                                     \
  do {                                                                                                                  \
    CRNN_LDS_ATTR((dw_bwd_stream_kernel<KD, true, DROP, STATS, true>), (DbsLds<KD, STATS, STATS>::Total));              \
    hipLaunchKernelGGL((dw_bwd_stream_kernel<KD, true, DROP, STATS, true>), grid, dim3(704), (DbsLds<KD, STATS, STATS>::Total), stream, p); \
  } while (0)
  if (bn2_stat_partials) { if (rate > 0.f) DBS_PRO_LAUNCH(true, true); else DBS_PRO_LAUNCH(false, true); }
  else { if (rate > 0.f) DBS_PRO_LAUNCH(true, false); else DBS_PRO_LAUNCH(false, false); }
#undef DBS_PRO_LAUNCH
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, B * g.nwgb, 9 * C, dk, 1.f, stream);
}
