// Weights-resident bf16 GEMM for the pointwise (1x1) convolutions (utils.py:49, Conv2D(1x1)):
//     Y[M][N] = X[M][K] . W[N][K]^T        X = pixels x channels (bf16, NHWC rows), W bf16 [N][K], Y bf16, fp32 accumulate
// M is 10^5..10^6 pixels, N and K are 64..512 channels: the weight matrix is at most 512 KB, a CU's register file is
// 512 KB.  The ring kernel (gemm_nt.hip) re-streams the weight rows of its pass through LDS for every 128-pixel stripe -- 2x
// to 4x the pixel bytes -- and that L2 -> LDS traffic, not HBM and not the MFMAs, is what it runs at.  Here the weights never
// move again after the prologue:
//   * a workgroup owns a SLICE of 128 output channels; each of its four MFMA waves keeps 32 channels x K of W as
//     v_mfma_f32_32x32x16_bf16 A-operand fragments in registers for the whole launch (K/4 VGPRs: 128 at K = 512);
//   * the only stream is the pixels: one persistent workgroup per CU walks over 128-pixel stripes, two LOADER waves feed
//     an LDS ring of R = 6 stages (128 pixels x 64 k = 16 KiB each) with global_load_lds (16 B per lane, no VGPR staging) under
//     counted s_waitcnt vmcnt(N); up to 4 stages = 64 KiB per CU are in flight (16 MB over the chip against the ~8 MB that
//     8 TB/s x 1 us of latency need);
//   * the N/128 slices of one stripe run at the same time on CUs of the SAME XCD (workgroup id -> xcd = id % 8), so the
//     stripe is read from HBM once and from that XCD's L2 by the other slices;
//   * pixels are the MFMA B operand (D = W X^T): a lane holds 4 consecutive output channels of one pixel per register
//     group, v_permlane32_swap pairs groups into 8 channels;
//   * the MFMA waves do not store to global memory: issuing the 8 result stores of a stripe costs a wave ~0.8 us (the
//     texture-address path takes ~100 cycles per 1-KiB piece next to the loaders' traffic), as much as three stages of
//     MFMAs.  They put the bf16 stripe (128 pixels x 128 channels, 32 KiB, XOR-swizzled 16-byte chunks) into one of two LDS
//     staging tiles with eight ds_write_b128 and go on; two STORER waves drain the tile to global memory in pieces spread
//     over the stages of the next stripe, as fully coalesced 256-byte rows;
//   * one raw s_barrier per stage; the loaders guarantee stage i + 1 (not just i) at barrier i, so a compute wave reads the
//     first fragments of the next stage before it reaches the next barrier and the MFMA pipe does not drain at stage edges.
// LDS rows are 128 B unpadded, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (applied to the per-lane global
// address; the LDS image stays lane-linear): conflict-free ds_read_b128 fragment reads.
// Numerics: the same MFMA and the same k order (64-k chunks ascending, 16-k steps ascending) as gemm_nt.hip / gemm_bf16.inc --
// results are bit-identical.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct WresParams {
  const bf16_t* X; const bf16_t* W; bf16_t* Y;
  int M, N, K;
  int stripes;   // ceil(M / 128)
  int S;         // channel slices of 128 (N / 128)
  int Q;         // stripes processed concurrently per XCD (workgroups per XCD / S)
  int nxcd;      // XCDs the grid spans (grid = nxcd * Q * S)
  const float* cscale; const float* cshift;   // EPI instantiation: per-output-channel epilogue Y = ReLU6(acc * cscale[n] + cshift[n])
  // BNS instantiation (Y = the gradient w.r.t. a = ReLU6(BN(d)) of a depthwise-separable block): the storer waves also take the statistics
  // of that BatchNorm's backward pass, sum(gy) and sum(gy * xhat) per channel with gy = Y where 0 < d * scale + shift < 6, from the
  // staged stripe and the matching rows of d -- the stand-alone statistics pass (bn_bwd_kernel<1>) would read Y and d again
  const bf16_t* D; const float* bnstate;      // d [M][N]; [mean | var | scale | shift] x N
  float* stats;                               // [2 * Q * nxcd][2][N] partial sums, one row per storer wave and stripe lane
#ifdef CRNN_WRES_EXP
  unsigned long long* trace;   // ablation build only: [64 iterations][4] s_memrealtime stamps of workgroup 0's first loader wave
  int exp;       // unused (the ablation mask is the compile-time value of CRNN_WRES_EXP: 1 no pixel loads, 2 no fragment reads, 8 no MFMAs, 4 no stores, 32 free-running loaders only)
#endif
};
#ifdef CRNN_WRES_EXP   // compile-time ablation mask: run-time tests between the MFMAs would change what is being measured
#define WRES_EXP(p, bit) ((CRNN_WRES_EXP) & (bit))
#else
#define WRES_EXP(p, bit) 0
#endif

constexpr int kStage = 128 * 128;           // 128 pixel rows x 64 k x 2 B
constexpr int kR = 6;                       // ring stages
constexpr int kOut = 128 * 256;             // one staging tile: 128 pixels x 128 channels x 2 B

#ifndef CRNN_WRESF_NT
#define CRNN_WRESF_NT 1    // round 5: the training forward reads d with nontemporal loads (not read again before the backward pass; cache policy only)
#endif
// (round 5: the data-gradient form reading dq -- its last use in the step -- with nontemporal LDS-DMA was measured: no effect on the step; plain DMA stays)
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The MFMA role shared by the kernels below: wave `wave` (0..3) of a workgroup keeps rows n0 .. n0+31 of W [N][K] as A-operand
// fragments, multiplies them with every 128-pixel stage the producers put into the ring (R slots at smem), and leaves each
// finished stripe as bf16 in the staging tile outs[stripe & 1].  Barrier protocol: one s_barrier before stage 0, one after every
// stage; at barrier i the producers have stages i and i+1 in LDS.
// EPI: the result leaves as ReLU6(acc * scale[ch] + shift[ch]) (inference: the BatchNorm after the convolution folded in, one rounding
// to bf16 as in the tile GEMM's epilogue); epi = LDS table [scale | shift] of the workgroup's 128 CB channels.
template <int KCH, int R, int CB = 1, int PB = 4, bool EPI = false>   // CB 32-channel blocks per wave, PB 32-pixel blocks per stage
__device__ __forceinline__ void wres_compute(const unsigned char* smem, unsigned char* outs, const bf16_t* __restrict__ W, int K, int n0,
                                             int wave, int lane, int mine, const float* epi = nullptr) {
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
  constexpr int stage = PB * 32 * 128, orow = 256 * CB;      // bytes of a ring stage / of a staging-tile row (128 CB channels)
  bf16x8_t wf[CB][KCH][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const bf16_t* wrow = W + (long)(n0 + 32 * cb + l31) * K + half * 8;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[cb][kc][ks] = *reinterpret_cast<const bf16x8_t*>(wrow + kc * 64 + ks * 16);
  }
  f32x16 acc[PB][CB];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int lrow = l31 * 128;
  int slot = 0;
  bf16x8_t fx[PB];                                              // fragments of k-step 0 of the stage about to be multiplied
  __builtin_amdgcn_s_barrier();                                // barrier 0: stages 0 and 1 have landed
  {
    const unsigned char* A = smem + lrow + ((half ^ sw) * 16);
#pragma unroll
    for (int b = 0; b < PB; ++b) fx[b] = *reinterpret_cast<const bf16x8_t*>(A + b * 32 * 128);
  }
  for (int it = 0; it < mine; ++it) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      const unsigned char* A = smem + slot * stage + lrow;
      slot = slot + 1 == R ? 0 : slot + 1;
      const unsigned char* An = smem + slot * stage + lrow;    // next stage (landed: the loaders run one stage ahead of the barrier)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t nx[PB];
        const unsigned char* src = ks < 3 ? A + (((2 * (ks + 1) + half) ^ sw) * 16) : An + ((half ^ sw) * 16);
#pragma unroll
        for (int b = 0; b < PB; ++b) if (!WRES_EXP(p, 2)) nx[b] = *reinterpret_cast<const bf16x8_t*>(src + b * 32 * 128);
        // pin the order "reads of the next k-step, then this k-step's MFMAs": left alone the scheduler shares two fragment
        // registers between all reads and the MFMA pipe waits out an LDS round trip every second MFMA
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            if (WRES_EXP(p, 8)) continue;
            if (kc == 0 && ks == 0) acc[b][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][kc][ks], fx[b], zero16, 0, 0, 0);   // C = 0: no clearing pass
            else acc[b][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][kc][ks], fx[b], acc[b][cb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < PB; ++b) fx[b] = nx[b];
      }
      if (kc + 1 < KCH) __builtin_amdgcn_s_barrier();          // releases this stage's slot; the stage after the next has landed
    }
    // ---- epilogue of the stripe: lane = one pixel; register group g of a 32x32 block = channels 8g + 4half + 0..3.
    // 16-byte chunk (4 (wave CB + cb) + 2 pr + half) of pixel row px goes to position chunk ^ (px & 15) of the staging tile
    unsigned char* ob = outs + (it & 1) * (PB * 32 * orow);
    if constexpr (EPI) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cs0 = wave * 32 * CB + 32 * cb + 8 * g + 4 * half;
          const float4 sv = *reinterpret_cast<const float4*>(epi + cs0), tv = *reinterpret_cast<const float4*>(epi + 128 * CB + cs0);
          const float ss[4] = {sv.x, sv.y, sv.z, sv.w}, tt[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
          for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[b][cb][4 * g + e] = relu6f(fmaf(acc[b][cb][4 * g + e], ss[e], tt[e]));
        }
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const int px = 32 * b + l31;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const f32x16& v = acc[b][cb];
          unsigned g0a = pack2_bf16(v[8 * pr + 0], v[8 * pr + 1]), g0b = pack2_bf16(v[8 * pr + 2], v[8 * pr + 3]);
          unsigned g1a = pack2_bf16(v[8 * pr + 4], v[8 * pr + 5]), g1b = pack2_bf16(v[8 * pr + 6], v[8 * pr + 7]);
          const u32x2 sa = __builtin_amdgcn_permlane32_swap(g0a, g1a, false, false);
          const u32x2 sb = __builtin_amdgcn_permlane32_swap(g0b, g1b, false, false);
          *reinterpret_cast<uint4*>(ob + px * orow + (((4 * (wave * CB + cb) + 2 * pr + half) ^ (px & 15)) * 16)) = make_uint4(sa.x, sb.x, sa.y, sb.y);
        }
    }
    // the stripe's last stage: also hands the staged tile to the storer / IO waves -- the raw barrier orders nothing by itself, the
    // tile's ds_writes must have completed before this wave arrives (the compiler does not wait for LDS stores at s_barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

// Storer wave of the BNS instantiation.  Same drain as below (16 pieces of 4 rows x 256 bytes per wave and stripe, a share of them after
// every stage of the next stripe), plus: the matching 16-byte chunks of d are loaded two piece steps ahead, and each drained chunk adds
// its 8 channels' gy and gy * xhat to per-lane sums kept over the whole launch (a lane always sees the same 8 channels).  The steady
// state is straight-line (M % 128 == 0: unguarded stores; loads past the end re-read a valid row) so that the wait-count insertion can
// count the loads in flight across the stores.  One partial row per storer wave: the same sums as bn_bwd_kernel<1> in another order.
template <int KCH>
__device__ __forceinline__ void wres_store_bnstats(const WresParams& p, const unsigned char* outs, int sw, int lane, int slice, int first, int step,
                                                   int mine, long srow) {
  constexpr int PP = 16 / KCH;                                 // pieces per storer wave and stage
  constexpr int PF = 2;                                        // piece steps the loads of d run ahead
  const int rsub = lane >> 4, c = lane & 15;
  const int ch0 = slice * 128 + c * 8;
  const int total = mine * KCH, nps = total - KCH;             // piece steps of the steady state (stripes 0 .. mine-2)
  float sc[8], sh[8], mu[8], inv[8], ss[8], qq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = p.bnstate[ch0 + e];
    inv[e] = 1.0f / sqrtf(p.bnstate[p.N + ch0 + e] + 1e-3f);   // BN_EPS, the spelling of bn_bwd_kernel
    sc[e] = p.bnstate[2 * p.N + ch0 + e]; sh[e] = p.bnstate[3 * p.N + ch0 + e];
    ss[e] = 0.f; qq[e] = 0.f;
  }
  auto row_of = [&](int stripe, int piece) { return (long)(first + stripe * step) * 128 + (sw * 16 + piece) * 4 + rsub; };
  auto load_d = [&](int ps, u32x4 (&xv)[PP]) {
    ps = ps < nps ? ps : (nps > 0 ? nps - 1 : 0);              // past the end: a valid address, the data is not used
    const int stripe = ps / KCH, t = ps % KCH;
#pragma unroll
    for (int u = 0; u < PP; ++u) xv[u] = *reinterpret_cast<const u32x4*>(p.D + row_of(stripe, t * PP + u) * p.N + ch0);
  };
  auto accum = [&](const u32x4& gv, const u32x4& xv) {
#pragma clang fp contract(off)
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int e = 2 * w + hf;
        const float xx = __uint_as_float(hf ? (xw[w] & 0xffff0000u) : (xw[w] << 16));
        const float gg = __uint_as_float(hf ? (gw[w] & 0xffff0000u) : (gw[w] << 16));
        const float t = fmaf(xx, sc[e], sh[e]);
        const float gy = (t > 0.f && t < 6.f) ? gg : 0.f;
        const float xh = (xx - mu[e]) * inv[e];
        ss[e] += gy;
        qq[e] = fmaf(gy, xh, qq[e]);
      }
  };
  auto piece_step = [&](int stripe, int t, const u32x4 (&xv)[PP]) {   // pieces t*PP .. of the finished stripe `stripe`
    const unsigned char* ob = outs + (stripe & 1) * kOut;
    u32x4 v[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const int r = (sw * 16 + t * PP + u) * 4 + rsub;
      v[u] = *reinterpret_cast<const u32x4*>(ob + r * 256 + ((c ^ (r & 15)) * 16));
    }
#pragma unroll
    for (int u = 0; u < PP; ++u) *reinterpret_cast<u32x4*>(p.Y + row_of(stripe, t * PP + u) * p.N + ch0) = v[u];
#pragma unroll
    for (int u = 0; u < PP; ++u) accum(v[u], xv[u]);
  };
  u32x4 xb[PF][PP];
#pragma unroll
  for (int k = 0; k < PF; ++k) load_d(k, xb[k]);
  for (int jb = 0; jb < KCH; ++jb) __builtin_amdgcn_s_barrier();   // barriers 0 .. KCH-1: no stripe has finished yet
  int ps = 0;
  for (; ps + PF <= nps; ps += PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      __builtin_amdgcn_s_barrier();
      // a stage's arithmetic stays in its stage: hoisted above the barrier (it does not depend on the staged tile) the next stage's
      // use of d would wait for loads issued one stage ago instead of two
      __builtin_amdgcn_sched_barrier(0);
      piece_step((ps + k) / KCH, (ps + k) % KCH, xb[k]);
      load_d(ps + k + PF, xb[k]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ps < nps) {                                              // nps odd: one more steady stage, its chunks of d are in buffer 0
    __builtin_amdgcn_s_barrier();
    piece_step(ps / KCH, ps % KCH, xb[0]);
  }
  __builtin_amdgcn_s_barrier();                                // barrier `total`: the last stripe is staged; everything at once
  for (int t = 0; t < KCH; ++t) {
    u32x4 xv[PP];
    const int stripe = mine - 1;
#pragma unroll
    for (int u = 0; u < PP; ++u) xv[u] = *reinterpret_cast<const u32x4*>(p.D + row_of(stripe, t * PP + u) * p.N + ch0);
    piece_step(stripe, t, xv);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ss[e] += __shfl_xor(ss[e], 16, 64); qq[e] += __shfl_xor(qq[e], 16, 64);
    ss[e] += __shfl_xor(ss[e], 32, 64); qq[e] += __shfl_xor(qq[e], 32, 64);
  }
  if (lane < 16) {
    float* r0 = p.stats + (srow * 2 + 0) * p.N + ch0;
    float* r1 = p.stats + (srow * 2 + 1) * p.N + ch0;
    *reinterpret_cast<float4*>(r0) = make_float4(ss[0], ss[1], ss[2], ss[3]);
    *reinterpret_cast<float4*>(r0 + 4) = make_float4(ss[4], ss[5], ss[6], ss[7]);
    *reinterpret_cast<float4*>(r1) = make_float4(qq[0], qq[1], qq[2], qq[3]);
    *reinterpret_cast<float4*>(r1 + 4) = make_float4(qq[4], qq[5], qq[6], qq[7]);
  }
}

template <int KCH, int NLW, bool EPI = false, bool BNS = false, int POOL = 1>   // K / 64, loader waves, folded-BatchNorm epilogue, BatchNorm-backward statistics, pooled rows
__global__ __launch_bounds__(384 + 64 * NLW) void gemm_wres_kernel(WresParams p) {
  constexpr int kIPS = 16 / NLW;            // LDS-DMA instructions per loader wave and stage
  constexpr int RR = EPI ? kR - 1 : kR;     // ring stages (the epilogue table takes the 160 KiB budget over: one stage less)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kR stages, then two output staging tiles
  unsigned char* const outs = smem + RR * kStage;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (xcd, slice, stripe lane): consecutive ids go to consecutive XCDs
  const int wg = blockIdx.x;
#ifdef CRNN_WRES_EXP
  const int per = gridDim.x / p.nxcd;
  const int x = WRES_EXP(p, 16) ? wg / per : wg % p.nxcd, j = WRES_EXP(p, 16) ? wg % per : wg / p.nxcd;
#else
  const int x = wg % p.nxcd, j = wg / p.nxcd;
#endif
  const int slice = j % p.S, q = j / p.S;
  const int step = p.Q * p.nxcd;                              // stripe stride between this workgroup's iterations
  const int first = q * p.nxcd + x;
  const int mine = first < p.stripes ? (p.stripes - first + step - 1) / step : 0;
  if (mine <= 0) {
    if constexpr (BNS) {                                      // no stripe: this workgroup's statistics rows must still read as zero
      if (wave >= 4 + NLW && lane < 16) {
        const long row = (long)(q * p.nxcd + x) * 2 + (wave - 4 - NLW);
        float4* r0 = reinterpret_cast<float4*>(p.stats + (row * 2 + 0) * p.N + slice * 128 + lane * 8);
        float4* r1 = reinterpret_cast<float4*>(p.stats + (row * 2 + 1) * p.N + slice * 128 + lane * 8);
        r0[0] = r0[1] = r1[0] = r1[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const int total = mine * KCH;
  const float* epi = nullptr;
  if constexpr (EPI) {                                        // the slice's epilogue scale | shift after the staging tiles
    float* tab = reinterpret_cast<float*>(outs + 2 * kOut);
    for (int i = tid; i < 128; i += 384 + 64 * NLW) { tab[i] = p.cscale[slice * 128 + i]; tab[128 + i] = p.cshift[slice * 128 + i]; }
    __syncthreads();
    epi = tab;
  }

  if constexpr (BNS) {
    if (wave >= 4 + NLW) { wres_store_bnstats<KCH>(p, outs, wave - 4 - NLW, lane, slice, first, step, mine, (q * p.nxcd + x) * 2 + (wave - 4 - NLW)); return; }
  }
  if (wave >= 4 + NLW) {
    // ------------------------------------------------------------------ storer waves: drain the staged stripes, a piece per stage
    const int sw = wave - 4 - NLW;
    constexpr int PP = 16 / KCH;                              // store instructions per storer wave and stage (16 per stripe)
    const int rsub = lane >> 4, c = lane & 15;
    auto drain = [&](int stripe_it, int t0, int t1) {        // pieces t0 .. t1-1 of the stripe of iteration stripe_it
      const unsigned char* ob = outs + (stripe_it & 1) * kOut;
      const int m0 = (first + stripe_it * step) * 128;
      u32x4 v[4];
      for (int t = t0; t < t1; t += 4) {
        if constexpr (POOL > 1) {
          // MaxPooling over groups of POOL consecutive staged rows (EPI only: values in [0, 6], so bf16 bit patterns order like signed 16-bit
          // integers and -0 is the smallest).  Four pieces = 16 rows per iteration (PP is 8 or 4 here); lane group rsub takes rows 4 rsub .. + 3
          // of them, one row per register: the maxima are taken between registers and every store instruction still writes whole rows.
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const int rb = (sw * 16 + t) * 4 + 4 * rsub;
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int r = rb + u; v[u] = *reinterpret_cast<const u32x4*>(ob + r * 256 + ((c ^ (r & 15)) * 16)); }
          const s16x8 a = __builtin_elementwise_max(__builtin_bit_cast(s16x8, v[0]), __builtin_bit_cast(s16x8, v[1]));
          const s16x8 b = __builtin_elementwise_max(__builtin_bit_cast(s16x8, v[2]), __builtin_bit_cast(s16x8, v[3]));
          if constexpr (POOL == 4) {
            if (m0 + rb < p.M && !WRES_EXP(p, 4))
              *reinterpret_cast<u32x4*>(p.Y + (long)((m0 + rb) >> 2) * p.N + slice * 128 + c * 8) = __builtin_bit_cast(u32x4, __builtin_elementwise_max(a, b));
          } else {
            if (m0 + rb < p.M && !WRES_EXP(p, 4)) *reinterpret_cast<u32x4*>(p.Y + (long)((m0 + rb) >> 1) * p.N + slice * 128 + c * 8) = __builtin_bit_cast(u32x4, a);
            if (m0 + rb + 2 < p.M && !WRES_EXP(p, 4)) *reinterpret_cast<u32x4*>(p.Y + (long)(((m0 + rb) >> 1) + 1) * p.N + slice * 128 + c * 8) = __builtin_bit_cast(u32x4, b);
          }
          continue;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (t + u < t1) {
            const int r = (sw * 16 + t + u) * 4 + rsub;
            v[u] = *reinterpret_cast<const u32x4*>(ob + r * 256 + ((c ^ (r & 15)) * 16));
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (t + u < t1) {
            const int r = (sw * 16 + t + u) * 4 + rsub;
            if (m0 + r < p.M && !WRES_EXP(p, 4)) *reinterpret_cast<u32x4*>(p.Y + (long)(m0 + r) * p.N + slice * 128 + c * 8) = v[u];
          }
      }
    };
    for (int jb = 0; jb <= total; ++jb) {
      __builtin_amdgcn_s_barrier();
      if (jb >= KCH) {
        const int done = jb / KCH - 1, t = jb % KCH;         // the stripe that completed last; its piece for this stage
        if (jb == total) drain(done, 0, 16);                  // the last stripe: everything at once
        else drain(done, t * PP, (t + 1) * PP);
      }
    }
    return;
  }
  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves: a share of every stage each
    const int lw = wave - 4;
    const int rsub = lane >> 3, pos = lane & 7;
    const long ldk = p.K;
    auto issue = [&](int lin) {
      const int slot = lin % RR;
      lin = lin < total ? lin : total - 1;              // past the end: re-read the last stage into an already consumed slot
      const int it = lin / KCH, kc = lin % KCH;
      const int r0 = (first + it * step) * 128;
      unsigned char* dst = smem + slot * kStage;
#pragma unroll
      for (int jj = 0; jj < kIPS; ++jj) {
        const int jrow = lw * kIPS + jj;
        int row = r0 + 8 * jrow + rsub;
        row = row < p.M ? row : p.M - 1;
        const int c = pos ^ ((4 * jrow + (lane >> 4)) & 7);
        if (!WRES_EXP(p, 1)) glds16(p.X + row * ldk + kc * 64 + c * 8, dst + jrow * 1024);
      }
    };
#pragma unroll
    for (int s = 0; s < RR - 1; ++s) issue(s);
    for (int i = 0; i < total; ++i) {
#ifdef CRNN_WRES_EXP
      const bool tr = p.trace && wg == 8 && lw == 0 && lane == 0 && i < 64;
      if (tr) p.trace[i * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RR - 3) * kIPS) : "memory");   // stages i and i+1 have landed
#ifdef CRNN_WRES_EXP
      if (tr) p.trace[i * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
      if (!WRES_EXP(p, 32)) __builtin_amdgcn_s_barrier();
#ifdef CRNN_WRES_EXP
      if (tr) p.trace[i * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
      issue(i + RR - 1);                                                        // into the slot stage i-1 has just released
#ifdef CRNN_WRES_EXP
      if (tr) p.trace[i * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- compute waves: 32 channels x K of W in registers
  if (WRES_EXP(p, 32)) return;
  wres_compute<KCH, RR, 1, 4, EPI>(smem, outs, p.W, p.K, slice * 128 + wave * 32, wave, lane, mine, epi);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Forward pointwise convolution fed by the PRE-BatchNorm depthwise output (utils.py:44-49, training):
//     q[M][N] = ReLU6(BN1(d))[M][K] . W[N][K]^T,   plus the column sums / sums of squares of q as stored (BatchNorm-2 statistics).
// Same MFMA role as above; the producers cannot be LDS-DMA (the operand is transformed on the way), so the four other waves
// are IO waves: each loads a quarter of every 16-KiB stage into registers three stages ahead (48 KiB per CU in flight), applies
// ReLU6(x * scale[ch] + shift[ch]) -> bf16 (the arithmetic of bn_act_pool_drop_kernel / gemm_bf16.inc's prologue, bit for bit)
// and writes the chunk to its swizzled ring position; between stages it drains a piece of the previous stripe's staged bf16 tile
// to global memory and accumulates that piece's column sums and sums of squares in registers over the whole launch.  One
// partial-statistics row per IO wave and stripe lane, written once at the end: [Q * 8 * 4][2][N] instead of [M / 128][2][N].
struct WresFwdParams {
  const bf16_t* X; const bf16_t* W; bf16_t* Y;
  const float* scale; const float* shift;   // [K] each
  float* stats;                             // [rows][2][N] or null
  int M, N, K;
  int stripes, S, Q, nxcd;
};
constexpr int kRf = 3;                      // ring slots (the data in flight lives in the IO waves' registers)

__device__ __forceinline__ unsigned wres_bnrelu6_pair(unsigned w, f32x2_t s, f32x2_t t) {
  f32x2_t v = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
  v = __builtin_elementwise_fma(v, s, t);
  v[0] = __builtin_amdgcn_fmed3f(v[0], 0.f, 6.f); v[1] = __builtin_amdgcn_fmed3f(v[1], 0.f, 6.f);
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <int KCH, int CB, int PB>   // K / 64; 32-channel blocks per MFMA wave (workgroup = 128 CB channels); 32-pixel blocks per stage / stripe
__global__ __launch_bounds__(512) void gemm_wres_fwd_kernel(WresFwdParams p) {
  constexpr int stage = PB * 32 * 128, orow = 256 * CB, otile = PB * 32 * orow, NS = 128 * CB, PX = 32 * PB;
  constexpr int CPL = PB;                                      // 16-byte stage chunks per IO lane and stage (PX rows x 8 chunks over 256 lanes)
  // stages in flight in the IO waves' registers: 64 KiB per CU for the deep-K shapes; the K <= 128 shapes are HBM-bound with a write
  // stream twice the read stream and run faster with 32 KiB (measured: 75 / 160 us against 85 / 178 us for blocks 2 / 3)
#ifndef CRNN_WRESF_DEEP
#define CRNN_WRESF_DEEP 1    // 1: stage depth trimmed for the deep-K shapes so that no register spills (round 4); 0: the round-2 depths
#endif
  // (deep-K: 64 KiB in flight asked for pf[4][4] / pf[8][2] and left the K = 512 and K = 256 instantiations 5 and 8 registers short of their
  // 256-register ceiling -- scratch traffic in the IO waves' steady state; 48 KiB in flight fits)
  constexpr int D = CRNN_WRESF_DEEP ? (KCH >= 8 ? 12 / PB : (KCH >= 4 ? (PB == 2 ? 6 : 4) : 8 / PB)) : (KCH >= 4 ? 16 : 8) / PB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kRf stages | two staging tiles | scale[K] | shift[K]
  unsigned char* const outs = smem + kRf * stage;
  float* const tab = reinterpret_cast<float*>(outs + 2 * otile);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int x = wg % p.nxcd, j = wg / p.nxcd;
  const int slice = j % p.S, q = j / p.S;
  const int step = p.Q * p.nxcd;
  const int first = q * p.nxcd + x;
  const int mine = first < p.stripes ? (p.stripes - first + step - 1) / step : 0;
  const int total = mine * KCH;
  const int srow = (q * p.nxcd + x) * 4 + (wave - 4);          // this IO wave's partial-statistics row
  if (mine <= 0) {                                             // no stripe: the statistics rows must still read as zero
    if (p.stats && wave >= 4 && lane < 16 * CB)
      for (int e = 0; e < 8; ++e) {
        p.stats[((long)srow * 2 + 0) * p.N + slice * NS + lane * 8 + e] = 0.f;
        p.stats[((long)srow * 2 + 1) * p.N + slice * NS + lane * 8 + e] = 0.f;
      }
    return;
  }
  for (int i = tid; i < p.K; i += 512) { tab[i] = p.scale[i]; tab[p.K + i] = p.shift[i]; }
  for (int i = tid; i < 2 * otile / 16; i += 512) reinterpret_cast<u32x4*>(outs)[i] = u32x4{0u, 0u, 0u, 0u};   // see stage_fast: dummy drains read zeros
  __syncthreads();

  if (wave < 4) {
    wres_compute<KCH, kRf, CB, PB>(smem, outs, p.W, p.K, slice * NS + wave * 32 * CB, wave, lane, mine);
    return;
  }
  // ------------------------------------------------------------------------ IO waves
  const int w = wave - 4;
  const int r8 = lane >> 3, c = lane & 7;                      // stage chunk: pixel row 8 (PB w + u) + r8, 16-byte piece c (8 channels)
  const long ldk = p.K;
  u32x4 pf[D][CPL];                                              // raw chunks of the stages in flight (stage lin in buffer lin % D)
  auto load = [&](int lin, u32x4 (&buf)[CPL]) {
    lin = lin < total ? lin : total - 1;                       // past the end: a valid address, the data is not used
    const int it = lin / KCH, kc = lin % KCH;
    const bf16_t* src = p.X + ((long)(first + it * step) * PX + 8 * PB * w + r8) * ldk + kc * 64 + c * 8;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      if (CRNN_WRESF_NT) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + (long)(8 * u) * ldk));
      else buf[u] = *reinterpret_cast<const u32x4*>(src + (long)(8 * u) * ldk);
    }
  };
  // (stages past the end are written too -- transformed junk from the clamped loads into a slot whose stage has been consumed:
  // keeping the steady-state loop free of branches is what lets the wait-count insertion count the outstanding loads exactly)
  auto write = [&](int lin, const u32x4 (&buf)[CPL]) {
    const int kc = lin % KCH;
    const float4 s0 = *reinterpret_cast<const float4*>(tab + kc * 64 + c * 8), s1 = *reinterpret_cast<const float4*>(tab + kc * 64 + c * 8 + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(tab + p.K + kc * 64 + c * 8), t1 = *reinterpret_cast<const float4*>(tab + p.K + kc * 64 + c * 8 + 4);
    unsigned char* dst = smem + (lin % kRf) * stage;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const int row = 8 * (PB * w + u) + r8;
      u32x4 o;
      if (WRES_EXP(p, 64)) o = buf[u]; else {
      o.x = wres_bnrelu6_pair(buf[u].x, f32x2_t{s0.x, s0.y}, f32x2_t{t0.x, t0.y});
      o.y = wres_bnrelu6_pair(buf[u].y, f32x2_t{s0.z, s0.w}, f32x2_t{t0.z, t0.w});
      o.z = wres_bnrelu6_pair(buf[u].z, f32x2_t{s1.x, s1.y}, f32x2_t{t1.x, t1.y});
      o.w = wres_bnrelu6_pair(buf[u].w, f32x2_t{s1.z, s1.w}, f32x2_t{t1.z, t1.w});
      }
      *reinterpret_cast<u32x4*>(dst + row * 128 + ((c ^ ((row >> 1) & 7)) * 16)) = o;
    }
  };
  // drain + statistics: a staging-tile row is 16 CB pieces of 16 bytes; lane = (row rg of a group of 4 / CB, piece cN = 8 channels);
  // 8 pieces per IO wave and stripe (PX rows x 16 CB pieces = 2048 per stripe either way)
  constexpr int PP = 8 / KCH;                                  // pieces per stage
  constexpr int RG = 4 / CB;                                   // rows a wave-instruction covers
  const int rg = lane / (16 * CB), cN = lane % (16 * CB);
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
  // stripe_it = -1 (nothing finished yet): reads staging tile 1, still all zeros (the MFMA waves first write it at the end of
  // stripe 1), adds zeros to the statistics and stores zeros over stripe 0's rows, which this wave rewrites with the result later
  auto drain = [&](int stripe_it, int t0, int t1) {
    const unsigned char* ob = outs + (stripe_it & 1) * otile;
    const long m0 = (long)(first + (stripe_it < 0 ? 0 : stripe_it) * step) * PX;
    for (int t = t0; t < t1; ++t) {
      const int r = (w * 8 + t) * RG + rg;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ob + r * orow + ((cN ^ (r & 15)) * 16));
      if (!WRES_EXP(p, 4)) *reinterpret_cast<u32x4*>(p.Y + (m0 + r) * p.N + slice * NS + cN * 8) = v;
      const unsigned ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (WRES_EXP(p, 128)) { ssum[2 * e] = __uint_as_float(ww[e]); continue; }
        const float lo = __uint_as_float(ww[e] << 16), hi = __uint_as_float(ww[e] & 0xffff0000u);
        ssum[2 * e] += lo; ssq[2 * e] = fmaf(lo, lo, ssq[2 * e]);
        ssum[2 * e + 1] += hi; ssq[2 * e + 1] = fmaf(hi, hi, ssq[2 * e + 1]);
      }
    }
  };
  auto stage_step = [&](int jb, u32x4 (&buf)[CPL]) {          // everything an IO wave does around barrier jb; buf = buffer (jb + 2) % D
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's ring writes are in LDS before the others pass the barrier
    __builtin_amdgcn_s_barrier();
    if (jb < total) { write(jb + 2, buf); load(jb + 2 + D, buf); }
    if (jb >= KCH) {
      const int done = jb / KCH - 1, t = jb % KCH;
      if (jb == total) drain(done, 0, 8);                       // the last stripe: everything at once
      else drain(done, t * PP, (t + 1) * PP);
    }
  };
  auto stage_fast = [&](int jb, u32x4 (&buf)[CPL]) {           // the same for jb < total, straight-line: no branch between the VMEM operations
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    write(jb + 2, buf); load(jb + 2 + D, buf);
    drain(jb / KCH - 1, (jb % KCH) * PP, (jb % KCH + 1) * PP);
  };
#pragma unroll
  for (int k = 0; k < D; ++k) load(k, pf[k]);
  write(0, pf[0]); load(D, pf[0]);
  write(1, pf[1]); load(D + 1, pf[1]);
  int jb = 0;
  for (; jb + D <= total; jb += D) {
#pragma unroll
    for (int k = 0; k < D; ++k) stage_fast(jb + k, pf[(k + 2) % D]);
  }
#pragma unroll
  for (int k = 0; k < D; ++k)
    if (jb + k <= total) stage_step(jb + k, pf[(k + 2) % D]);
  if (p.stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (CB == 1) { ssum[e] += __shfl_xor(ssum[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); }
      ssum[e] += __shfl_xor(ssum[e], 32, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
    }
    if (lane < 16 * CB) {
      // the row addresses are formed HERE: left to itself the compiler hoists them above the stage loops and, at the 256-register
      // ceiling of the deep-K instantiations, spills them across (an opaque copy of the lane index pins the address arithmetic)
      int cq = cN;
      asm volatile("" : "+v"(cq));
      float* r0 = p.stats + ((long)srow * 2 + 0) * p.N + slice * NS + cq * 8;
      float* r1 = p.stats + ((long)srow * 2 + 1) * p.N + slice * NS + cq * 8;
      *reinterpret_cast<float4*>(r0) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
      *reinterpret_cast<float4*>(r0 + 4) = make_float4(ssum[4], ssum[5], ssum[6], ssum[7]);
      *reinterpret_cast<float4*>(r1) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
      *reinterpret_cast<float4*>(r1 + 4) = make_float4(ssq[4], ssq[5], ssq[6], ssq[7]);
    }
  }
}

template <int KCH, int NLW, bool EPI = false, bool BNS = false, int POOL = 1>
int launch_wres(const WresParams& p, int grid, hipStream_t stream) {
  const int lds = (EPI ? kR - 1 : kR) * kStage + 2 * kOut + (EPI ? 1024 : 0);
  CRNN_LDS_ATTR((gemm_wres_kernel<KCH, NLW, EPI, BNS, POOL>), lds);
  hipLaunchKernelGGL((gemm_wres_kernel<KCH, NLW, EPI, BNS, POOL>), dim3(grid), dim3(384 + 64 * NLW), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

}  // namespace

// 0 if crnn_gemm_wres_bf16 handles (N, K), else -3: N a multiple of 128 up to 1024, K in {64, 128, 256, 512}
extern "C" int crnn_gemm_wres_supported(int N, int K) {
  return (N >= 128 && N % 128 == 0 && (N <= 1024 || (K <= 128 && N <= 8192)) && (K == 64 || K == 128 || K == 256 || K == 512)) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}

// Y[M][N] (bf16) = X[M][K] (bf16, row stride K) . W[N][K]^T (bf16, row stride K), weights resident in registers.
// One persistent workgroup per CU (512 threads: 4 MFMA waves + 2 LDS-DMA loader waves + 2 storer waves, 160 KiB of LDS).
static int wres_geom(int M, int N, WresParams& p) {             // -> grid
  p.stripes = cdiv(M, 128); p.S = N / 128;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  p.nxcd = 8;
  int per_xcd = cus / 8;                                       // workgroups per XCD: one per CU
  if (p.S > per_xcd) {                                         // more slices than an XCD has CUs (round 5: dense1's data gradient, 36 slices of a 3.4 MB operand):
    p.nxcd = cus / p.S < 1 ? 1 : cus / p.S;                    // one workgroup per (slice, stripe lane), at most one per CU -- the stripe lanes no longer
    p.Q = 1;                                                   // coincide with XCDs, which only matters for operands that do not fit the L2s
    if (p.nxcd > p.stripes) p.nxcd = p.stripes;
    return p.nxcd * p.S;
  }
  p.Q = per_xcd / p.S;
  const int need = cdiv(p.stripes, p.nxcd);                    // stripe lanes that have any work
  if (p.Q > need) p.Q = need;
  return p.nxcd * p.Q * p.S;
}
static int gemm_wres_impl(const void* X, const void* W, void* Y, int M, int N, int K, const float* cscale, const float* cshift, hipStream_t stream,
                          const void* D = nullptr, const float* bnstate = nullptr, float* stats = nullptr, int pool = 1) {
  if (M <= 0 || N <= 0 || K <= 0) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_gemm_wres_supported(N, K));
  if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)M * (K > N ? K : N) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;     // 32-bit row offsets
  WresParams p;
  p.X = (const bf16_t*)X; p.W = (const bf16_t*)W; p.Y = (bf16_t*)Y; p.M = M; p.N = N; p.K = K;
  p.D = (const bf16_t*)D; p.bnstate = bnstate; p.stats = stats;
  const int grid = wres_geom(M, N, p);
  if (D) {                                                     // storer waves take the BatchNorm-backward statistics (K = 256 | 512 here)
    if (K == 256) return launch_wres<4, 2, false, true>(p, grid, stream);
    return launch_wres<8, 2, false, true>(p, grid, stream);
  }
#ifdef CRNN_WRES_EXP
  p.exp = crnn_knob("CRNN_WRES_EXP", 0);
  { const char* e = getenv("CRNN_WRES_TRACE"); p.trace = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  p.cscale = cscale; p.cshift = cshift;
  if (pool != 1) {      // the two pooled blocks of the CRNN: 128 -> 256 channels with MaxPooling2D((2,2)), 256 -> 512 with (1,2); pooled by the storer waves
    if (!cscale || M % pool) return CRNN_ERR_ARG;
#ifdef CRNN_WRES_POOL_NOEPI      // timing experiment: the pooled kernels without the BatchNorm + ReLU6 arithmetic in the MFMA waves (wrong values)
    if (pool == 4 && K == 128) return launch_wres<2, 2, false, false, 4>(p, grid, stream);
    if (pool == 2 && K == 256) return launch_wres<4, 2, false, false, 2>(p, grid, stream);
#endif
    if (pool == 4 && K == 128) return launch_wres<2, 2, true, false, 4>(p, grid, stream);
    if (pool == 2 && K == 256) return launch_wres<4, 2, true, false, 2>(p, grid, stream);
    return CRNN_ERR_UNSUPPORTED;
  }
  if (cscale) {
    switch (K / 64) {
      case 1: return launch_wres<1, 2, true>(p, grid, stream);
      case 2: return launch_wres<2, 2, true>(p, grid, stream);
      case 4: return launch_wres<4, 2, true>(p, grid, stream);
      default: return launch_wres<8, 2, true>(p, grid, stream);
    }
  }
  switch (K / 64) {
    case 1: return launch_wres<1, 2>(p, grid, stream);
    case 2: return launch_wres<2, 2>(p, grid, stream);
    case 4: return launch_wres<4, 2>(p, grid, stream);
    default: return launch_wres<8, 2>(p, grid, stream);
  }
}
extern "C" int crnn_gemm_wres_bf16(const void* X, const void* W, void* Y, int M, int N, int K, hipStream_t stream) {
  return gemm_wres_impl(X, W, Y, M, N, K, nullptr, nullptr, stream);
}
// crnn_gemm_wres_bf16 for the data gradient da = dq . W^T of a depthwise-separable block (utils.py:45-49 backwards) that also takes the
// statistics pass of the BatchNorm in front of the pointwise convolution: stat_partials [crnn_gemm_wres_bnstats_rows(M, N, K)][2][N]
// = per-channel partial sums of gy and gy * xhat, gy = da where 0 < d * scale + shift < 6 (ReLU6 gate), xhat = (d - mean) / sqrt(var + eps),
// d [M][N] bf16 = the BatchNorm's input, bnstate = [mean | var | scale | shift] x N.  Feed them to crnn_bn_bwd_finalize.  Shapes: as
// crnn_gemm_wres_bf16 with whole 128-row stripes (M % 128 == 0) and K in {256, 512}; -3 otherwise (run crnn_bn_bwd_ex's statistics pass).
extern "C" int crnn_gemm_wres_bnstats_supported(long M, int N, int K) {
  return (M > 0 && M % 128 == 0 && M <= 0x7fffffffL && M * (long)(K > N ? K : N) < (1L << 31) && (K == 256 || K == 512) &&
          crnn_gemm_wres_supported(N, K) == CRNN_OK) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_gemm_wres_bnstats_rows(long M, int N, int K) {
  if (crnn_gemm_wres_bnstats_supported(M, N, K) != CRNN_OK) return 0;
  WresParams p;
  wres_geom((int)M, N, p);
  return 2 * p.Q * p.nxcd;
}
extern "C" int crnn_gemm_wres_bf16_bnstats(const void* X, const void* W, void* Y, long M, int N, int K, const void* d, const float* bnstate,
                                           float* stat_partials, hipStream_t stream) {
  if (!d || !bnstate || !stat_partials) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_gemm_wres_bnstats_supported(M, N, K));
  if ((((uintptr_t)d | (uintptr_t)stat_partials | (uintptr_t)bnstate) & 15)) return CRNN_ERR_UNSUPPORTED;
  return gemm_wres_impl(X, W, Y, (int)M, N, K, nullptr, nullptr, stream, d, bnstate, stat_partials);
}
// Inference forward of a pointwise convolution with the BatchNorm + ReLU6 that follows folded in (utils.py:49-51, learning_phase 0):
// Y[M][N] (bf16) = ReLU6((X . W^T) * scale[n] + shift[n]), out_bnstate = [mean|var|scale|shift] of that BatchNorm (crnn_bn_infer_state).
// The epilogue runs on the fp32 accumulators in the MFMA waves (one rounding to bf16): bit-identical to crnn_pwconv_fwd(out_bnstate).
extern "C" int crnn_pwconv_fwd_wres_folded(const void* a, const void* wT, void* y, long M, int N, int K, const float* out_bnstate, hipStream_t stream) {
  if (!out_bnstate) return CRNN_ERR_ARG;
  if (M > 0x7fffffffL || (((uintptr_t)out_bnstate) & 15)) return CRNN_ERR_UNSUPPORTED;
  return gemm_wres_impl(a, wT, y, (int)M, N, K, out_bnstate + 2L * N, out_bnstate + 3L * N, stream);
}
// The same with the MaxPooling2D that follows the block's ReLU6 (utils.py:52-54) in the epilogue: y [M / pool_rows][N] = max over every group of
// pool_rows consecutive rows of ReLU6((X . W^T) * scale + shift).  pool_rows = 2: MaxPooling2D((1,2)) on a map of even width in NHWC order;
// pool_rows = 4: MaxPooling2D((2,2)) when the rows of X are in 2x2-window-major order (crnn_dwconv3x3_fwd_stream_ex, out_order 1).  The un-pooled
// map is never written.  Equal to max-pooling crnn_pwconv_fwd_wres_folded's output (rounding to bf16 is monotonic).  Shapes: (K, pool_rows) =
// (128, 4) | (256, 2) -- the CRNN's two pooled blocks --, M % pool_rows == 0; -3 otherwise.
extern "C" int crnn_pwconv_fwd_wres_folded_pool_supported(long M, int N, int K, int pool_rows) {
  return (M > 0 && M <= 0x7fffffffL && M % pool_rows == 0 && ((K == 128 && pool_rows == 4) || (K == 256 && pool_rows == 2)) &&
          crnn_gemm_wres_supported(N, K) == CRNN_OK) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_pwconv_fwd_wres_folded_pool(const void* a, const void* wT, void* y, long M, int N, int K, const float* out_bnstate, int pool_rows,
                                                hipStream_t stream) {
  if (!out_bnstate) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_pwconv_fwd_wres_folded_pool_supported(M, N, K, pool_rows));
  if ((((uintptr_t)out_bnstate) & 15)) return CRNN_ERR_UNSUPPORTED;
  return gemm_wres_impl(a, wT, y, (int)M, N, K, out_bnstate + 2L * N, out_bnstate + 3L * N, stream, nullptr, nullptr, nullptr, pool_rows);
}

namespace {
// two shapes of the forward kernel: (CB 1, PB 4) = 128 pixels x 128 channels per workgroup; (CB 2, PB 2) = 64 pixels x 256 channels,
// where the doubled weight fragments still fit the registers (K <= 256): half as many channel slices transform / read every pixel.
// K = 64 stays on the narrow shape: its 8 drain pieces per stage beside the doubled accumulators left <1, 2, 2> at the 256-register
// ceiling with 47 spilled registers (round 4's recompile), and no block of the CRNN has K = 64 with N a multiple of 256 (block 2: N = 128)
bool wres_fwd_wide(int N, int K) { return N % 256 == 0 && K >= 128 && K <= 256 && crnn_knob("CRNN_WRES_WIDE", 1); }
void wres_fwd_geom(long M, int N, int K, WresFwdParams& p, int& grid) {
  const bool wide = wres_fwd_wide(N, K);
  p.stripes = cdiv(M, wide ? 64 : 128); p.S = N / (wide ? 256 : 128);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  p.nxcd = 8;
  int per_xcd = cus / 8;
  if (per_xcd < p.S) per_xcd = p.S;
  p.Q = per_xcd / p.S;
  const int need = cdiv(p.stripes, p.nxcd);
  if (p.Q > need) p.Q = need;
  grid = p.nxcd * p.Q * p.S;
}
template <int KCH, int CB, int PB>
int launch_wres_fwd(const WresFwdParams& p, int grid, hipStream_t stream) {
  constexpr int fixed = kRf * PB * 32 * 128 + 2 * PB * 32 * 256 * CB;
  const int lds = fixed + 2 * p.K * (int)sizeof(float);
  CRNN_LDS_ATTR((gemm_wres_fwd_kernel<KCH, CB, PB>), fixed + 2 * 512 * (int)sizeof(float));
  hipLaunchKernelGGL((gemm_wres_fwd_kernel<KCH, CB, PB>), dim3(grid), dim3(512), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
}  // namespace

// 0 if crnn_pwconv_bnrelu6_fwd_wres handles the shape (whole 128-pixel stripes, N a multiple of 128 up to 1024, K in {64,128,256,512}), else -3
extern "C" int crnn_pwconv_fwd_wres_supported(long M, int N, int K) {
  return (M > 0 && M % 128 == 0 && M * (long)(K > N ? K : N) < (1L << 31) && N <= 1024 && crnn_gemm_wres_supported(N, K) == CRNN_OK) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
// rows of the partial statistics [rows][2][N] the kernel writes (every row and column of that block is written)
extern "C" int crnn_pwconv_fwd_wres_rows(long M, int N, int K) {
  if (crnn_pwconv_fwd_wres_supported(M, N, K) != CRNN_OK) return 0;
  WresFwdParams p; int grid;
  wres_fwd_geom(M, N, K, p, grid);
  return p.Q * p.nxcd * 4;
}
// q[M][N] (bf16) = ReLU6(d * scale + shift)[M][K] . wT[N][K]^T with in_bnstate = [mean|var|scale|shift] of the BatchNorm on d
// (as crnn_pwconv_bnrelu6_fwd with w_transposed = 1 and bf16 q: the same bf16 result bit for bit); stat_partials (may be NULL):
// [crnn_pwconv_fwd_wres_rows(M, N, K)][2][N] column sums / sums of squares of q as stored.
extern "C" int crnn_pwconv_bnrelu6_fwd_wres(const void* d, const float* in_bnstate, const void* wT, void* q, long M, int N, int K,
                                            float* stat_partials, hipStream_t stream) {
  if (!in_bnstate) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_pwconv_fwd_wres_supported(M, N, K));
  if ((((uintptr_t)d | (uintptr_t)wT | (uintptr_t)q | (uintptr_t)stat_partials) & 15)) return CRNN_ERR_UNSUPPORTED;
  WresFwdParams p; int grid;
  p.X = (const bf16_t*)d; p.W = (const bf16_t*)wT; p.Y = (bf16_t*)q; p.scale = in_bnstate + 2L * K; p.shift = in_bnstate + 3L * K;
  p.stats = stat_partials; p.M = (int)M; p.N = N; p.K = K;
  wres_fwd_geom(M, N, K, p, grid);
  if (wres_fwd_wide(N, K)) {
    if (K == 128) return launch_wres_fwd<2, 2, 2>(p, grid, stream);
    return launch_wres_fwd<4, 2, 2>(p, grid, stream);
  }
  switch (K / 64) {
    case 1: return launch_wres_fwd<1, 1, 4>(p, grid, stream);
    case 2: return launch_wres_fwd<2, 1, 4>(p, grid, stream);
    case 4: return launch_wres_fwd<4, 1, 4>(p, grid, stream);
    default: return launch_wres_fwd<8, 1, 4>(p, grid, stream);
  }
}
