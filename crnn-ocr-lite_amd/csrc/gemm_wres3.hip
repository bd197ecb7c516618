// Weights-resident PLANE GEMMs for the parity mode's pointwise (1x1) convolutions (utils.py:48-49, Conv2D(1x1) of every depthwise-separable block):
//     forward         q[M][N]  = ReLU6(BN1(d))[M][K] . W[K][N]            + column sums / sums of squares of q (BatchNorm-2 statistics)
//     data gradient   da[M][N] = dq[M][K] . W[N][K]^T                     + the statistics pass of BatchNorm-1's backward (sum gy, sum gy * xhat)
// fp32 tensors, fp32-accurate products from bf16 planes on v_mfma_f32_32x32x16_bf16 (three planes per operand and six products per k-step, or two
// planes and three products: common.h crnn_split3_pair, gemm_bf16.inc gemm_x3p_kernel).  The tile kernel these replace re-loads a 128 x K fp32 slab
// of W out of L2 for every 128-pixel tile, splits it into planes again and reads it from LDS once per MFMA; its staging waves spend as much on W as
// on the pixels.  Here the design of the bf16 mode's gemm_wres_fwd_kernel is carried over:
//   * a workgroup owns a SLICE of NS = 128 / KHN output channels for its whole life: each of its four MFMA waves keeps the bf16 planes of a
//     32-channel x (K / KHN) block of W as MFMA A-operand fragments in registers (NPL K / (4 KHN) VGPRs: 192 for three planes at K = 256), split ONCE
//     in the prologue.  K = 512 does not fit one wave (384 registers): the reduction is cut in two halves (KHN = 2; waves 0-1 take k < K / 2, waves
//     2-3 the rest, of two 32-channel blocks) and the halves' accumulators meet in LDS at the end of a stripe;
//   * the pixels stream ONCE: four IO waves load the fp32 rows of a 32 PB-pixel stripe (64 k per stage; registers, D stages ahead), apply
//     ReLU6(x * scale[k] + shift[k]) (forward: the arithmetic of bn_act_pool_drop_kernel, bit for bit), split into planes and write them to a
//     three-slot LDS ring (rows of 64 k = 128 B per plane, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): conflict-free ds_read_b128
//     fragment reads); the slices of a stripe run at the same time on CUs of one XCD, so HBM is read once and the other slices hit that XCD's L2;
//   * the MFMA waves leave a finished stripe as fp32 in one of two LDS staging tiles; the IO waves drain it to global memory in 16-byte pieces spread
//     over the stages of the next stripe (fully coalesced rows) and take the statistics from the same registers: a lane owns four channels over the
//     whole launch, one partial row per IO wave and stripe lane;
//   * one raw s_barrier per stage; at barrier i the IO waves have stages i and i + 1 in LDS, so the MFMA waves read the next stage's first fragments
//     before they reach the barrier.
// Numerics: the planes, the six (three) products per 16-k step in the tile kernel's order (small terms first) and ascending k -- for KHN = 1 the same
// fp32 accumulation chain as gemm_x3p_kernel (bit-identical results); KHN = 2 adds the two half-reduction chains at the end (summation order only).
// The statistics are the same sums in another order (per-lane fp32 chains of at most kMaxLaneTerms terms, then crnn_bn_finalize / crnn_bn_bwd_finalize in double).
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef W3_EXP
#define W3_EXP 0   // timing experiments (scripts/wres3_variants.sh; wrong results): 1 no transform / split arithmetic, 2 no MFMAs, 4 no fragment reads, 8 no pixel loads, 16 no drain, 32 no plane writes
#endif

namespace {

struct W3Params {
  const float* X;                  // streamed operand [M][K] fp32
  const float* W; long wrs, wks;   // weights: element strides per output channel / per reduction index
  float* Y;                        // [M][N] fp32
  const float* scale; const float* shift;   // MODE 0: BatchNorm-1 scale / shift [K]
  float* stats;                    // [rows][2][N] partial statistics (may be null in MODE 0)
  const float* D; const float* bnstate;     // MODE 1: d [M][N]; [mean | var | scale | shift] x N
  int M, N, K;
  int stripes, S, Q, nxcd;
};

constexpr int kMaxLaneTerms = 4096;   // longest fp32 chain of a statistics lane (rows per lane and launch); longer launches take the tile kernel

__device__ __forceinline__ bf16x8_t w3_frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(p)); }

// ---------------------------------------------------------------------------------------------------------------- MFMA waves
// wave (0..3) = (cb = wave % CBN, kh = wave / CBN): channels slice * NS + 32 cb .. + 31, reduction half kh.  Stage j of a stripe holds, per plane,
// PX = 32 PB pixel rows x 8 chunks of 8 k; chunk c is k = 64 j + 8 c (KHN = 1) or k = (c >> 2) K / 2 + 32 j + 8 (c & 3) (KHN = 2).
// Accumulator chains: a v_mfma_f32_32x32x16_bf16 that accumulates onto the result of the one before it issues every ~75-95 cycles, not every 32 (measured
// with one chain per wave: profiles/r06_wres3_bench.txt, first form).  Consecutive MFMAs of the instruction stream therefore go to DIFFERENT accumulators:
// the products of a 16-k step are dealt round-robin over NC chains per 32-pixel block (product t -> chain (t - T0) % NC; NC = 3, or 2 with two pixel
// blocks and six products), every k-step adds to the same NC chains, and the chains are added when the stripe ends: (c0 + c1) + c2 -- the two chains of
// small terms first with three planes.  Same products, another order of the fp32 accumulation than gemm_x3p_kernel's single chain.
// WL: with three planes of a 32 x 256 block of W a wave would hold 192 registers of fragments and have room for one chain; the LOW plane (one product of
// six) then lives in LDS -- written once by the wave that reads it, lane-linear, one ds_read_b128 per k-step -- and hi / mid (128 registers) stay resident.
template <int KST, int KHN, int NPL, int PB, bool WL>
__device__ __forceinline__ void w3_compute(const unsigned char* ring, unsigned char* outs, unsigned char* exch, unsigned char* wlo, const W3Params& p,
                                           int slice, int wave, int lane, int mine) {
  constexpr int CBN = 4 / KHN, KS = 4 / KHN, NS = 32 * CBN, NCH = NS / 4, NC = (PB == 2 && NPL == 3) ? 2 : 3, NG = KST * KS;
  constexpr int PLANE = PB * 4096, STAGE = NPL * PLANE, OROW = NS * 4, OTILE = PB * 32 * OROW;
  constexpr int NR = WL ? NPL - 1 : NPL;                       // planes of W in registers
  static_assert(!WL || NPL == 3, "the plane in LDS is the third");
  const int cb = wave % CBN, kh = wave / CBN;
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
  unsigned char* const wl = wlo + wave * (NG * 1024) + lane * 16;   // WL: this wave's low-plane fragments, [k-step g][lane] x 16 B
  // ---- the slice's planes, once: lane (l31, half) holds k = 8 chunk + 0..7 of channel l31 for every (stage, k-step)
  bf16x8_t wf[NR][KST][KS];
  {
    const float* wrow = p.W + (long)(slice * NS + 32 * cb + l31) * p.wrs;
#pragma unroll
    for (int j = 0; j < KST; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = 4 * kh * (KHN - 1) + 2 * ks + half;
        const int k0 = KHN == 1 ? 64 * j + 8 * c : (c >> 2) * (32 * KST) + 32 * j + 8 * (c & 3);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = wrow[(long)(k0 + e) * p.wks];
        unsigned w[3][4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) crnn_split3_pair(v[2 * pr], v[2 * pr + 1], w[0][pr], w[1][pr], w[2][pr]);
#pragma unroll
        for (int pl = 0; pl < NR; ++pl) wf[pl][j][ks] = __builtin_bit_cast(bf16x8_t, u32x4{w[pl][0], w[pl][1], w[pl][2], w[pl][3]});
        if constexpr (WL) *reinterpret_cast<u32x4*>(wl + (j * KS + ks) * 1024) = u32x4{w[2][0], w[2][1], w[2][2], w[2][3]};
      }
  }
  constexpr int PXp[6] = {2, 0, 1, 1, 0, 0}, PWp[6] = {0, 2, 1, 0, 1, 0};   // x_lo w_hi, x_hi w_lo, mid mid, x_mid w_hi, x_hi w_mid, hi hi: small terms first
  constexpr int T0 = NPL == 3 ? 0 : 3;                                      // two planes: the last three products
  f32x16 acc[PB][NC];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int lrow = l31 * 128, cbase = 4 * kh * (KHN - 1) + half;
  int slot = 0;
  bf16x8_t fx[NPL][PB], wlf = {};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (WL: this wave's own fragments are in LDS)
  __builtin_amdgcn_s_barrier();                                // barrier 0: stages 0 and 1 are in LDS
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int b = 0; b < PB; ++b) fx[pl][b] = w3_frag(ring + pl * PLANE + b * 4096 + lrow + ((cbase ^ sw) * 16));
  if constexpr (WL) wlf = w3_frag(wl);
  for (int it = 0; it < mine; ++it) {
#pragma unroll
    for (int j = 0; j < KST; ++j) {
      const unsigned char* A = ring + slot * STAGE + lrow;
      slot = slot + 1 == 3 ? 0 : slot + 1;
      const unsigned char* An = ring + slot * STAGE + lrow;    // the next stage has landed (the IO waves run one stage ahead of the barrier)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int g = j * KS + ks;                             // (compile-time after unrolling)
        bf16x8_t nx[NPL][PB], nwl = {};
        const unsigned char* src = ks + 1 < KS ? A + (((cbase + 2 * (ks + 1)) ^ sw) * 16) : An + ((cbase ^ sw) * 16);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int b = 0; b < PB; ++b) { if (W3_EXP & 4) nx[pl][b] = fx[pl][b]; else nx[pl][b] = w3_frag(src + pl * PLANE + b * 4096); }
        if constexpr (WL) { if (W3_EXP & 4) nwl = wlf; else nwl = w3_frag(wl + ((g + 1) % NG) * 1024); }
        __builtin_amdgcn_sched_barrier(0);                     // reads of the next k-step first, then this k-step's MFMAs (as wres_compute)
#pragma unroll
        for (int t = T0; t < 6; ++t)
#pragma unroll
          for (int b = 0; b < PB; ++b) {
            const bf16x8_t wa = (WL && PWp[t] == 2) ? wlf : wf[PWp[t] < NR ? PWp[t] : 0][j][ks];
            if (W3_EXP & 2) { if (g == 0 && t - T0 < NC) acc[b][(t - T0) % NC] = zero16; continue; }
            const int ch = (t - T0) % NC;
            if (g == 0 && t - T0 < NC) acc[b][ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, fx[PXp[t]][b], zero16, 0, 0, 0);   // C = 0: no clearing pass
            else acc[b][ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, fx[PXp[t]][b], acc[b][ch], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int b = 0; b < PB; ++b) fx[pl][b] = nx[pl][b];
        if constexpr (WL) wlf = nwl;
      }
      if (j + 1 < KST) __builtin_amdgcn_s_barrier();           // releases this stage's slot
    }
    // ---- end of the stripe: the chains' sum.  res[b][4 g + e] = channel 8 g + 4 half + e of the wave's 32, pixel 32 b + l31.  Staging tile:
    // [PX rows][NS channels] fp32, 16-byte piece pc of row px at position pc ^ (px & (NCH - 1)).
    f32x16 res[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if constexpr (NC == 3) res[b] = (acc[b][0] + acc[b][1]) + acc[b][2];
      else res[b] = acc[b][0] + acc[b][1];
    }
    unsigned char* ob = outs + (it & 1) * OTILE;
    if constexpr (KHN == 2) {
      unsigned char* ex = exch + (((it & 1) * CBN + cb) * PB) * 4096 + lane * 16;
      if (kh == 1) {                                           // upper half of the reduction: hand the partial sums over, lane for lane
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(ex + b * 4096 + g * 1024) = make_float4(res[b][4 * g], res[b][4 * g + 1], res[b][4 * g + 2], res[b][4 * g + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // barrier E
      } else {
        __builtin_amdgcn_s_barrier();                          // barrier E: the partner's sums are in LDS
#pragma unroll
        for (int b = 0; b < PB; ++b) {
          const int px = 32 * b + l31;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 o = *reinterpret_cast<const float4*>(ex + b * 4096 + g * 1024);
            const int pc = 8 * cb + 2 * g + half;
            *reinterpret_cast<float4*>(ob + px * OROW + ((pc ^ (px & (NCH - 1))) * 16)) =
                make_float4(res[b][4 * g] + o.x, res[b][4 * g + 1] + o.y, res[b][4 * g + 2] + o.z, res[b][4 * g + 3] + o.w);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // in LDS before this wave reaches the next barrier (which hands the tile to the IO waves)
      }
    } else {
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const int px = 32 * b + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int pc = 8 * cb + 2 * g + half;
          *reinterpret_cast<float4*>(ob + px * OROW + ((pc ^ (px & (NCH - 1))) * 16)) =
              make_float4(res[b][4 * g], res[b][4 * g + 1], res[b][4 * g + 2], res[b][4 * g + 3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the raw barrier orders nothing by itself
      __builtin_amdgcn_s_barrier();
    }
  }
  if constexpr (KHN == 2) __builtin_amdgcn_s_barrier();        // barrier F: the last stripe's tile is complete
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
// MODE 0: forward (BatchNorm-1 + ReLU6 on the way in, column statistics of the result); MODE 1: data gradient (plain operand, BatchNorm-1 backward statistics)
template <int KST, int KHN, int NPL, int PB, int MODE, int D>
__global__ __launch_bounds__(512) void gemm_wres3_kernel(W3Params p) {
  constexpr bool WL = NPL == 3 && KST * (4 / KHN) >= 16;       // 192 registers of fragments: the low plane goes to LDS (w3_compute)
  constexpr int CBN = 4 / KHN, NS = 32 * CBN, NCH = NS / 4, PX = 32 * PB;
  constexpr int PLANE = PB * 4096, STAGE = NPL * PLANE, OROW = NS * 4, OTILE = PX * OROW;
  constexpr int DLY = KHN == 2 ? 1 : 0;                        // KHN = 2: a stripe's tile is complete one barrier after its last stage
  // D = stages in flight in the IO waves' registers (8 PB KiB each).  The kernel's register count is set by the MFMA waves (W planes: ~250), so the IO waves
  // have the same budget for free: up to 80 raw registers = 80 KiB per CU in flight.  (With 32 KiB the K >= 256 shapes ran at the latency of their
  // loads: one 8-KiB stage per 0.59 us whatever the MFMA work, profiles/r06_wres3_bench.txt.)  Per shape: the deepest that does not spill (w3_dispatch).
  constexpr int NP = PX * NCH / 256;                           // staging-tile pieces per IO lane and stripe
  constexpr int RSTEP = 256 / NCH;                             // rows between a lane's consecutive pieces
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // ring (3 stages) | two staging tiles | exchange (KHN = 2) | W low plane (WL) | scale[K] | shift[K]
  unsigned char* const outs = smem + 3 * STAGE;
  unsigned char* const exch = outs + 2 * OTILE;
  unsigned char* const wlo = exch + (KHN == 2 ? 2 * CBN * PB * 4096 : 0);
  float* const tab = reinterpret_cast<float*>(wlo + (WL ? 4 * KST * (4 / KHN) * 1024 : 0));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int x = wg % p.nxcd, jq = wg / p.nxcd;
  const int slice = jq % p.S, q = jq / p.S;
  const int step = p.Q * p.nxcd;
  const int first = q * p.nxcd + x;
  const int mine = first < p.stripes ? (p.stripes - first + step - 1) / step : 0;
  const int total = mine * KST;
  const int tio = tid - 256;
  const int srow = (q * p.nxcd + x) * 4 + (wave - 4);          // this IO wave's partial-statistics row
  const int pc = tio % NCH, prow = tio / NCH;                   // IO lane: staging-tile piece pc (4 channels) of rows prow + RSTEP t
  if (mine <= 0) {                                             // no stripe: the statistics rows must still read as zero
    if (p.stats && wave >= 4 && lane < NCH) {
      float* r0 = p.stats + ((long)srow * 2 + 0) * p.N + slice * NS + lane * 4;
      float* r1 = p.stats + ((long)srow * 2 + 1) * p.N + slice * NS + lane * 4;
      *reinterpret_cast<float4*>(r0) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(r1) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  if constexpr (MODE == 0) for (int i = tid; i < p.K; i += 512) { tab[i] = p.scale[i]; tab[p.K + i] = p.shift[i]; }
  for (int i = tid; i < 2 * OTILE / 16; i += 512) reinterpret_cast<u32x4*>(outs)[i] = u32x4{0u, 0u, 0u, 0u};   // dummy drains (below) read zeros
  __syncthreads();

  if (wave < 4) {
    w3_compute<KST, KHN, NPL, PB, WL>(smem, outs, exch, wlo, p, slice, wave, lane, mine);
    return;
  }
  // ------------------------------------------------------------------------ IO waves
  const int r32 = tio >> 3, c = tio & 7;                       // stage item: rows r32 + 32 u, chunk c (8 k)
  const int csw = (c ^ ((r32 >> 1) & 7)) * 16;
  const long ldk = p.K;
  auto kofs = [&](int j) { return KHN == 1 ? 64 * j + 8 * c : (c >> 2) * (32 * KST) + 32 * j + 8 * (c & 3); };
  float4 pf[D][PB][2];
  auto load = [&](int lin, float4 (&buf)[PB][2]) {
    lin = lin < total ? lin : total - 1;                       // past the end: a valid address, the data is not used
    const int it = lin / KST, j = lin % KST;
    const float* src = p.X + ((long)(first + it * step) * PX + r32) * ldk + kofs(j);
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      if (W3_EXP & 8) { buf[u][0] = make_float4(1.f, 2.f, 3.f, (float)lin); buf[u][1] = buf[u][0]; continue; }
      buf[u][0] = *reinterpret_cast<const float4*>(src + (long)(32 * u) * ldk);
      buf[u][1] = *reinterpret_cast<const float4*>(src + (long)(32 * u) * ldk + 4);
    }
  };
  // A stage's planes are FORMED (transform + split: ~75 vector operations per 8 values) one stage interval before they are WRITTEN: compute(s + 1) runs after
  // store(s) in the same interval, so the LDS write latency, the table reads and the drain's LDS / memory operations all overlap arithmetic instead of
  // queueing in front of the barrier (the first form did load -> table -> arithmetic -> write -> wait -> barrier in series: ~850 cycles per stage with
  // the MFMAs and the drain compiled out, against 384 cycles of MFMAs at K = 512; profiles/r06_wres3_ablate.txt).
  // (stages past the end are formed and written too -- junk from the clamped loads into a slot whose stage has been consumed: the steady state stays branch-free)
  u32x4 pw[PB][NPL];
  auto compute = [&](int lin, const float4 (&buf)[PB][2]) {
    const int j = lin % KST;
    float sc[8], sh[8];
    if constexpr (MODE == 0) {
      const float* ts = tab + kofs(j);
      const float4 s0 = *reinterpret_cast<const float4*>(ts), s1 = *reinterpret_cast<const float4*>(ts + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(ts + p.K), t1 = *reinterpret_cast<const float4*>(ts + p.K + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      float v[8] = {buf[u][0].x, buf[u][0].y, buf[u][0].z, buf[u][0].w, buf[u][1].x, buf[u][1].y, buf[u][1].z, buf[u][1].w};
      if (W3_EXP & 1) {
        const u32x4 a = __builtin_bit_cast(u32x4, buf[u][0]), b = __builtin_bit_cast(u32x4, buf[u][1]);
        pw[u][0] = a; pw[u][1] = b;
        if constexpr (NPL == 3) pw[u][2] = a ^ b;
        continue;
      }
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = relu6f(fmaf(v[e], sc[e], sh[e]));
      }
      unsigned w[3][4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) crnn_split3_pair(v[2 * pr], v[2 * pr + 1], w[0][pr], w[1][pr], w[2][pr]);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) pw[u][pl] = u32x4{w[pl][0], w[pl][1], w[pl][2], w[pl][3]};
    }
  };
  auto store = [&](int lin) {
    unsigned char* dst = smem + (lin % 3) * STAGE + r32 * 128 + csw;
    if (W3_EXP & 32) { if (pw[0][0].x == 0x12345678u) dst[0] = 1; return; }
#pragma unroll
    for (int u = 0; u < PB; ++u)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(dst + pl * PLANE + u * 4096) = pw[u][pl];
  };
  // ---- drain + statistics.  Per lane: 4 channels (piece pc) of rows prow + RSTEP t, t < NP, of every stripe
  float ss[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
  float bmu[4], binv[4], bsc[4], bsh[4];
  if constexpr (MODE == 1) {
    const int ch = slice * NS + 4 * pc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bmu[e] = p.bnstate[ch + e];
      binv[e] = 1.0f / sqrtf(p.bnstate[p.N + ch + e] + 1e-3f);   // BN_EPS, the spelling of bn_bwd_kernel
      bsc[e] = p.bnstate[2 * p.N + ch + e]; bsh[e] = p.bnstate[3 * p.N + ch + e];
    }
  }
  // CNT pieces from tb on of the stripe of iteration stripe_it; stripe_it < 0 (nothing finished yet): a staging tile that is still all zeros -- adds zeros
  // to the statistics and stores zeros over stripe 0's rows, which this lane rewrites with the result later (same lane, same address: in order)
  auto drain = [&](int stripe_it, int tb, auto cnt) {
    constexpr int CNT = decltype(cnt)::value;
    const unsigned char* ob = outs + (stripe_it & 1) * OTILE;
    const long m0 = (long)(first + (stripe_it < 0 ? 0 : stripe_it) * step) * PX;
    float4 dv[CNT], v[CNT];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < CNT; ++u) dv[u] = *reinterpret_cast<const float4*>(p.D + (m0 + prow + RSTEP * (tb + u)) * p.N + slice * NS + 4 * pc);
    }
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
      const int r = prow + RSTEP * (tb + u);
      v[u] = *reinterpret_cast<const float4*>(ob + r * OROW + ((pc ^ (r & (NCH - 1))) * 16));
    }
#pragma unroll
    for (int u = 0; u < CNT; ++u) *reinterpret_cast<float4*>(p.Y + (m0 + prow + RSTEP * (tb + u)) * p.N + slice * NS + 4 * pc) = v[u];
#pragma unroll
    for (int u = 0; u < CNT; ++u) {
      const float ve[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ss[e] += ve[e]; qq[e] = fmaf(ve[e], ve[e], qq[e]); }
      } else {
        const float de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tt = fmaf(de[e], bsc[e], bsh[e]);
          const float gy = (tt > 0.f && tt < 6.f) ? ve[e] : 0.f;
          ss[e] += gy; qq[e] = fmaf(gy, (de[e] - bmu[e]) * binv[e], qq[e]);
        }
      }
    }
  };
  constexpr int PPS = NP >= KST ? NP / KST : 1;                // pieces drained per stage ...
  constexpr int PEV = NP >= KST ? 1 : KST / NP;                // ... of every PEV-th stage
  static_assert(NP >= KST ? NP % KST == 0 : KST % NP == 0, "pieces per stripe against stages per stripe");
  // everything an IO wave does around barrier jb; buf = buffer (jb + 3) % D; pw = the planes of stage jb + 2.  Barrier jb >= KST + DLY: stripe
  // (jb - DLY) / KST - 1 is in its staging tile.
  auto stage_step = [&](int jb, float4 (&buf)[PB][2], bool steady) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's ring writes are in LDS before the others pass the barrier
    __builtin_amdgcn_s_barrier();
    if (steady || jb < total) store(jb + 2);
    const int jd = jb - DLY + KST;                             // >= 0
    const int done = jd / KST - 2, t = jd % KST;
    if (!(W3_EXP & 16)) {
      if (steady || jb < total + DLY) {
        if (PEV == 1 || t % PEV == 0) drain(done, (t / PEV) * PPS, std::integral_constant<int, PPS>{});
      } else drain(done, 0, std::integral_constant<int, NP>{});    // the last stripe: everything at once
    }
    if (steady || jb < total) { compute(jb + 3, buf); load(jb + 3 + D, buf); }
  };
  static_assert(D >= 3, "three stages are formed before the first barrier");
#pragma unroll
  for (int k = 0; k < D; ++k) load(k, pf[k]);
  compute(0, pf[0]); store(0); load(D, pf[0]);
  compute(1, pf[1]); store(1); load(D + 1, pf[1]);
  compute(2, pf[2]); load(D + 2, pf[2]);
  int jb = 0;
  for (; jb + D <= total; jb += D) {                           // straight-line groups of D stages: the loads in flight stay countable
#pragma unroll
    for (int k = 0; k < D; ++k) stage_step(jb + k, pf[(k + 3) % D], true);
  }
#pragma unroll
  for (int k = 0; k < D + DLY; ++k)
    if (jb + k <= total + DLY) stage_step(jb + k, pf[(k + 3) % D], false);
  if (p.stats) {
    // lanes with the same piece: tio and tio + NCH (+ 2 NCH ...) within the wave
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (NCH <= 16) { ss[e] += __shfl_xor(ss[e], 16, 64); qq[e] += __shfl_xor(qq[e], 16, 64); }
      ss[e] += __shfl_xor(ss[e], 32, 64); qq[e] += __shfl_xor(qq[e], 32, 64);
    }
    if (lane < NCH) {
      float* r0 = p.stats + ((long)srow * 2 + 0) * p.N + slice * NS + lane * 4;
      float* r1 = p.stats + ((long)srow * 2 + 1) * p.N + slice * NS + lane * 4;
      *reinterpret_cast<float4*>(r0) = make_float4(ss[0], ss[1], ss[2], ss[3]);
      *reinterpret_cast<float4*>(r1) = make_float4(qq[0], qq[1], qq[2], qq[3]);
    }
  }
}

#ifndef W3_D1
#define W3_PBS 2   // 32-pixel blocks per stripe of the K <= 128 shapes
#define W3_D1 3     // K = 64 (64-pixel stripes)
#define W3_D2 3     // K = 128 (64-pixel stripes)
#define W3_D4F 6    // K = 256 forward
#define W3_D4B 6    // K = 256 data gradient
#define W3_D8 6     // K = 512
#endif
// (K, N) -> the instantiation: KST = K / 64 stages per stripe; K = 512 in two reduction halves on 64-channel slices; 32-pixel stripes where the planes
// fill the register file (192 VGPRs of W), 64-pixel stripes otherwise
struct W3Shape { int kst, khn, pb; };
bool w3_shape(int N, int K, W3Shape& s) {
  if (K != 64 && K != 128 && K != 256 && K != 512) return false;
  s.kst = K / 64; s.khn = K == 512 ? 2 : 1; s.pb = K >= 256 ? 1 : W3_PBS;
  const int ns = 128 / s.khn;
  return N >= ns && N % ns == 0 && N <= 1024;
}
void w3_geom(long M, int N, const W3Shape& s, W3Params& p, int& grid) {
  p.stripes = (int)(M / (32 * s.pb)); p.S = N / (128 / s.khn);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  p.nxcd = 8;
  int per_xcd = cus / 8;
  if (per_xcd < p.S) per_xcd = p.S;
  p.Q = per_xcd / p.S;
  const int need = cdiv(p.stripes, p.nxcd);
  if (p.Q > need) p.Q = need;
  grid = p.nxcd * p.Q * p.S;
}
int w3_supported(long M, int N, int K) {
  W3Shape s;
  if (M <= 0 || !w3_shape(N, K, s) || M % (32 * s.pb) != 0 || M * (long)(K > N ? K : N) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  W3Params p; int grid;
  w3_geom(M, N, s, p, grid);
  const long per_wg = cdiv(p.stripes, p.Q * p.nxcd);          // stripes of the busiest workgroup
  const int nch = (128 / s.khn) / 4;
  if (per_wg * (32 * s.pb * nch / 256) > kMaxLaneTerms) return CRNN_ERR_UNSUPPORTED;
  return CRNN_OK;
}
template <int KST, int KHN, int NPL, int PB, int MODE, int D>
int w3_launch(const W3Params& p, int grid, hipStream_t stream) {
  constexpr int CBN = 4 / KHN, NS = 32 * CBN;
  constexpr bool WL = NPL == 3 && KST * (4 / KHN) >= 16;
  constexpr int fixed = 3 * NPL * PB * 4096 + 2 * 32 * PB * NS * 4 + (KHN == 2 ? 2 * CBN * PB * 4096 : 0) + (WL ? 4 * KST * (4 / KHN) * 1024 : 0);
  const int lds = fixed + (MODE == 0 ? 2 * p.K * (int)sizeof(float) : 0);
  CRNN_LDS_ATTR((gemm_wres3_kernel<KST, KHN, NPL, PB, MODE, D>), fixed + 2 * 512 * (int)sizeof(float));
  hipLaunchKernelGGL((gemm_wres3_kernel<KST, KHN, NPL, PB, MODE, D>), dim3(grid), dim3(512), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
template <int NPL, int MODE>
int w3_dispatch(const W3Params& p, const W3Shape& s, int grid, hipStream_t stream) {
  if constexpr (MODE == 0) {
    if (s.kst == 1) return w3_launch<1, 1, NPL, W3_PBS, MODE, W3_D1>(p, grid, stream);
    if (s.kst == 2) return w3_launch<2, 1, NPL, W3_PBS, MODE, W3_D2>(p, grid, stream);
  }
  // (the data gradient's reductions are the blocks' OUTPUT channels: 256 or 512 wherever its result has a multiple of 128 channels)
  if (s.kst == 4) return w3_launch<4, 1, NPL, 1, MODE, (MODE == 0 ? W3_D4F : W3_D4B)>(p, grid, stream);
  if (s.kst == 8) return w3_launch<8, 2, NPL, 1, MODE, W3_D8>(p, grid, stream);
  return CRNN_ERR_UNSUPPORTED;
}

}  // namespace

// 0 if the weights-resident plane kernels handle (M pixels, N output channels, K reduction), else -3: K in {64, 128, 256, 512}; N a multiple of 128 (of 64
// at K = 512) up to 1024; whole stripes (M % 64 == 0; % 32 at K >= 256)
extern "C" int crnn_gemm_wres3_supported(long M, int N, int K) { return w3_supported(M, N, K); }
// rows of the partial statistics [rows][2][N] the kernels write (every row and column of that block is written)
extern "C" int crnn_gemm_wres3_stat_rows(long M, int N, int K) {
  if (w3_supported(M, N, K) != CRNN_OK) return 0;
  W3Shape s; w3_shape(N, K, s);
  W3Params p; int grid;
  w3_geom(M, N, s, p, grid);
  return p.Q * p.nxcd * 4;
}
// q[M][N] = ReLU6(d * scale + shift)[M][K] . w[K][N]   (in_bnstate = [mean|var|scale|shift] x K of the BatchNorm on d), planes = 3 | 2 bf16 planes per
// operand; stat_partials (may be NULL): [crnn_gemm_wres3_stat_rows(M, N, K)][2][N] column sums / sums of squares of q.
extern "C" int crnn_pwconv_bnrelu6_fwd_wres3(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K, int planes,
                                             float* stat_partials, hipStream_t stream) {
  if (!d || !in_bnstate || !w || !q || (planes != 2 && planes != 3)) return CRNN_ERR_ARG;
  CRNN_TRY(w3_supported(M, N, K));
  if ((((uintptr_t)d | (uintptr_t)q | (uintptr_t)stat_partials | (uintptr_t)in_bnstate) & 15)) return CRNN_ERR_UNSUPPORTED;
  W3Shape s; w3_shape(N, K, s);
  W3Params p{};
  p.X = d; p.W = w; p.wrs = 1; p.wks = N; p.Y = q; p.scale = in_bnstate + 2L * K; p.shift = in_bnstate + 3L * K; p.stats = stat_partials;
  p.M = (int)M; p.N = N; p.K = K;
  int grid; w3_geom(M, N, s, p, grid);
  return planes == 3 ? w3_dispatch<3, 0>(p, s, grid, stream) : w3_dispatch<2, 0>(p, s, grid, stream);
}
// da[M][N] = dq[M][K] . w[N][K]^T (w = the convolution's kernel [N = input channels][K = output channels]) and the statistics pass of the BatchNorm in
// front of the convolution: stat_partials [crnn_gemm_wres3_stat_rows(M, N, K)][2][N] = partial sums of gy and gy * xhat, gy = da where
// 0 < d * scale + shift < 6, xhat = (d - mean) / sqrt(var + eps); d [M][N] = that BatchNorm's input, bnstate = [mean|var|scale|shift] x N.
extern "C" int crnn_gemm_wres3_bnstats(const float* dq, const float* w, float* da, long M, int N, int K, int planes, const float* d, const float* bnstate,
                                       float* stat_partials, hipStream_t stream) {
  if (!dq || !w || !da || !d || !bnstate || !stat_partials || (planes != 2 && planes != 3)) return CRNN_ERR_ARG;
  if (K < 256) return CRNN_ERR_UNSUPPORTED;
  CRNN_TRY(w3_supported(M, N, K));
  if ((((uintptr_t)dq | (uintptr_t)da | (uintptr_t)d | (uintptr_t)stat_partials) & 15)) return CRNN_ERR_UNSUPPORTED;
  W3Shape s; w3_shape(N, K, s);
  W3Params p{};
  p.X = dq; p.W = w; p.wrs = K; p.wks = 1; p.Y = da; p.D = d; p.bnstate = bnstate; p.stats = stat_partials;
  p.M = (int)M; p.N = N; p.K = K;
  int grid; w3_geom(M, N, s, p, grid);
  return planes == 3 ? w3_dispatch<3, 1>(p, s, grid, stream) : w3_dispatch<2, 1>(p, s, grid, stream);
}
