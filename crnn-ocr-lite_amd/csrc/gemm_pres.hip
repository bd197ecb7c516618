// Data gradient of a pointwise (1x1) convolution in the PARITY mode from PRE-SPLIT planes (utils.py:48-49, training backward):
//     da[M][N] = dq[M][K] . W[N][K]^T     + the statistics pass of BatchNorm-1's backward (sum gy, sum gy * xhat)   [+ the planes of a = ReLU6(BN1(d))]
// dq arrives as bf16 PLANES [planes][M][K] (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): the words of common.h crnn_split3_pair), written once
// by the kernel that produces dq (BatchNorm-2's backward) -- two planes are the bytes of the fp32 tensor they replace.  The kernels this replaces
// (gemm_bf16.inc gemm_x3p_kernel, gemm_wres3.hip) load fp32 dq, split it in their staging / IO waves once per 128- or 64-channel slice of the result (four to
// eight times per element at N = 512) and were bound by exactly that: the IO side alone 335 us at K = N = 512 against 163 us for the MFMA side
// (profiles/r06_wres3_ablate.txt).  Here nothing is split in the GEMM and nothing passes through registers on the way in:
//   * a workgroup is FOUR waves, one per SIMD, each with the whole register file of its SIMD (up to 512 registers): wave w keeps the planes of the 32 x K block
//     W[slice * 128 + 32 w ..][0 .. K) as MFMA A-operand fragments for its whole life (planes * K / 4 registers: 256 for two planes at K = 512), split once
//     in the prologue.  A slice is 128 result channels for every K <= 512: half the slices (and L2 -> LDS traffic) of gemm_wres3's K = 512 form;
//   * the pixel planes stream through an LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs, no VALU): every wave issues its
//     quarter of a stage (64 k x 32 PB pixels x planes) right after the stage barrier, R - 1 stages ahead; rows of 64 k = 128 B, 16-byte chunk c of row r at
//     position c ^ ((r >> 1) & 7) (on the per-lane SOURCE address: the LDS image stays lane-linear): conflict-free ds_read_b128 fragment reads;
//   * the rows of d the epilogue needs (BatchNorm-1's backward statistics) come by LDS-DMA too, one stripe ahead, into the wave's own buffer: ALL loads of the
//     steady state are DMA, issued in a fixed order, so one counted s_waitcnt vmcnt(N) per stage is exact (stores are not counted: a later store may retire
//     before an earlier load; N = younger LOADS only is then still sufficient);
//   * a finished stripe leaves the accumulators through the wave's own LDS tile (transposition to pixel rows: no cross-wave traffic, no barrier) and is
//     drained in 8-row pieces between the MFMAs of the next stripe: fully coalesced 128-byte row segments of da (+ the planes of a), statistics in registers
//     (a lane owns four channels for the whole launch);
//   * one raw s_barrier per stage (four waves); the DMA is inline asm, so the compiler neither drains it (it would wait vmcnt(0) before every LDS read
//     that follows a DMA it knows of) nor counts it (its own waits only become more conservative).
// Numerics: the planes of crnn_split3_pair, the six (three) products per 16-k step in the tile kernel's order (small terms first) and ascending k on ONE
// accumulator per 32-pixel block -- the fp32 accumulation chain of gemm_x3p_kernel, so da is bit-identical to crnn_gemm_f32x2_bnstats / _f32x3_bnstats (a
// v_mfma_f32_32x32x16_bf16 that accumulates onto the one before it issues every 32 cycles like an independent one: scripts/probes/mfma_probe.hip).  The
// statistics are the same sums in another order (per-lane fp32 chains, crnn_bn_bwd_finalize_folded in double).
#include "common.h"
#include <type_traits>
#include <utility>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef PRES_EXP
#define PRES_EXP 0   // timing experiments (wrong results): 1 no drain, 2 no MFMAs, 4 no fragment reads, 8 no ring DMA, 16 no stage barriers, 32 no end-of-stripe tile writes, 64 no stage waits, 128 trace, 256 no da stores, 512 no statistics
#endif

namespace {

struct PresParams {
  const bf16_t* XP; long xps;      // planes of the streamed operand [planes][M][K], element stride between planes
  const float* W; long ldw;        // weights fp32 [N][ldw]: a result channel's K reduction values contiguous
  float* Y;                        // [M][N] fp32
  const float* D; const float* bnstate;   // d [M][N]; [mean | var | scale | shift] x N
  float* stats;                    // [rows][2][N]
  bf16_t* AP; long aps;            // NPA > 0: planes of a = ReLU6(d * scale + shift) [NPA][M][N]
  int M, N, K;
  int stripes, S, Q, nxcd;
};

constexpr int kMaxLaneTermsP = 4096;   // longest fp32 chain of a statistics lane

template <int... Is, class F> __device__ __forceinline__ void pr_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
// a value the compiler can neither re-derive from its inputs inside the loop nor keep anywhere but the accumulator half of the register file
__device__ __forceinline__ bf16x8_t pr_pin_a(bf16x8_t v) { asm volatile("" : "+a"(v)); return v; }
__device__ __forceinline__ bf16x8_t pr_pin_v(bf16x8_t v) { asm volatile("" : "+v"(v)); return v; }   // ... or keep where it is (opaque only)
template <int N, class F> __device__ __forceinline__ void pr_for(F&& f) { pr_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ unsigned pr_lds_addr(const void* p) { return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p); }
// LDS-DMA the compiler does not see: 16 bytes per lane from each lane's own global address to lds_dst (wave-uniform) + 16 lane
__device__ __forceinline__ void pr_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void pr_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x8_t pr_frag(const unsigned char* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(p)); }

// stages whose issue lies between the issue of stage lin + 1 and the wait for it (iterations lin + 3 - R .. lin - 1) that open a stripe (and so carry that
// stripe's d pieces): d = 1 .. R - 3 with (j - d) % KST == 0
constexpr int pr_opens(int j, int R, int KST) {
  int c = 0;
  for (int d = 1; d <= R - 3; ++d) if ((((j - d) % KST) + KST) % KST == 0) ++c;
  return c;
}

// KST = K / 64 stages per stripe, NPL planes per operand, PB 32-pixel blocks per stripe, R ring stages, OCC workgroups per CU the registers are budgeted for,
// NPA planes of a written by the drain (0: none), DB buffers for the rows of d (2: a stripe's rows are requested when the stripe begins; 1: after the
// previous stripe's drain has read the buffer -- frees LDS for ring stages where a stripe is long enough to hide the latency behind it)
template <int KST, int NPL, int PB, int R, int OCC, int NPA, int DB>
__global__ __launch_bounds__(256, OCC) void pres_dgrad_kernel(PresParams p) {
  constexpr int PPW = NPL * PB, STAGE = PPW * 4096, PX = 32 * PB, T0 = NPL == 3 ? 0 : 3;
  constexpr int NM = (6 - T0) * PB;                             // MFMAs per 16-k step = gaps for other instructions
  constexpr int NFA = OCC == 1 ? (256 - 16 * PB) / 4 : 0;       // fragments of W pinned to accumulator registers (one wave per SIMD: 256 of them)
  constexpr int WARM = (R + KST - 1) / KST;                    // stripes after which the steady-state load counts hold
  constexpr int NDR = 2 * PB;                                   // drain steps per stripe (16 rows x 128 B each)
  constexpr int NQ = NPA == 0 ? 22 : NPA == 2 ? 34 : 40;        // operations of a drain step (below)
  constexpr int GAPS = (4 * KST - 1) * NM, OPG = (NQ * NDR + GAPS - 1) / GAPS;   // gaps of a stripe after its first k-step; drain operations per gap
  constexpr int JD = DB == 2 ? 0 : (NM + (NQ * NDR + OPG - 1) / OPG + 4 * NM - 1) / (4 * NM);   // the stage whose gaps carry the stripe's d pieces (one buffer: the first after the drain)
  constexpr int ND = (KST - JD) * PPW;                          // loads issued between a stripe's d pieces and the next stripe (its drain begins there)
  static_assert(JD < KST, "the drain fills the stripe");
  static_assert(R >= 4 && OPG >= 1, "ring depth / drain gaps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // ring [R][planes][PB][32][128 B] | transposition tiles [4 waves][PB][4 KiB] | d [4 waves][DB][PB][4 KiB]
  const unsigned long long re0 = (PRES_EXP & 128) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int x = wg % p.nxcd, jq = wg / p.nxcd;
  const int slice = jq % p.S, q = jq / p.S;
  const int step = p.Q * p.nxcd;
  const int first = q * p.nxcd + x;
  const int mine = first < p.stripes ? (p.stripes - first + step - 1) / step : 0;
  const int srow = q * p.nxcd + x;
  const int c8 = lane & 3, r16 = lane >> 2;                     // drain lane: channels 8 c8 .. + 7 of the wave's 32, rows r16 + 16 t
  const int chan = slice * 128 + 32 * wave + 8 * c8;
  if (mine <= 0) {                                             // no stripe: the statistics row must still read as zero
    if (lane < 4) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<float4*>(p.stats + ((long)srow * 2 + 0) * p.N + chan + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(p.stats + ((long)srow * 2 + 1) * p.N + chan + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  unsigned char* const Tt = smem + R * STAGE + wave * (PB * 4096);
  unsigned char* const Db = smem + R * STAGE + 4 * PB * 4096 + wave * (DB * PB * 4096);
  const unsigned ring_a = __builtin_amdgcn_readfirstlane(pr_lds_addr(smem));
  const unsigned db_a = __builtin_amdgcn_readfirstlane(pr_lds_addr(Db));
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;

  // ---- the slice's planes, once: lane (l31, half) holds k = 64 j + 16 ks + 8 half + 0..7 of channel slice * 128 + 32 wave + l31.  The wave's 32 rows of W
  // come through LDS (the ring is not in use yet) 256 (128) k at a time: one LDS-DMA piece = 1 KiB of one (two) rows, fully coalesced; 16-byte chunk c of row r at
  // position c ^ (r & 15), so that the fragment reads (32 rows, the same chunk) spread over the banks.  (Read straight from memory -- two 16-byte loads per
  // fragment and lane, every lane in another 2-KiB row -- the prologue took 15 us of a 230-us launch at K = 512: profiles/r06_pres_ablate.txt.)
  bf16x8_t wf[NPL][KST][4];
  {
    constexpr int LDSB = R * STAGE + 4 * PB * 4096 + 4 * DB * PB * 4096;   // the launch's LDS: a quarter of it per wave for the staging
    constexpr int KH = LDSB >= 4 * 32768 ? 256 : 128, NH = 64 * KST / KH;  // k per pass (32 rows x KH floats per wave), passes
    constexpr int ROWB = KH * 4, RPP = 1024 / ROWB, CPRW = ROWB / 16;      // bytes per staged row, rows per 1-KiB DMA piece, 16-byte chunks per row
    static_assert(LDSB >= 4 * 32 * ROWB, "staging area");
    unsigned char* const wst = smem + wave * (32 * ROWB);       // this wave's staging area
    const unsigned wst_a = __builtin_amdgcn_readfirstlane(pr_lds_addr(wst));
#pragma unroll
    for (int H = 0; H < NH; ++H) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the previous pass's fragment reads are done)
#pragma unroll
      for (int pc = 0; pc < 32 / RPP; ++pc) {                   // piece pc = rows RPP pc .. : lane -> row RPP pc + lane / CPRW, chunk position lane % CPRW
        const int row = RPP * pc + lane / CPRW;
        pr_dma16(p.W + (long)(slice * 128 + 32 * wave + row) * p.ldw + KH * H + 4 * ((lane % CPRW) ^ (row & 15)), wst_a + pc * 1024);
      }
      pr_wait_vm<0>();
#pragma unroll
      for (int jj = 0; jj < KH / 64; ++jj)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int j = (KH / 64) * H + jj;
          const int c = (64 * jj + 16 * ks + 8 * half) >> 2;   // chunk of 4 floats within the staged row
          const float4 v0 = *reinterpret_cast<const float4*>(wst + l31 * ROWB + ((c ^ (l31 & 15)) << 4));
          const float4 v1 = *reinterpret_cast<const float4*>(wst + l31 * ROWB + (((c + 1) ^ (l31 & 15)) << 4));
          unsigned w[3][4];
          crnn_split3_pair(v0.x, v0.y, w[0][0], w[1][0], w[2][0]); crnn_split3_pair(v0.z, v0.w, w[0][1], w[1][1], w[2][1]);
          crnn_split3_pair(v1.x, v1.y, w[0][2], w[1][2], w[2][2]); crnn_split3_pair(v1.z, v1.w, w[0][3], w[1][3], w[2][3]);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {   // the accumulator half of the file holds the result blocks and the first NFA fragments; the rest stay in VGPRs
            const bf16x8_t f = __builtin_bit_cast(bf16x8_t, u32x4{w[pl][0], w[pl][1], w[pl][2], w[pl][3]});
            wf[pl][j][ks] = ((pl * KST + j) * 4 + ks < NFA) ? pr_pin_a(f) : pr_pin_v(f);
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // every wave has its fragments: the ring may land on the staging areas
  }
  float bmu[8], binv[8], bsc[8], bsh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bmu[e] = p.bnstate[chan + e];
    binv[e] = 1.0f / sqrtf(p.bnstate[p.N + chan + e] + 1e-3f);   // BN_EPS, the spelling of bn_bwd_kernel
    bsc[e] = p.bnstate[2 * p.N + chan + e]; bsh[e] = p.bnstate[3 * p.N + chan + e];
  }
  float ss[8], qq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ss[e] = 0.f; qq[e] = 0.f; }

  // ---- the DMA streams (per-lane base pointers; everything added later is wave-uniform).  Ring: wave w loads rows 8 w .. 8 w + 7 of every 32-row block of
  // a stage; lane -> row 8 w + (lane >> 3), chunk position lane & 7.  d: the wave's own 32 rows x 32 channels per block, 8 rows per piece g, same chunk
  // swizzle (the drain reads it as it reads the transposition tile)
  const bf16_t* const xlane = p.XP + (long)(8 * wave + (lane >> 3)) * p.K + 8 * ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));
  const float* dlane[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) dlane[g] = p.D + (long)(8 * g + (lane >> 3)) * p.N + slice * 128 + 32 * wave + 4 * ((lane & 7) ^ ((4 * g + (lane >> 4)) & 7));
  float* const ylane = p.Y + (long)r16 * p.N + chan;
  bf16_t* const aplane = NPA > 0 ? p.AP + (long)r16 * p.N + chan : nullptr;
  // stage (its, js) into ring slot `slot`, piece t of the wave's PPW; past the end: the last stage again, into a slot whose stage has been consumed (the
  // steady state stays branch-free and the load counts static)
  auto issue_ring = [&](int its, int js, int slot, int t) __attribute__((always_inline)) {
    if (its >= mine) { its = mine - 1; js = KST - 1; }
    const long uo = ((long)(first + its * step) * PX + 32 * (t % PB)) * p.K + 64 * js + (long)(t / PB) * p.xps;   // wave-uniform
    if (!(PRES_EXP & 8)) pr_dma16(xlane + uo, ring_a + slot * STAGE + wave * 1024 + t * 4096);
  };
  auto issue_d = [&](int it, int i) __attribute__((always_inline)) {   // piece i of 4 PB: block i / 4, rows 8 (i % 4) + (lane >> 3)
    const long uo = ((long)(first + it * step) * PX + 32 * (i / 4)) * p.N;
    pr_dma16(dlane[i % 4] + uo, db_a + ((it & (DB - 1)) * PB + i / 4) * 4096 + (i % 4) * 1024);
  };
  // ---- a drain step = rows r16 + 16 t of block b of a finished stripe: da, statistics, planes of a -- as NQ operations of four INDEPENDENT instructions
  // each, issued in the shadow of one MFMA each.  (A wave alone on its SIMD issues in order: an instruction that waits for the one before it -- the
  // fma -> compare -> select -> add chain of one channel's statistics -- holds up the MFMA behind it; written channel by channel the statistics cost
  // 2240 cycles per 64-pixel stripe next to 6144 cycles of MFMAs, profiles/r06_pres_ablate.txt.  So the chains run side by side: every operation
  // advances several channels by one instruction.)  State between the operations of a step:
  float4 dv0, dv1, dd0, dd1;
  long dro = 0;
  float dtt[8], dgy[8], dxh[8];
  unsigned dw[3][4];
  auto drain_op = [&](int itd, int b, int t, auto OPC) __attribute__((always_inline)) {
    constexpr int op = decltype(OPC)::value;
    auto VE = [&](int e) __attribute__((always_inline)) { return e < 4 ? dv0[e & 3] : dv1[e & 3]; };
    auto DE = [&](int e) __attribute__((always_inline)) { return e < 4 ? dd0[e & 3] : dd1[e & 3]; };
    if constexpr (op == 0) {                                    // the rows out of the transposition tile ...
      const int row = 16 * t + r16, rsw = (row >> 1) & 7;
      const unsigned char* tp = Tt + b * 4096 + row * 128;
      dv0 = *reinterpret_cast<const float4*>(tp + (((2 * c8) ^ rsw) << 4)); dv1 = *reinterpret_cast<const float4*>(tp + (((2 * c8 + 1) ^ rsw) << 4));
    } else if constexpr (op == 1) {                             // ... and the d buffer
      const int row = 16 * t + r16, rsw = (row >> 1) & 7;
      const unsigned char* dp = Db + ((itd & (DB - 1)) * PB + b) * 4096 + row * 128;
      dd0 = *reinterpret_cast<const float4*>(dp + (((2 * c8) ^ rsw) << 4)); dd1 = *reinterpret_cast<const float4*>(dp + (((2 * c8 + 1) ^ rsw) << 4));
      dro = ((long)(first + itd * step) * PX + 32 * b + 16 * t) * p.N;   // wave-uniform
    } else if constexpr (op == 2) {
      if (!(PRES_EXP & 256)) *reinterpret_cast<float4*>(ylane + dro) = dv0;
    } else if constexpr (op == 3) {
      if (!(PRES_EXP & 256)) *reinterpret_cast<float4*>(ylane + dro + 4) = dv1;
    } else if constexpr (op < 22 && (PRES_EXP & 512)) {
    } else if constexpr (op < 6) {                              // tt = d * scale + shift (scalar VALU forms: packed fp32 contends with the matrix pipe)
#pragma unroll
      for (int e = 4 * (op - 4); e < 4 * (op - 4) + 4; ++e) dtt[e] = fma_unpacked(DE(e), bsc[e], bsh[e]);
    } else if constexpr (op < 10) {                             // gy = da where 0 < tt < 6, as two selects
#pragma unroll
      for (int e = 2 * (op - 6); e < 2 * (op - 6) + 2; ++e) dgy[e] = dtt[e] > 0.f ? VE(e) : 0.f;
    } else if constexpr (op < 14) {
#pragma unroll
      for (int e = 2 * (op - 10); e < 2 * (op - 10) + 2; ++e) dgy[e] = dtt[e] < 6.f ? dgy[e] : 0.f;
    } else if constexpr (op < 16) {
#pragma unroll
      for (int e = 4 * (op - 14); e < 4 * (op - 14) + 4; ++e) ss[e] = add_unpacked(ss[e], dgy[e]);
    } else if constexpr (op < 18) {                             // xhat = (d - mean) * inv
#pragma unroll
      for (int e = 4 * (op - 16); e < 4 * (op - 16) + 4; ++e) dxh[e] = sub_unpacked(DE(e), bmu[e]);
    } else if constexpr (op < 20) {
#pragma unroll
      for (int e = 4 * (op - 18); e < 4 * (op - 18) + 4; ++e) dxh[e] = mul_unpacked(dxh[e], binv[e]);
    } else if constexpr (op < 22) {
#pragma unroll
      for (int e = 4 * (op - 20); e < 4 * (op - 20) + 4; ++e) qq[e] = fma_unpacked(dgy[e], dxh[e], qq[e]);
    } else if constexpr (op < 26) {                             // a = ReLU6(tt), two channels per operation; then the planes (the words of crnn_split3_pair)
#pragma unroll
      for (int e = 2 * (op - 22); e < 2 * (op - 22) + 2; ++e) dtt[e] = relu6f(dtt[e]);
    } else if constexpr (op == 26) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) dw[0][pr] = pack2_bf16(dtt[2 * pr], dtt[2 * pr + 1]);
    } else if constexpr (op < 31) {
      constexpr int pr = op - 27;
      dtt[2 * pr] = sub_unpacked(dtt[2 * pr], __uint_as_float(dw[0][pr] << 16)); dtt[2 * pr + 1] = sub_unpacked(dtt[2 * pr + 1], __uint_as_float(dw[0][pr] & 0xffff0000u));
    } else if constexpr (op == 31) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) dw[1][pr] = pack2_bf16(dtt[2 * pr], dtt[2 * pr + 1]);
    } else if constexpr (NPA == 3 && op < 36) {
      constexpr int pr = op - 32;
      dtt[2 * pr] = sub_unpacked(dtt[2 * pr], __uint_as_float(dw[1][pr] << 16)); dtt[2 * pr + 1] = sub_unpacked(dtt[2 * pr + 1], __uint_as_float(dw[1][pr] & 0xffff0000u));
    } else if constexpr (NPA == 3 && op == 36) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) dw[2][pr] = pack2_bf16(dtt[2 * pr], dtt[2 * pr + 1]);
    } else {
      constexpr int pl = op - (NPA == 3 ? 37 : 32);
      *reinterpret_cast<u32x4*>(aplane + (long)pl * p.aps + dro) = u32x4{dw[pl][0], dw[pl][1], dw[pl][2], dw[pl][3]};
    }
  };

#pragma unroll
  for (int s = 0; s < R - 1; ++s)
#pragma unroll
    for (int t = 0; t < PPW; ++t) issue_ring(s / KST, s % KST, s, t);

  constexpr int PXp[6] = {2, 0, 1, 1, 0, 0}, PWp[6] = {0, 2, 1, 0, 1, 0};   // x_lo w_hi, x_hi w_lo, mid mid, x_mid w_hi, x_hi w_mid, hi hi: small terms first
  f32x16 acc[PB];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bf16x8_t fx[NPL][PB];
  const int lofs = l31 * 128;
  int cbase = 0;                                               // ring slot of the stripe's first stage
  const unsigned long long tr0 = (PRES_EXP & 128) ? __builtin_amdgcn_s_memtime() : 0ull, rr0 = (PRES_EXP & 128) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // one stripe; DRAINS: the stripe before it leaves meanwhile (all but the first), WARMC: the steady-state load counts hold -- compile-time, so that the
  // gaps between the MFMAs carry straight-line code
  auto stripe = [&](int it, auto DRAINS, auto WARMC) __attribute__((always_inline)) {
    constexpr bool drains = decltype(DRAINS)::value, warm = decltype(WARMC)::value;
    pr_for<KST>([&](auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      // the DMA operations of this stage, spread over its gaps: (j == JD: the stripe's 4 PB pieces of d, then) the wave's PPW pieces of stage lin + R - 1
      constexpr int NOPD = j == JD ? 4 * PB : 0, NOP = NOPD + PPW, OSTR = (4 * NM) / NOP;
      static_assert(OSTR >= 1, "more DMA pieces than gaps");
      // every piece of stage lin + 1 this wave issued has landed: the loads issued since are (R - 3) stages + the d pieces of the stripes opened meanwhile
      if (!(PRES_EXP & 64)) pr_wait_vm<(R - 3) * PPW + (warm ? 4 * PB * pr_opens(j - JD, R, KST) : 0)>();
      if (!(PRES_EXP & 16)) __builtin_amdgcn_s_barrier();      // stage lin + 1 is in LDS (all waves); stage lin - 1's slot is free
      const int cslot = cbase + j >= R ? cbase + j - R : cbase + j, nslot = cslot + 1 == R ? 0 : cslot + 1, islot = cslot == 0 ? R - 1 : cslot - 1;
      const unsigned char* A = smem + cslot * STAGE + lofs;
      const unsigned char* An = smem + nslot * STAGE + lofs;
      if constexpr (j == 0 && !drains) {
        {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int b = 0; b < PB; ++b) fx[pl][b] = pr_frag(A + (pl * PB + b) * 4096 + ((half ^ sw) << 4));
        }
      }
      pr_for<4>([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value, sl = 4 * j + ks;
        bf16x8_t nx[NPL][PB];
        const unsigned char* src = ks < 3 ? A + (((2 * (ks + 1) + half) ^ sw) << 4) : An + ((half ^ sw) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int b = 0; b < PB; ++b) { if (PRES_EXP & 4) nx[pl][b] = fx[pl][b]; else nx[pl][b] = pr_frag(src + (pl * PB + b) * 4096); }
        __builtin_amdgcn_sched_barrier(0);                     // reads of the next k-step first, then this k-step's MFMAs, each followed by what issues in its shadow
        pr_for<NM>([&](auto MI) __attribute__((always_inline)) {
          constexpr int m = decltype(MI)::value, t = T0 + m / PB, b = m % PB;
          if (!(PRES_EXP & 2)) {
            if (sl == 0 && m < PB) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PWp[t]][j][ks], fx[PXp[t]][b], zero16, 0, 0, 0);   // C = 0: no clearing pass
            else acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PWp[t]][j][ks], fx[PXp[t]][b], acc[b], 0, 0, 0);
          } else if (sl == 0 && m < PB) acc[b] = zero16;
          constexpr int gs = ks * NM + m;                      // gap of the stage
          if constexpr (gs % OSTR == 0 && gs / OSTR < NOP) {
            constexpr int o = gs / OSTR;
            if constexpr (o < NOPD) issue_d(it, o);
            else issue_ring(it + (j + R - 1) / KST, (j + R - 1) % KST, islot, o - NOPD);   // stage lin + R - 1 into the slot stage lin - 1 has just released
          }
          constexpr int dg = sl * NM + m - NM;                 // gap of the stripe, counted from its second k-step: the previous stripe's drain
          if constexpr (drains && dg >= 0 && dg * OPG < NQ * NDR) {
            if (!(PRES_EXP & 1)) {
              if constexpr (dg == 0) pr_wait_vm<ND>();         // its rows of d have landed (issued a whole stripe ago)
              pr_for<OPG>([&](auto OI) __attribute__((always_inline)) {
                constexpr int dq = dg * OPG + decltype(OI)::value;
                if constexpr (dq < NQ * NDR) drain_op(it - 1, (dq / NQ) / 2, (dq / NQ) % 2, std::integral_constant<int, dq % NQ>{});
              });
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
          for (int b = 0; b < PB; ++b) fx[pl][b] = nx[pl][b];
      });
    });
    cbase = (cbase + KST) % R;
    // ---- end of the stripe: the result block into the wave's transposition tile.  Register 4 g + e of a block = channel 8 g + 4 half + e of the wave's 32,
    // pixel l31: 16-byte piece 2 g + half of row l31 at position piece ^ ((l31 >> 1) & 7)
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (PRES_EXP & 32) { if (acc[b][0] + acc[b][5] + acc[b][10] == 1.2345f) Tt[0] = 1; continue; }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(Tt + b * 4096 + lofs + (((2 * g + half) ^ sw) << 4)) = make_float4(acc[b][4 * g], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]);
    }
  };
  stripe(0, std::false_type{}, std::false_type{});
  {
    int it = 1;
    for (; it < mine && it < WARM; ++it) stripe(it, std::true_type{}, std::false_type{});
    for (; it < mine; ++it) stripe(it, std::true_type{}, std::true_type{});
  }
  if ((PRES_EXP & 128) && tid == 0) {   // [workgroup]: shader cycles, 100 MHz ticks, stripes of the main loop, ticks of the prologue
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(p.Y) + 4 * wg;
    tr[0] = __builtin_amdgcn_s_memtime() - tr0; tr[1] = __builtin_amdgcn_s_memrealtime() - rr0; tr[2] = (unsigned long long)mine; tr[3] = rr0 - re0;
  }
  // ---- the last stripe
  pr_wait_vm<0>();
  if (!(PRES_EXP & 1)) {
    pr_for<NDR>([&](auto MI) __attribute__((always_inline)) {
      constexpr int m = decltype(MI)::value;
      pr_for<NQ>([&](auto OPC) __attribute__((always_inline)) { drain_op(mine - 1, m / 2, m % 2, OPC); });
    });
  }
  // lanes with the same channels: lane, lane + 4, + 8, ...
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { ss[e] += __shfl_xor(ss[e], o, 64); qq[e] += __shfl_xor(qq[e], o, 64); }
  }
  if (lane < 4) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(p.stats + ((long)srow * 2 + 0) * p.N + chan + 4 * h) = make_float4(ss[4 * h], ss[4 * h + 1], ss[4 * h + 2], ss[4 * h + 3]);
      *reinterpret_cast<float4*>(p.stats + ((long)srow * 2 + 1) * p.N + chan + 4 * h) = make_float4(qq[4 * h], qq[4 * h + 1], qq[4 * h + 2], qq[4 * h + 3]);
    }
  }
}

#ifndef PRES_R8
#define PRES_R8 6     // ring stages at K = 512 (16 KiB each with two planes and 64-pixel stripes)
#define PRES_PB8 2    // 32-pixel blocks per stripe at K = 512: consecutive MFMAs go to different result blocks
#define PRES_DB8 1    // buffers for the rows of d at K = 512
#define PRES_R4 4     // K = 256
#define PRES_PB4 2    // 32-pixel blocks per stripe at K = 256
#define PRES_OCC4 1   // workgroups per CU at K = 256
#endif

struct PresShape { int kst, pb, occ, r; };
bool pres_shape(int N, int K, int planes, PresShape& s) {
  if (K != 256 && K != 512) return false;
  if (planes == 3 && K == 512) return false;                   // 384 registers of planes + three chains: not built
  s.kst = K / 64;
  s.pb = K == 512 ? PRES_PB8 : PRES_PB4; s.occ = K == 512 ? 1 : PRES_OCC4; s.r = K == 512 ? PRES_R8 : PRES_R4;
  if (planes == 3) { s.pb = 1; s.occ = 1; s.r = 8; }
  return N >= 128 && N % 128 == 0 && N <= 1024;
}
void pres_geom(long M, int N, const PresShape& s, PresParams& p, int& grid) {
  p.stripes = (int)(M / (32 * s.pb)); p.S = N / 128;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  p.nxcd = 8;
  int per_xcd = cus / 8 * s.occ;
  if (per_xcd < p.S) per_xcd = p.S;
  p.Q = per_xcd / p.S;
  const int need = cdiv(p.stripes, p.nxcd);
  if (p.Q > need) p.Q = need;
  grid = p.nxcd * p.Q * p.S;
}
int pres_supported(long M, int N, int K, int planes) {
  PresShape s;
  if (M <= 0 || (planes != 2 && planes != 3) || !pres_shape(N, K, planes, s) || M % (32 * s.pb) != 0 || M * (long)(K > N ? K : N) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  PresParams p; int grid;
  pres_geom(M, N, s, p, grid);
  const long per_wg = cdiv(p.stripes, p.Q * p.nxcd);
  if (per_wg * 2 * s.pb > kMaxLaneTermsP) return CRNN_ERR_UNSUPPORTED;
  return CRNN_OK;
}
template <int KST, int NPL, int PB, int R, int OCC, int NPA, int DB>
int pres_launch(const PresParams& p, int grid, hipStream_t stream) {
  constexpr int lds = R * NPL * PB * 4096 + 4 * PB * 4096 + 4 * DB * PB * 4096;
  static_assert(lds * OCC <= 160 * 1024, "LDS per CU");
  CRNN_LDS_ATTR((pres_dgrad_kernel<KST, NPL, PB, R, OCC, NPA, DB>), lds);
  hipLaunchKernelGGL((pres_dgrad_kernel<KST, NPL, PB, R, OCC, NPA, DB>), dim3(grid), dim3(256), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
template <int NPL, int NPA>
int pres_dispatch(const PresParams& p, const PresShape& s, int grid, hipStream_t stream) {
  if constexpr (NPL == 2) {
    if (s.kst == 8) return pres_launch<8, 2, PRES_PB8, PRES_R8, 1, NPA, PRES_DB8>(p, grid, stream);
    return pres_launch<4, 2, PRES_PB4, PRES_R4, PRES_OCC4, NPA, 2>(p, grid, stream);
  } else {
    if (s.kst == 4) return pres_launch<4, 3, 1, 8, 1, NPA, 2>(p, grid, stream);
    return CRNN_ERR_UNSUPPORTED;
  }
}

}  // namespace

// 0 if the planes-resident data-gradient kernel handles (M pixels, N result channels, K reduction) with `planes` planes per operand, else -3:
// K in {256, 512} (three planes: 256), N a multiple of 128 up to 1024, whole stripes (M % 32 == 0)
extern "C" int crnn_gemm_pres_supported(long M, int N, int K, int planes) { return pres_supported(M, N, K, planes); }
// rows of the partial statistics [rows][2][N] the kernel writes (every row and column of that block is written)
extern "C" int crnn_gemm_pres_stat_rows(long M, int N, int K, int planes) {
  if (pres_supported(M, N, K, planes) != CRNN_OK) return 0;
  PresShape s; pres_shape(N, K, planes, s);
  PresParams p; int grid;
  pres_geom(M, N, s, p, grid);
  return p.Q * p.nxcd;
}
// da[M][N] = dq[M][K] . w[N][K]^T from the planes of dq (dq_planes: plane pl of dq[m][k] at dq_planes[pl * plane_stride + m * K + k], bf16 words, the
// words of crnn_split3_planes; planes = 2 | 3), w fp32 [N][K]; stat_partials [crnn_gemm_pres_stat_rows][2][N] = partial sums of gy and gy * xhat,
// gy = da where 0 < d * scale + shift < 6 (d [M][N] fp32, bnstate = [mean|var|scale|shift] x N); a_planes (may be NULL): a_count (2 | 3) planes of
// a = ReLU6(d * scale + shift) [a_count][M][N] at element stride a_stride -- the operand of the same convolution's weight gradient.
extern "C" int crnn_gemm_pres_bnstats(const void* dq_planes, long plane_stride, const float* w, float* da, long M, int N, int K, int planes, const float* d,
                                      const float* bnstate, float* stat_partials, void* a_planes, long a_stride, int a_count, hipStream_t stream) {
  if (!dq_planes || !w || !da || !d || !bnstate || !stat_partials || (planes != 2 && planes != 3)) return CRNN_ERR_ARG;
  if (a_planes && a_count != 2 && a_count != 3) return CRNN_ERR_ARG;
  CRNN_TRY(pres_supported(M, N, K, planes));
  if ((((uintptr_t)dq_planes | (uintptr_t)da | (uintptr_t)d | (uintptr_t)stat_partials | (uintptr_t)a_planes) & 15) || ((uintptr_t)w & 15) || ((plane_stride | a_stride) & 7)) return CRNN_ERR_UNSUPPORTED;
  PresShape s; pres_shape(N, K, planes, s);
  PresParams p{};
  p.XP = reinterpret_cast<const bf16_t*>(dq_planes); p.xps = plane_stride; p.W = w; p.ldw = K; p.Y = da; p.D = d; p.bnstate = bnstate; p.stats = stat_partials;
  p.AP = reinterpret_cast<bf16_t*>(a_planes); p.aps = a_stride;
  p.M = (int)M; p.N = N; p.K = K;
  int grid; pres_geom(M, N, s, p, grid);
  const int npa = a_planes ? a_count : 0;
  if (planes == 2) return npa == 0 ? pres_dispatch<2, 0>(p, s, grid, stream) : npa == 2 ? pres_dispatch<2, 2>(p, s, grid, stream) : pres_dispatch<2, 3>(p, s, grid, stream);
  return npa == 0 ? pres_dispatch<3, 0>(p, s, grid, stream) : npa == 2 ? pres_dispatch<3, 2>(p, s, grid, stream) : pres_dispatch<3, 3>(p, s, grid, stream);
}
