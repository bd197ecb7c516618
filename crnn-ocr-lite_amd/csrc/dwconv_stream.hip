// Depthwise 3x3 convolution (reference utils.py:44, DepthwiseConv2D(3x3, padding='same', no bias)) of a bf16 NHWC map as a ROW STREAM:
//     out[n][y][x][c] = sum_{i,j} k[3i+j][c] * in[n][y+i-1][x+j-1][c]      (+ BatchNorm statistics of out, or the folded BN + ReLU6)
//
// The halo-tile kernel (conv.hip) alternates "fill a tile" and "compute it" inside a workgroup and relies on three workgroups per
// CU to overlap the two; its waves are parked a third of the time.  Here one workgroup walks a band of image rows top to bottom:
//   * a LOADER wave puts whole rows (all channels: W*C*2 contiguous bytes, 9 KiB for every block of the CRNN) into an LDS ring with
//     global_load_lds (16 B per lane, no VGPR staging) D rows ahead, under a counted s_waitcnt vmcnt(N);
//   * up to 9 COMPUTE waves own one 16-byte column (pixel x, 8 channels) of the row each.  The three-row window is never held: when
//     input row r arrives it is read ONCE from LDS (x-1, x, x+1: three ds_read_b128) and contributes its taps to three running
//     sums - output rows r-1 (taps 6..8: now complete, rounded, stored), r (taps 3..5) and r+1 (taps 0..2).  Every output is the
//     same fp32 fma chain (tap order 0..8) as the tile kernel's: bit-identical results;
//   * one s_barrier per row hands the row over; the ring needs D+1 slots only because nothing is re-read.
// No halo re-reads inside a band (a band boundary re-reads 2 rows), zero padding = a 16-byte zero chunk in LDS for the row ends and
// peeled first/last steps for the rows above and below the image.  Narrow maps run NS bands of the same image side by side in one
// workgroup so that the step row always fills the 9 waves (block 2: 2 x 36 px x 64 ch; blocks 3-7: W*C = 4608).
// BatchNorm statistics: per-lane fp32 sums over the band, combined over the pixel columns through LDS in a fixed order: one
// partial row [2][C] per workgroup.
//
// PROLOGUE form (round 4, training): the input is the PREVIOUS block's pointwise output q and the kernel applies that block's
// BatchNorm-2 + ReLU6 + Dropout(.1) (utils.py:48-56: x = drop(relu6(q * scale + shift)), the arithmetic of bn_act_pool_drop_kernel
// bit for bit) to every row after it has landed in LDS -- the block output x is never written to or read from HBM (one read pass +
// one write pass of the largest tensors of the step less per un-pooled block).  Two TRANSFORM waves rewrite the row that arrived one
// step ahead in place (bf16 again) while the compute waves work on the current row; the ring is one slot deeper so that as many
// rows stay in flight; the 12 waves are dealt over the SIMDs so that the transform waves share theirs with fewer compute waves.
// The dropout decisions arrive as one keep byte per chunk (crnn_dropout_keep_bytes), brought into LDS by the loader wave next to
// the row.  How it got there (profiles/r04_*, four forward launches, cold): two-pass path 0.51 ms; transform waves that also evaluate
// the counter RNG 0.43 (a wave issues one VALU operation per 4 cycles: ~100 operations per chunk x 5 chunks per lane outlast the
// row's 2400-cycle HBM budget; interleaving the five RNG chains changed nothing -- issue, not latency); keep bytes loaded by the
// transform waves themselves 0.35 (a vmcnt wait per row); the re-forming done by the nine compute waves on their own chunks 0.50
// (two dependent LDS round trips per row in every wave, register spills); keep bytes through the loader's DMA: this version.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef CRNN_DWS_EXP
#define CRNN_DWS_EXP 0      // experiment builds: 1 = no DMA, 2 = no stores, 4 = no fmas, 8 = nontemporal stores, 16 = nontemporal loads, 32 = no transform (prologue form)
#endif

struct DwsParams {
  const unsigned char* x; const float* k; unsigned char* out; float* partials; const float* bnstate;
  int H, W, C, HB, NS, nwgb, flip, cols, rowbytes, wmaj;
  int nsplit, cppw;          // channel ranges per row (fp32 form: 2 for rows of 18 KiB), 16-byte columns per pixel of one range
  int xstride = 1;           // workgroup-id distance between the channel ranges of one band (see the kernel)
#ifdef CRNN_DWS_TRACE
  unsigned long long* trace = nullptr; // timing build only: [workgroup][4] s_memrealtime stamps (entry, first row landed, last step done, statistics written)
#endif
  // prologue form: BatchNorm state [mean|var|scale|shift] of the producer, dropout of its output
  const float* pro_bn; const unsigned char* keep; float rate;
};

#ifndef CRNN_DWS_NT
#define CRNN_DWS_NT 1       // round 5: the input map is not read again before the backward pass -> nontemporal LDS-DMA (cache policy only; see conv.hip)
#endif
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, (CRNN_DWS_NT != 0 || (CRNN_DWS_EXP & 16) != 0) ? 2 : 0);
}
__device__ __forceinline__ void widen8(const u32x4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// element forms of a 16-byte chunk: 8 bf16 (EPC = 8) or 4 fp32 (EPC = 4; round 4, the parity mode's prologue form)
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

constexpr int kDwsMaxWaves = 9;     // compute waves (576 columns of 16 bytes)

// NI: 1 KiB DMA instructions per step row; D: rows in flight; EPI: out = ReLU6(conv * scale + shift) (inference), no statistics;
// PRO: prologue form (the input is q of the previous block: BatchNorm-2 + ReLU6 [+ dropout: DROP] applied in LDS)
constexpr int kDwsKeepNI = 3;        // prologue form with dropout: 4-byte DMA instructions per step row for the keep bytes (<= 768 chunks)
// role of wave slot w in the prologue form (12 waves, SIMD = w mod 4): 9 compute waves, the loader, two transform waves --
//   SIMD 0: w0 w4 w8 = C C C;  SIMD 1: w1 w5 w9 = C T L;  SIMD 2: w2 w6 w10 = C C C;  SIMD 3: w3 w7 w11 = C C T
// (the transform wave with five chunks per lane shares its SIMD with one compute wave, the one with four with two; letting the loader
// re-form a third of the row between its issues was measured slower: 0.35 against 0.31 ms for the four launches -- its DMA issues slip)
__device__ __forceinline__ int dws_pro_role(int w, int& idx) {   // 0 = compute, 1 = transform, 2 = loader; idx: compute index / transformer index
  if (w == 9) { idx = 0; return 2; }
  if (w == 5) { idx = 0; return 1; }
  if (w == 11) { idx = 1; return 1; }
  idx = w < 5 ? w : (w < 9 ? w - 1 : 8);                        // 0 1 2 3 4 | 6 7 8 -> 5 6 7 | 10 -> 8
  return 0;
}
constexpr int kDwsProChunks = 5;     // 16-byte chunks per transforming lane and row: chunks j * 128 + 64 * (transformer index) + lane
constexpr int kDwsProStride = 128;

// Re-forming x = Dropout(ReLU6(q * scale + shift)) of a transformer's chunks of one row, in place (bf16 -> bf16): bn_act_pool_drop_kernel's
// arithmetic bit for bit; dropout decisions from the row's keep bytes (one per chunk) next to it in LDS.
template <bool DROP, bool F32 = false>
struct DwsXform {
  static constexpr int EPC = F32 ? 4 : 8;
  int coff[kDwsProChunks]; bool cact[kDwsProChunks]; int nj;
  f32x2_t sc[EPC / 2], sh[EPC / 2]; float ik;
  __device__ __forceinline__ void init(const DwsParams& p, int tw, int lane, int c0) {
    nj = 0;
#pragma unroll
    for (int j = 0; j < kDwsProChunks; ++j) {
      const int c = j * kDwsProStride + tw * 64 + lane;
      cact[j] = c < p.cols;
      coff[j] = cact[j] ? c : 0;
      if (j * kDwsProStride + tw * 64 < p.cols) nj = j + 1;     // (uniform) chunks this wave has
    }
    // kDwsProStride % cppw == 0 (launcher): every chunk of a lane holds the same channels
    const int ch0 = c0 + ((tw * 64 + lane) % p.cppw) * EPC;
#pragma unroll
    for (int e = 0; e < EPC / 2; ++e) {
      sc[e] = (f32x2_t){p.pro_bn[2 * p.C + ch0 + 2 * e], p.pro_bn[2 * p.C + ch0 + 2 * e + 1]};
      sh[e] = (f32x2_t){p.pro_bn[3 * p.C + ch0 + 2 * e], p.pro_bn[3 * p.C + ch0 + 2 * e + 1]};
    }
    ik = DROP ? 1.f / (1.f - p.rate) : 1.f;                     // (spelled as bn_act_pool_drop_kernel spells it: the same bits)
  }
  __device__ __forceinline__ void run(unsigned char* sb, const unsigned char* kbp) const {
    if (CRNN_DWS_EXP & 32) return;                               // experiment build: no re-forming
    u32x4 v[kDwsProChunks]; uint32_t kc[kDwsProChunks];
    // keep bytes: one per group of 8 elements = per chunk (bf16) | per pair of chunks (fp32: the chunk's nibble)
#pragma unroll
    for (int j = 0; j < kDwsProChunks; ++j) {
      v[j] = *reinterpret_cast<const u32x4*>(sb + coff[j] * 16);
      kc[j] = DROP ? (F32 ? ((uint32_t)kbp[coff[j] >> 1] >> ((coff[j] & 1) * 4)) : (uint32_t)kbp[coff[j]]) : 0xffu;
    }
    const f32x2_t ik2 = (f32x2_t){ik, ik};
#pragma unroll
    for (int j = 0; j < kDwsProChunks; ++j) {
      if (j >= nj) continue;
      u32x4 o;
      if (F32) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2_t x2 = (f32x2_t){__uint_as_float(v[j][2 * q]), __uint_as_float(v[j][2 * q + 1])};
          f32x2_t y = __builtin_elementwise_fma(x2, sc[q], sh[q]);
          y = (f32x2_t){relu6f(y.x), relu6f(y.y)};
          if (DROP) {
            y = y * ik2;                                           // y >= 0: a dropped element is +0 like y * 0
            y = (f32x2_t){(kc[j] >> (2 * q)) & 1u ? y.x : 0.f, (kc[j] >> (2 * q + 1)) & 1u ? y.y : 0.f};
          }
          o[2 * q] = __float_as_uint(y.x); o[2 * q + 1] = __float_as_uint(y.y);
        }
        if (cact[j]) *reinterpret_cast<u32x4*>(sb + coff[j] * 16) = o;
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2_t x2 = (f32x2_t){__uint_as_float(v[j][q] << 16), __uint_as_float(v[j][q] & 0xffff0000u)};
        f32x2_t y = __builtin_elementwise_fma(x2, sc[q % (EPC / 2)], sh[q % (EPC / 2)]);   // per element fmaf(x, scale, shift): v_pk_fma_f32
        y = (f32x2_t){relu6f(y.x), relu6f(y.y)};
        if (DROP) {
          y = y * ik2;                                             // per element y * inv_keep (v_pk_mul_f32), as the stand-alone pass
          // dropped elements as an AND mask on the packed pair (y >= 0: y * 0 and 0 are the same bits): sign-extending 1-bit extracts
          const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)kc[j], 2 * q, 1), hi = (uint32_t)__builtin_amdgcn_sbfe((int)kc[j], 2 * q + 1, 1);
          o[q] = pack2_bf16(y.x, y.y) & __builtin_amdgcn_perm(hi, lo, 0x07060100u);
        } else {
          o[q] = pack2_bf16(y.x, y.y);
        }
      }
      if (cact[j]) *reinterpret_cast<u32x4*>(sb + coff[j] * 16) = o;
    }
  }
};

// NI = compute waves = 1 KiB DMA instructions per step row: 9 for every block of the CRNN at image width 32 (the 9 KiB step row); 5..8 (round 5) for the
// step rows of other image widths (e.g. width 48: 416 columns = 7 waves), which used to fall back to the halo-tile kernels
template <int NI, int D, bool EPI, bool PRO, bool DROP = false, bool F32 = false>
__global__ __launch_bounds__((NI + (PRO ? 3 : 1)) * 64) void dw_fwd_stream_kernel(DwsParams p) {
  static_assert(NI == kDwsMaxWaves || !PRO, "the prologue form runs on the 9 KiB step row");
  constexpr int EPC = F32 ? 4 : 8, ES = F32 ? 4 : 2;            // elements per 16-byte chunk, bytes per element
  constexpr int NR = D + 1, SLOT = NI * 1024;
  constexpr int NIT = NI + (DROP ? kDwsKeepNI : 0);            // DMA instructions per step row
  static_assert((D - 1) * NIT <= 63, "vmcnt is a 6-bit counter");
  static_assert(!(PRO && EPI) && !(DROP && !PRO), "the prologue form is the training form");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncw = PRO ? kDwsMaxWaves : (int)(blockDim.x >> 6) - 1;
  int ridx = 0;
  const int role = PRO ? dws_pro_role(wave0, ridx) : 0;
  // `wave`: index of a compute wave among the compute waves; == ncw for the loader (the transform waves take their own branch first)
  const int wave = PRO ? (role == 0 ? ridx : ncw) : wave0;
  constexpr int KOFF = NR * SLOT + 64;                         // prologue form: NR x 1 KiB of keep bytes behind the zero chunk
  // workgroup = (image, band, channel range).  The ranges of a band sit p.xstride workgroup ids apart (8 = the XCD count: workgroup i runs on
  // XCD i mod 8, so the ranges of one image row share an XCD and its L2 / memory channel queue; 1 = adjacent ids)
  const int xs = p.xstride, blk = (int)blockIdx.x;
  const int grp = blk / (xs * p.nsplit), rem = blk - grp * (xs * p.nsplit);
  const int split = rem / xs;
  const int bidx = grp * xs + (rem - split * xs);                             // statistics row of the workgroup's band
  const int img = bidx / p.nwgb, wb = bidx - img * p.nwgb;
  const int c0 = split * (p.C / p.nsplit);         // first channel of the range
  const int r0 = wb * p.NS * p.HB;                 // first output row of sub-band 0
  const int steps = p.HB + 2;                      // input rows r0-1 .. r0+HB of every sub-band
  const int zoff = NR * SLOT;                      // 16 zero bytes (the pixels left of x = 0 and right of x = W-1)
  if (tid < 4) reinterpret_cast<unsigned*>(lds + zoff)[tid] = 0u;
#ifdef CRNN_DWS_TRACE
  unsigned long long* const trc = (p.trace && tid == 0) ? p.trace + (long)blockIdx.x * 4 : nullptr;
  if (trc) trc[0] = __builtin_amdgcn_s_memrealtime();
#define DWS_TRC(i) do { if (trc) trc[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DWS_TRC(i) do {} while (0)
#endif

  if (PRO && role == 1) {
    // ------------------------------------------------------------------ transform waves (prologue form): half of every row each
    DwsXform<DROP, F32> xf;
    xf.init(p, ridx, lane, c0);
    auto xform = [&](int slot) { xf.run(lds + slot * SLOT, lds + KOFF + slot * 1024); };
    __builtin_amdgcn_s_barrier();                                // P: row 0 has landed
    xform(0);
    int slot = 1;                                                // slot of row t + 1
    for (int t = 0; t < steps; ++t) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // row t + 1 has landed; row t is transformed
      if (t + 1 < steps) xform(slot);
      slot = slot + 1 == NR ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!EPI && p.partials) __builtin_amdgcn_s_barrier();
  } else if (wave == ncw) {
    // ------------------------------------------------------------------ loader wave
    const unsigned char* gx = p.x + (long)img * p.H * p.rowbytes;
    int rowfirst[NI], within[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // chunk j of the step row: sub-band j / (W cppw), pixel (j / cppw) % W of it, 16-byte column j % cppw of the channel range
      int j = i * 64 + lane; j = j < p.cols ? j : p.cols - 1;    // past the step row: re-read its last chunk (lands in the slot's unused tail)
      const int sw = p.W * p.cppw, s = j / sw, jj = j - s * sw, px = jj / p.cppw, o = jj - px * p.cppw;
      rowfirst[i] = r0 + s * p.HB - 1; within[i] = (px * p.C + c0) * ES + o * 16;
    }
    // prologue form with dropout: the keep bytes of the step row (one per 16-byte chunk, same order), 4 bytes per lane and instruction
    int krow[kDwsKeepNI], kwithin[kDwsKeepNI];
    const int rowcols = p.W * (p.C >> 3);                      // keep bytes per image row (one per 8 elements)
    const unsigned char* gk = DROP ? p.keep + (long)img * p.H * rowcols : nullptr;
    if (DROP) {
      const int kpp = p.cppw * EPC / 8, kcols = p.cols * EPC / 8;   // keep bytes per pixel of the channel range, per step row
#pragma unroll
      for (int i = 0; i < kDwsKeepNI; ++i) {
        int b = (i * 64 + lane) * 4;
        if (b >= kcols) b = kcols - 4;                         // past the step row: its last dword again (lands in the unused tail)
        const int sw = p.W * kpp, s2 = b / sw, bb = b - s2 * sw, px = bb / kpp, o = bb - px * kpp;
        krow[i] = r0 + s2 * p.HB - 1; kwithin[i] = px * (p.C >> 3) + (c0 >> 3) + o;
      }
    }
    auto issue = [&](int t, int slot) {
      t = t < steps ? t : steps - 1;                           // past the end: the last row again, into a slot nobody reads any more
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int row = rowfirst[i] + t;
        row = row < 0 ? 0 : (row >= p.H ? p.H - 1 : row);      // rows outside the image: any valid row (the compute waves substitute zeros)
        if (!(CRNN_DWS_EXP & 1)) glds16(gx + (long)row * p.rowbytes + within[i], lds + slot * SLOT + i * 1024);
      }
      if (DROP) {
#pragma unroll
        for (int i = 0; i < kDwsKeepNI; ++i) {
          int row = krow[i] + t;
          row = row < 0 ? 0 : (row >= p.H ? p.H - 1 : row);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gk + (long)row * rowcols + kwithin[i]),
                                           (__attribute__((address_space(3))) void*)(lds + KOFF + slot * 1024 + i * 256), 4, 0, 0);
        }
      }
    };
#pragma unroll
    for (int t = 0; t < D; ++t) issue(t, t);
    int slot = D;                                              // slot of row t + D
    if (PRO) {                                                 // the rows are re-formed one row ahead of their use
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NIT) : "memory");   // row 0 has landed
      __builtin_amdgcn_s_barrier();
    }
    for (int t = 0; t < steps; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - (PRO ? 2 : 1)) * NIT) : "memory");   // row t (prologue form: row t + 1) has landed
      __builtin_amdgcn_s_barrier();
      issue(t + D, slot);                                      // the slot row t-1 has just released
      slot = slot + 1 == NR ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // ring free: the statistics reduction may use it
    if (!EPI && p.partials) __builtin_amdgcn_s_barrier();
  } else {
    // ------------------------------------------------------------------ compute waves
    const int col = wave * 64 + lane;                          // also this thread's index among the compute threads
    const bool act = col < p.cols;
    const int ccol = act ? col : p.cols - 1;                   // idle lanes of the last wave shadow the last column
    const int cpp = p.cppw;                                    // 16-byte columns per pixel (of the channel range)
    const int pxs = ccol / cpp, oct = ccol - pxs * cpp;
    const int sub = pxs / p.W, px = pxs - sub * p.W;
    const int offC = ccol * 16, pitch = cpp * 16;              // LDS bytes between horizontally adjacent pixels
    const int gpitch = p.C * ES;                               // the same in global memory
    const int ch0 = c0 + oct * EPC;                            // first channel of the lane
    const bool hasL = px > 0, hasR = px < p.W - 1;
    float kw[9][EPC];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* kp = p.k + (long)(p.flip ? 8 - t : t) * p.C + ch0;
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) {
        const float4 a = *reinterpret_cast<const float4*>(kp + 4 * h);
        kw[t][4 * h] = a.x; kw[t][4 * h + 1] = a.y; kw[t][4 * h + 2] = a.z; kw[t][4 * h + 3] = a.w;
      }
    }
    float esc[EPC], esh[EPC];
    if (EPI) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { esc[e] = p.bnstate[2 * p.C + ch0 + e]; esh[e] = p.bnstate[3 * p.C + ch0 + e]; }
    }
    float X0[EPC], X1[EPC], X2[EPC], s[EPC], ss[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { X0[e] = X1[e] = X2[e] = 0.f; s[e] = ss[e] = 0.f; }
    const int rsub = r0 + sub * p.HB;                          // this lane's first output row
    if (PRO) __builtin_amdgcn_s_barrier();                     // P (the transform waves take row 0 now)
    unsigned char* orow = p.out + ((long)img * p.H + rsub) * p.rowbytes + px * gpitch + ch0 * ES;
    // window-major output (EPI only; crnn_dwconv3x3_fwd_stream_ex out_order 1): pixel (y, x) is row ((y/2) (W/2) + x/2) 4 + (y&1) 2 + (x&1) of
    // the image -- the four pixels of a 2x2 pooling window are consecutive rows for the pointwise GEMM whose epilogue pools them
    unsigned char* const owin = p.out + (long)img * p.H * p.rowbytes + ((px >> 1) * 4 + (px & 1)) * gpitch + ch0 * ES;
    int slot = 0;
    // one step: input row t of the band (image row rsub - 1 + t) -> taps 6..8 of output t-2 (A: complete), 3..5 of t-1 (Bc), 0..2 of t (Cn)
    auto step = [&](int t, float (&A)[EPC], float (&Bc)[EPC], float (&Cn)[EPC], bool edge) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* sb = lds + slot * SLOT;
      u32x4 vL = *reinterpret_cast<const u32x4*>(hasL ? sb + offC - pitch : lds + zoff);
      u32x4 vC = *reinterpret_cast<const u32x4*>(sb + offC);
      u32x4 vR = *reinterpret_cast<const u32x4*>(hasR ? sb + offC + pitch : lds + zoff);
      slot = slot + 1 == NR ? 0 : slot + 1;
      if (edge) {
        const int g = rsub - 1 + t;
        if (g < 0 || g >= p.H) { vL = (u32x4)(0u); vC = (u32x4)(0u); vR = (u32x4)(0u); }
      }
      if (!(CRNN_DWS_EXP & 4)) {
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
      if (t >= 2 && act) {
        if (EPI) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) A[e] = relu6f(fmaf(A[e], esc[e], esh[e]));
        } else {
#pragma unroll
          for (int e = 0; e < EPC; ++e) { s[e] += A[e]; ss[e] = fmaf(A[e], A[e], ss[e]); }
        }
        const u32x4 o = packE<EPC>(A);
        unsigned char* dst = orow + (long)(t - 2) * p.rowbytes;
        if (EPI && p.wmaj) { const int y = rsub + t - 2; dst = owin + (long)(y >> 1) * 2 * p.rowbytes + (y & 1) * 2 * gpitch; }
        if (CRNN_DWS_EXP & 8) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dst));
        // (write-through stores -- sc0 sc1, sc1 or sc0 -- were measured in round 5: 3.2-3.3 TB/s instead of 4.3 cold, +0.15 ms in the step; plain stores stay)
        else if (!(CRNN_DWS_EXP & 2)) *reinterpret_cast<u32x4*>(dst) = o;
      }
    };
    step(0, X1, X2, X0, true);
    DWS_TRC(1);
    step(1, X2, X0, X1, false);
    int t = 2;
    for (; t + 3 <= steps - 1; t += 3) {          // whole groups of three that do not contain the last step
      step(t, X0, X1, X2, false);
      step(t + 1, X1, X2, X0, false);
      step(t + 2, X2, X0, X1, false);
    }
    for (int r = 0; t < steps; ++t, ++r) {        // 1..3 remaining steps, the last one reads the row below the band
      if (r == 0) step(t, X0, X1, X2, true);
      else if (r == 1) step(t, X1, X2, X0, true);
      else step(t, X2, X0, X1, true);
    }
    DWS_TRC(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!EPI && p.partials) {
      float* red = reinterpret_cast<float*>(lds);
      const int nthr = ncw * 64;
      if (cpp <= 64 && (cpp & (cpp - 1)) == 0) {
        // lanes l, l + cpp, l + 2 cpp, ... of a wave hold the same channels of different pixels: xor-shuffle them together, then one
        // record [wave][octet][sum 8 | sumsq 8] per wave in LDS and a fixed-order sum over the waves
        for (int o = cpp; o < 64; o <<= 1) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) { s[e] += __shfl_xor(s[e], o, 64); ss[e] += __shfl_xor(ss[e], o, 64); }
        }
        constexpr int REC = 2 * EPC;                              // record per (wave, column of a pixel): sums | sums of squares
        const int cw = cpp * EPC;                                 // channels of the workgroup's range (bf16: all C)
        if (lane < cpp) {
          float* dst = red + (wave * cpp + lane) * REC;
#pragma unroll
          for (int e = 0; e < EPC; e += 4) {
            *reinterpret_cast<float4*>(dst + e) = make_float4(s[e], s[e + 1], s[e + 2], s[e + 3]);
            *reinterpret_cast<float4*>(dst + EPC + e) = make_float4(ss[e], ss[e + 1], ss[e + 2], ss[e + 3]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // wave w holds columns (64 w + lane) % cpp: with cpp < 64 every wave holds all of them, with cpp == 64 likewise (column = lane)
        for (int ch = col; ch < 2 * cw; ch += nthr) {
          const int c = ch < cw ? ch : ch - cw;
          const float* src = red + (c / EPC) * REC + (ch < cw ? 0 : EPC) + (c % EPC);
          float a = 0.f;
          for (int w = 0; w < ncw; ++w) a += src[w * cpp * REC];
          p.partials[(long)bidx * 2 * p.C + (ch < cw ? 0 : p.C) + c0 + c] = a;
        }
      } else {
        constexpr int REC = 2 * EPC;
        const int cw = cpp * EPC;
        if (act) {   // [cols][REC]
#pragma unroll
          for (int e = 0; e < EPC; e += 4) {
            *reinterpret_cast<float4*>(red + col * REC + e) = make_float4(s[e], s[e + 1], s[e + 2], s[e + 3]);
            *reinterpret_cast<float4*>(red + col * REC + EPC + e) = make_float4(ss[e], ss[e + 1], ss[e + 2], ss[e + 3]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int npx = p.NS * p.W;
        for (int ch = col; ch < 2 * cw; ch += nthr) {             // ch < cw: sum, else sum of squares; pixel columns in ascending order
          const int c = ch < cw ? ch : ch - cw;
          const float* src = red + (c / EPC) * REC + (ch < cw ? 0 : EPC) + (c % EPC);
          float a = 0.f;
          for (int q = 0; q < npx; ++q) a += src[(long)q * cpp * REC];
          p.partials[(long)bidx * 2 * p.C + (ch < cw ? 0 : p.C) + c0 + c] = a;
        }
      }
    }
    DWS_TRC(3);
  }
}

struct DwsGeom { int NS, nwgb, HB, cols, ncw, nsplit, cppw, xstride; bool ok; };
constexpr int kDwsMinWaves = 5;      // step rows that fill fewer compute waves stay with the halo-tile kernels
// es: bytes per element (2: bf16 maps, 4: fp32 maps).  A row of W * C * es bytes is cut into the smallest number of channel ranges (whole groups of 8
// channels) whose 16-byte columns fit the nine compute waves; narrow ranges run NS bands of the same image side by side.  Every block of the CRNN at image
// width 32: bf16 one range of 576 columns, fp32 two.  Round 5: any step row of 257..576 columns (five to nine compute waves, NI = waves), e.g. image width
// 48 (416 columns, bf16 blocks 3..7 as two ranges) or 64 (544) -- these shapes ran the halo-tile kernels at about half the rate.
DwsGeom dws_geom(int B, int H, int W, int C, int es = 2) {
  DwsGeom g; g.ok = false; g.NS = g.nwgb = g.HB = g.cols = g.ncw = g.cppw = 0; g.nsplit = 1; g.xstride = 1;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return g;
  const int epc = 16 / es;
  int ns = 0;
  for (int n = 1; n <= C / 8; ++n) {               // smallest number of channel ranges that fits the step row
    if ((C / 8) % n) continue;
    if ((long)W * C / epc / n <= kDwsMaxWaves * 64) { ns = n; break; }
  }
  if (!ns) return g;
  const long cols1 = (long)W * C / epc / ns;       // 16-byte columns of one channel range
  int NS = (int)(kDwsMaxWaves * 64 / cols1);
  while (NS > 1 && H % NS) --NS;
  const int cols = (int)(NS * cols1), ncw = (cols + 63) / 64;
  if (ncw < kDwsMinWaves) return g;
  g.nsplit = ns; g.cppw = C / epc / ns;
  // bands per image over workgroups: enough workgroups for the chip, bands of at least 8 rows
  int nwgb = 1;
#ifndef CRNN_DWS_WGS
#define CRNN_DWS_WGS 256
#endif
  const int want = CRNN_DWS_WGS;
  for (int n = 1; n <= H / NS; ++n) {
    if ((H / NS) % n || H / NS / n < 8) continue;
    nwgb = n;
    if ((long)B * n * ns >= want) break;
  }
  if (H % (NS * nwgb)) return g;
  g.NS = NS; g.nwgb = nwgb; g.HB = H / (NS * nwgb); g.cols = cols; g.ncw = ncw; g.ok = true;
  // bf16 maps: the ranges of a band 8 workgroup ids apart (one XCD); fp32 maps keep round 4's adjacent ids (measured with them)
  g.xstride = (es == 2 && ns > 1 && ((long)B * nwgb) % 8 == 0) ? 8 : 1;
  return g;
}

#ifndef CRNN_DWS_D
#define CRNN_DWS_D 4
#endif
constexpr int kDwsD = CRNN_DWS_D;     // rows in flight per workgroup (2..7 measured: 4.4 / 4.6 / 4.6 / 4.5 TB/s at 2 / 3 / 4 / 7)

template <int NI, bool EPI, bool F32>
int dws_launch_ni(const DwsParams& p, const DwsGeom& g, int B, hipStream_t stream) {
  constexpr int lds = (kDwsD + 1) * NI * 1024 + 64;
  CRNN_LDS_ATTR((dw_fwd_stream_kernel<NI, kDwsD, EPI, false, false, F32>), lds);
  hipLaunchKernelGGL((dw_fwd_stream_kernel<NI, kDwsD, EPI, false, false, F32>), dim3(B * g.nwgb * g.nsplit), dim3((NI + 1) * 64), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
template <bool EPI, bool F32 = false>
int dws_launch(const DwsParams& p, const DwsGeom& g, int B, hipStream_t stream) {
  switch (g.ncw) {
    case 9: return dws_launch_ni<9, EPI, F32>(p, g, B, stream);
    case 8: return dws_launch_ni<8, EPI, F32>(p, g, B, stream);
    case 7: return dws_launch_ni<7, EPI, F32>(p, g, B, stream);
    case 6: return dws_launch_ni<6, EPI, F32>(p, g, B, stream);
    case 5: return dws_launch_ni<5, EPI, F32>(p, g, B, stream);
    default: return CRNN_ERR_UNSUPPORTED;
  }
}
// prologue form: one more row in flight (the transform waves need row t + 1 landed when the compute waves take row t)
constexpr int kDwsProD = kDwsD + 1;
bool dws_pro_ok(const DwsGeom& g, int B, int H, int W, int C, int es = 2) {
  const int cpp = g.cppw, kpp = cpp * (16 / es) / 8, kcols = g.cols * (16 / es) / 8;   // keep bytes (one per 8 elements) per pixel of the range, per step row
  // the keep bytes of a step row travel as 4-byte DMA pieces: whole dwords per pixel range, at most kDwsKeepNI * 256 of them
  return g.ok && g.ncw == kDwsMaxWaves && (es == 4 || g.nsplit == 1) && cpp > 0 && kDwsProStride % cpp == 0 && g.cols <= kDwsProChunks * kDwsProStride &&
         (es == 2 ? (W * cpp) % 4 == 0 : kpp % 4 == 0) && kcols % 4 == 0 && kcols <= kDwsKeepNI * 256 && (long)B * H * W * (C / 8) < (1L << 31);
}
template <bool DROP, bool F32 = false>
int dws_launch_pro(const DwsParams& p, const DwsGeom& g, int B, hipStream_t stream) {
  constexpr int lds = (kDwsProD + 1) * (9 + 1) * 1024 + 64;     // rows + 1 KiB of keep bytes per slot
  CRNN_LDS_ATTR((dw_fwd_stream_kernel<9, kDwsProD, false, true, DROP, F32>), lds);
  hipLaunchKernelGGL((dw_fwd_stream_kernel<9, kDwsProD, false, true, DROP, F32>), dim3(B * g.nwgb * g.nsplit), dim3((kDwsMaxWaves + 3) * 64), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

}  // namespace

// CRNN_OK when crnn_dwconv3x3_fwd_stream takes the shape (bf16 storage; dws_geom: 257..576 sixteen-byte columns per step row after cutting the row into
// channel ranges / laying bands side by side), else CRNN_ERR_UNSUPPORTED: the caller uses the halo-tile kernel (crnn_dwconv3x3_fwd_ex).
extern "C" int crnn_dwconv_fwd_stream_supported(int B, int H, int W, int C) { return dws_geom(B, H, W, C).ok ? CRNN_OK : CRNN_ERR_UNSUPPORTED; }
// rows [2][C] of statistics partials the launch writes (one per workgroup)
extern "C" int crnn_dwconv_fwd_stream_rows(int B, int H, int W, int C) { DwsGeom g = dws_geom(B, H, W, C); return g.ok ? B * g.nwgb : 0; }
// out = dwconv3x3(x, k[9][C]) on bf16 NHWC maps (flip = 1: the data gradient).  stat_partials != NULL: [rows][2][C] sums / sums of squares of
// the fp32 results (BatchNorm batch statistics); bnstate != NULL ([mean|var|scale|shift]): out = ReLU6(conv * scale + shift) (inference), no
// statistics.  Results bit-identical to crnn_dwconv3x3_fwd_ex / crnn_dwconv3x3_bn_relu6_fwd.
// out_order 1 (inference form only: bnstate != NULL; H, W even): the rows of `out` are in 2x2-window-major order -- pixel (y, x) of an image is
// its row ((y/2) (W/2) + x/2) 4 + (y&1) 2 + (x&1) -- for crnn_pwconv_fwd_wres_folded_pool(..., pool_rows = 4), whose epilogue max-pools groups of
// four consecutive rows (MaxPooling2D((2,2)) after the pointwise conv, utils.py:52-54, without the un-pooled map ever reaching HBM).
extern "C" int crnn_dwconv3x3_fwd_stream_ex(const void* x, const float* k, void* out, float* stat_partials, const float* bnstate, int B, int H, int W,
                                            int C, int flip, int out_order, hipStream_t stream) {
  if (!x || !k || !out || out_order < 0 || out_order > 1) return CRNN_ERR_ARG;
  if (out_order && (!bnstate || (H & 1) || (W & 1))) return CRNN_ERR_ARG;
  DwsGeom g = dws_geom(B, H, W, C);
  if (!g.ok || (((uintptr_t)x | (uintptr_t)out | (uintptr_t)k | (uintptr_t)bnstate) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 2 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DwsParams p;
  p.x = (const unsigned char*)x; p.k = k; p.out = (unsigned char*)out; p.partials = bnstate ? nullptr : stat_partials; p.bnstate = bnstate;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.NS = g.NS; p.nwgb = g.nwgb; p.flip = flip; p.cols = g.cols; p.rowbytes = W * C * 2; p.wmaj = out_order;
  p.nsplit = g.nsplit; p.cppw = g.cppw; p.xstride = g.xstride;
  p.pro_bn = nullptr; p.keep = nullptr; p.rate = 0.f;
#ifdef CRNN_DWS_TRACE
  { const char* e = getenv("CRNN_DWS_TRACE_PTR"); p.trace = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  return bnstate ? dws_launch<true>(p, g, B, stream) : dws_launch<false>(p, g, B, stream);
}
// Prologue form (training): `q` is the previous block's pointwise output, pro_bnstate its BatchNorm-2 state [mean|var|scale|shift]; the
// kernel convolves x = Dropout(ReLU6(q * scale + shift)) without x ever existing in HBM.  rate > 0: `keep` = the site's keep bytes
// (crnn_dropout_keep_bytes(keep, B*H*W*C/8, rate, seed, layer) for the dropout site of crnn_bn_act_pool_drop_ex); rate == 0: keep may be NULL.
// out / stat_partials bit-identical to crnn_bn_act_pool_drop_ex(q -> x, ph = pw = 1) + crnn_dwconv3x3_fwd_stream(x).
// CRNN_ERR_UNSUPPORTED where crnn_dwconv_fwd_stream_pro_supported says so (the caller materialises x).
extern "C" int crnn_dwconv_fwd_stream_pro_supported(int B, int H, int W, int C) {
  return dws_pro_ok(dws_geom(B, H, W, C), B, H, W, C) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dwconv3x3_fwd_stream_pro(const void* q, const float* pro_bnstate, float rate, const void* keep, const float* k, void* out,
                                             float* stat_partials, int B, int H, int W, int C, hipStream_t stream) {
  if (!q || !pro_bnstate || !k || !out || rate < 0.f || rate >= 1.f || (rate > 0.f && !keep)) return CRNN_ERR_ARG;
  DwsGeom g = dws_geom(B, H, W, C);
  if (!dws_pro_ok(g, B, H, W, C) || (((uintptr_t)q | (uintptr_t)out | (uintptr_t)k | (uintptr_t)pro_bnstate) & 15) || ((uintptr_t)keep & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 2 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DwsParams p;
  p.x = (const unsigned char*)q; p.k = k; p.out = (unsigned char*)out; p.partials = stat_partials; p.bnstate = nullptr;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.NS = g.NS; p.nwgb = g.nwgb; p.flip = 0; p.cols = g.cols; p.rowbytes = W * C * 2; p.wmaj = 0;
  p.nsplit = 1; p.cppw = C / 8;
  p.pro_bn = pro_bnstate; p.keep = (const unsigned char*)keep; p.rate = rate;
  return rate > 0.f ? dws_launch_pro<true>(p, g, B, stream) : dws_launch_pro<false>(p, g, B, stream);
}
// The training form on fp32 maps (dtype CRNN_F32; CRNN_BF16 = crnn_dwconv3x3_fwd_stream without bnstate): rows of 18 KiB as two channel ranges of
// 9 KiB, four channels per lane.  out bit-identical to crnn_dwconv3x3_fwd_ex on fp32 tensors, the statistics one partial row per workgroup band.
extern "C" int crnn_dwconv_fwd_stream_supported_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_fwd_stream_supported(B, H, W, C);
  if (dtype != CRNN_F32) return CRNN_ERR_ARG;
  return dws_geom(B, H, W, C, 4).ok ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dwconv3x3_fwd_stream_dt(const void* x, const float* k, void* out, float* stat_partials, int B, int H, int W, int C, int flip, int dtype,
                                            hipStream_t stream) {
  if (dtype == CRNN_BF16) return crnn_dwconv3x3_fwd_stream_ex(x, k, out, stat_partials, nullptr, B, H, W, C, flip, 0, stream);
  if (dtype != CRNN_F32 || !x || !k || !out) return CRNN_ERR_ARG;
  DwsGeom g = dws_geom(B, H, W, C, 4);
  if (!g.ok || (((uintptr_t)x | (uintptr_t)out | (uintptr_t)k) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 4 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DwsParams p;
  p.x = (const unsigned char*)x; p.k = k; p.out = (unsigned char*)out; p.partials = stat_partials; p.bnstate = nullptr;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.NS = g.NS; p.nwgb = g.nwgb; p.flip = flip; p.cols = g.cols; p.rowbytes = W * C * 4; p.wmaj = 0;
  p.nsplit = g.nsplit; p.cppw = g.cppw; p.xstride = g.xstride;
  p.pro_bn = nullptr; p.keep = nullptr; p.rate = 0.f;
  return dws_launch<false, true>(p, g, B, stream);
}
// The prologue form by storage type (dtype CRNN_BF16 = the entry points above; CRNN_F32, round 4 -- the parity mode): q, out fp32; rows of 18 KiB run as
// two channel ranges of 9 KiB (one workgroup each), four channels per lane, the keep bytes (still one per 8 elements) as nibbles.  out / stat_partials
// bit-identical to crnn_bn_act_pool_drop_ex(q -> x) + crnn_dwconv3x3_fwd_ex(x) on fp32 tensors (the statistics to the order of their partial sums).
extern "C" int crnn_dwconv_fwd_stream_pro_supported_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_fwd_stream_pro_supported(B, H, W, C);
  if (dtype != CRNN_F32) return CRNN_ERR_ARG;
  return dws_pro_ok(dws_geom(B, H, W, C, 4), B, H, W, C, 4) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" int crnn_dwconv_fwd_stream_rows_ex(int B, int H, int W, int C, int dtype) {
  if (dtype == CRNN_BF16) return crnn_dwconv_fwd_stream_rows(B, H, W, C);
  DwsGeom g = dws_geom(B, H, W, C, 4); return (dtype == CRNN_F32 && g.ok) ? B * g.nwgb : 0;
}
extern "C" int crnn_dwconv3x3_fwd_stream_pro_ex(const void* q, const float* pro_bnstate, float rate, const void* keep, const float* k, void* out,
                                                float* stat_partials, int B, int H, int W, int C, int dtype, hipStream_t stream) {
  if (dtype == CRNN_BF16) return crnn_dwconv3x3_fwd_stream_pro(q, pro_bnstate, rate, keep, k, out, stat_partials, B, H, W, C, stream);
  if (dtype != CRNN_F32 || !q || !pro_bnstate || !k || !out || rate < 0.f || rate >= 1.f || (rate > 0.f && !keep)) return CRNN_ERR_ARG;
  DwsGeom g = dws_geom(B, H, W, C, 4);
  if (!dws_pro_ok(g, B, H, W, C, 4) || (((uintptr_t)q | (uintptr_t)out | (uintptr_t)k | (uintptr_t)pro_bnstate) & 15) || ((uintptr_t)keep & 3)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C * 4 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  DwsParams p;
  p.x = (const unsigned char*)q; p.k = k; p.out = (unsigned char*)out; p.partials = stat_partials; p.bnstate = nullptr;
  p.H = H; p.W = W; p.C = C; p.HB = g.HB; p.NS = g.NS; p.nwgb = g.nwgb; p.flip = 0; p.cols = g.cols; p.rowbytes = W * C * 4; p.wmaj = 0;
  p.nsplit = g.nsplit; p.cppw = g.cppw;
  p.pro_bn = pro_bnstate; p.keep = (const unsigned char*)keep; p.rate = rate;
  return rate > 0.f ? dws_launch_pro<true, true>(p, g, B, stream) : dws_launch_pro<false, true>(p, g, B, stream);
}
extern "C" int crnn_dwconv3x3_fwd_stream(const void* x, const float* k, void* out, float* stat_partials, const float* bnstate, int B, int H, int W,
                                         int C, int flip, hipStream_t stream) {
  return crnn_dwconv3x3_fwd_stream_ex(x, k, out, stat_partials, bnstate, B, H, W, C, flip, 0, stream);
}
