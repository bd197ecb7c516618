// Persistent Bidirectional-LSTM recurrence (Keras 2.2.2 LSTMCell, utils.py:77-82): ONE launch per layer instead of
// one launch per timestep (rnn.hip), forward and BPTT.
//
// Decomposition.  h_t[b, :] depends on h_{t-1}[b, :] only -- batch rows never mix -- so the recurrence of a batch
// tile (BT = 16*MT rows) of one direction is an independent chain.  A chain is run by a CLUSTER of NS = u/16
// workgroups; workgroup `sl` owns hidden units j0 = 16*sl .. +15, i.e. the 4x16 gate columns {g*u + j0 + 0..15}:
//   * its slice of the recurrent weights (256 x 64 for u = 256) is loaded ONCE into registers as MFMA B fragments
//     (wave g holds gate g: 32 VGPRs bf16 / 64 VGPRs fp32) and stays there for all T steps;
//   * the cell state c (forward) / the cell-gradient carry dc (backward) of its (row, unit) pairs never leaves
//     registers;
//   * per step the only inter-workgroup traffic is the all-gather of the cluster's h_t (forward: BT x u) or dz_t
//     (backward: BT x 4u) slices through a write-once exchange buffer in device memory, staged through LDS as
//     the MFMA A operand of the next step.
// Hand-off protocol (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores): the
// exchange buffer is a ring of kRing = 4 step slots per chain, pre-filled with an all-ones sentinel (hipMemsetAsync 0xFF
// before the launch).  Producers store their slice of step s into slot s % 4 with 16-byte write-through (sc1) stores -- or, with
// the XCD-local map and after the cluster's census has shown that all its members share one XCD (rnn_exchange.h), plain stores:
// that XCD's L2 is then the meeting point (forward 131 -> 98 us, BPTT 198 -> 142 us per layer at B = 256, u = 256, bf16);
// consumers re-read the tile with 16-byte sc1 loads (L1-bypassing) until no dword equals the sentinel -- the data
// is its own ready flag (a valid |h| < 1 / a finite dz never has an all-ones bf16 pair or fp32 pattern, and a NaN
// produced by arithmetic is 0x7fc0..., not 0xffff...), so there is no flag, no fence and no drain on the critical
// path.  Slot reuse: once a workgroup has gathered the complete tile of step s-1, every member has finished reading
// step s-2 (a member publishes s-1 only after its gather of s-2 returned), so it re-poisons ITS slice of slot (s-2) % 4 =
// (s+2) % 4 after publishing step s, and drains its stores (s_waitcnt vmcnt(0)) before it publishes step s+1: whoever
// later sees its step s+1 data -- a precondition for polling slot (s+2) % 4 -- can no longer see the stale step s-2 there.
// Results do not depend on workgroup placement or dispatch order; a cluster's workgroups have consecutive
// block ids (or, with the XCD-local map, block ids congruent modulo 8: one XCD, one L2) and the whole grid is sized to be
// co-resident.  Every spin is bounded: on give-up the chain free-runs (wrong numbers, no hang) and says so twice in the
// first kStatusBytes of xbuf: the unsigned at byte 0 is a STICKY give-up counter that no launch ever resets (the caller
// zeroes it once when it allocates xbuf and compares it with the value it saw last -- Engine.check_rnn_status), the
// unsigned at byte 16 is the per-launch status word (all ones after a clean launch, bit 0 cleared on give-up).
//
// Numerics: bit-identical to the per-step kernels of rnn.hip in both modes -- the K split into four quarters, the
// k order inside a quarter, the ((q0+q1)+(q2+q3)) + x combination and the cell epilogue are the same; in the bf16
// modes the exchanged h / dz are the round-to-nearest-even bf16 values the step kernels formed while packing.
#include "common.h"
#include "rnn_cell.h"

#include "rnn_exchange.h"

#ifndef CRNN_RNN_EXP
#define CRNN_RNN_EXP 0   // experiment builds (scripts/lstm_ablate.py; wrong numbers): 1 = the per-step operands (x W / gates, c, dout) are constants instead of loads
#endif                   // -- what the loads cost and which re-schedulings did not recover it: profiles/r06_lstm_operand_ablation.txt

namespace {

struct FwdDir {
  const float* xw;   // [T][B][4u]  x*W + b
  const void* ut;    // U^T [4u][u], fp32 or bf16
  float* h; int ldh; // h(t,b,j) = h[(t*B+b)*ldh + j]
  float* c;          // [T][B][u]
  float* gates;      // [T][B][4u] activated i,f,g,o
};
struct BwdDir {
  const void* uw;    // U [u][4u], fp32 or bf16
  const float* c;    // [T][B][u]
  const float* gates;
  const float* dout; int ldo;
  float* dz;         // [T][B][4u]
  float* dbp;        // may be null: [ceil(B / 16)][4u] bias-gradient partials, the column sums of dz over t of every 16-row batch tile
};

// A workgroup has 4*UW waves = UW "unit groups" of 16 hidden units; wave w works for unit group w>>2 on gate (forward) /
// K quarter (backward) w&3.  The cluster of one chain has NSW = u/(16*UW) workgroups: larger workgroups mean fewer
// cluster members to wait for and 1/UW of the all-gather read traffic (the LDS tile is shared by the UW groups).

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <bool WBF, int MT, int U, int UW>
__global__ __launch_bounds__(256 * UW) void lstm_fwd_persist_kernel(FwdDir d0, FwdDir d1, int T, int B, int b_lo, int b_cnt, unsigned char* xbuf, int xmap) {
  typedef typename XE<WBF>::type E;
  constexpr int ES = sizeof(E), BT = 16 * MT, NSW = U / (16 * UW), NT = 256 * UW;
  constexpr int LDA = U + 16 / ES;                       // +16 bytes per row
  constexpr int NCH = BT * U * ES / 16;                  // 16-byte chunks of one exchange tile
  constexpr int KC = WBF ? U / 32 : U / 16;              // k-chunks (one bf16 MFMA = 32 k; four fp32 MFMAs = 16 k)
  constexpr int CPR = ES;                                // 16-byte chunks per published row (16 elements)
  __shared__ __attribute__((aligned(16))) E As[BT * LDA];
  __shared__ __attribute__((aligned(16))) float red[UW][MT][4][256];
  __shared__ __attribute__((aligned(16))) E hout[UW][BT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int gate = wave & 3, ug = wave >> 2;
  const int bid = cluster_block_id(blockIdx.x, NSW, xmap);
  const int sl = bid % NSW, cl = bid / NSW, dir = cl & 1, bt = cl >> 1;
  const int nbt = (b_cnt + BT - 1) / BT;
  const FwdDir d = dir ? d1 : d0;
  const int sg = sl * UW + ug;                           // this wave's unit group within the layer
  const int b0 = b_lo + bt * BT, b_end = b_lo + b_cnt, j0 = sg * 16;
  unsigned* status = reinterpret_cast<unsigned*>(xbuf);
  E* xdata = reinterpret_cast<E*>(xbuf + kStatusBytes);
  const long tile_elems = (long)BT * U;
  bool dead = false;
  const bool local = xmap && cluster_shares_xcd(xbuf, cl, sl, NSW, tid, status, dead);   // plain (L2-resident) exchange stores

  // this wave's gate columns j0 .. j0+15 of U as B fragments, resident for all T steps
  u32x4 breg[KC];
  {
    const long rowoff = ((long)gate * U + j0 + r) * U;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if constexpr (WBF) breg[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(d.ut) + rowoff + 32 * kc + 8 * q);
      else breg[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(d.ut) + rowoff + 16 * kc + 4 * q);
    }
  }
  const int tl = tid & 255, row = tl >> 4, col = tl & 15, j = j0 + col;
  float cprev[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) cprev[m] = 0.f;

#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
#define TRACE_STEP s
    RNN_TRACE(0);
    float xwv[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = b0 + 16 * m + row;
      const float* xw = d.xw + ((long)t * B + (b < b_end ? b : b_lo)) * 4 * U;
#pragma unroll
      for (int g = 0; g < 4; ++g) xwv[m][g] = (CRNN_RNN_EXP & 1) ? 0.01f * (float)(g + col) : xw[g * U + j];
    }
    if (s > 0) {
      const E* tile = xdata + (((long)dir * kRing + ((s - 1) & (kRing - 1))) * nbt + bt) * tile_elems;
      gather_tile<NCH, NT>(tile, tid, status, dead, [&](int idx, const u32x4& v) {
        const int e0 = idx * (16 / ES), sg2 = e0 / (BT * 16), rem = e0 % (BT * 16);
        *reinterpret_cast<u32x4*>(&As[(rem >> 4) * LDA + sg2 * 16 + (rem & 15)]) = v;
      });
      RNN_TRACE(1);
      __syncthreads();
      RNN_TRACE(2);
      f32x4 acc[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[m][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        constexpr int QC = KC / 4;                     // k-chunks per K quarter
        const int a = kc / QC;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (WBF) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(&As[(16 * m + r) * LDA + 32 * kc + 8 * q]);
            acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, breg[kc]), acc[m][a], 0, 0, 0);
          } else {
            const float4 av = *reinterpret_cast<const float4*>(&As[(16 * m + r) * LDA + 16 * kc + 4 * q]);
            const float4 bv = __builtin_bit_cast(float4, breg[kc]);
            acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[m][a], 0, 0, 0);
            acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[m][a], 0, 0, 0);
            acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[m][a], 0, 0, 0);
            acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[m][a], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e)   // C/D: row = 4q+e, col = r
          red[ug][m][gate][(q * 4 + e) * 16 + r] = (acc[m][0][e] + acc[m][1][e]) + (acc[m][2][e] + acc[m][3][e]);
      __syncthreads();
      RNN_TRACE(3);
    }
    LstmFwdOut o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = b0 + 16 * m + row;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = (s > 0 ? red[ug][m][g][tl] : 0.f) + xwv[m][g];
      o[m] = lstm_cell_fwd(z, cprev[m]);
      if (b >= b_end) o[m].hn = 0.f;
      cprev[m] = o[m].cn;
      if constexpr (WBF) hout[ug][(16 * m + row) * 16 + col] = (bf16_t)(pack2_bf16(o[m].hn, 0.f) & 0xffffu);
      else hout[ug][(16 * m + row) * 16 + col] = o[m].hn;
    }
    if (s + 1 < T) {
      // publish this unit group's h_t slice first (the other workgroups of the cluster wait for it): each wave stores the rows
      // it produced (4 per 16-row sub-tile), 16 B per lane, write-through
      E* tile = xdata + (((long)dir * kRing + (s & (kRing - 1))) * nbt + bt) * tile_elems + (long)sg * BT * 16;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(tile, BT * 16 * ES);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-poisoning of step s-1 has landed before step s goes out
      if (lane < MT * 4 * CPR) {
        const int part = lane % CPR, rl = (lane / CPR) % 4, m = lane / (4 * CPR);
        const int eoff = (16 * m + 4 * gate + rl) * 16 + part * (16 / ES);
        const u32x4 v = *reinterpret_cast<const u32x4*>(&hout[ug][eoff]);
        xstore(v, rs, eoff * ES, local);
      }
    }
    RNN_TRACE(4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {   // what the next layer / the backward pass read: off the critical path
      const int b = b0 + 16 * m + row;
      if (b < b_end) {
        float* gt = d.gates + ((long)t * B + b) * 4 * U;
        gt[j] = o[m].ig; gt[U + j] = o[m].fg; gt[2 * U + j] = o[m].gg; gt[3 * U + j] = o[m].og;
        d.c[((long)t * B + b) * U + j] = o[m].cn;
        d.h[((long)t * B + b) * d.ldh + j] = o[m].hn;
      }
    }
    if (s + 1 < T && lane < MT * 4 * CPR) {   // re-poison this wave's chunks of the slot step s+2 will use (last: nobody waits for it)
      E* stale = xdata + (((long)dir * kRing + ((s + 2) & (kRing - 1))) * nbt + bt) * tile_elems + (long)sg * BT * 16;
      const int part = lane % CPR, rl = (lane / CPR) % 4, m = lane / (4 * CPR);
      const int eoff = (16 * m + 4 * gate + rl) * 16 + part * (16 / ES);
      xstore((u32x4){kSentinel, kSentinel, kSentinel, kSentinel}, make_rsrc(stale, BT * 16 * ES), eoff * ES, local);
    }
#undef TRACE_STEP
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT): dh_{t}[b, j'] = sum_k dz_{t_next}[b, k] U[j', k] + dout_t[b, j'], then the gate gradients dz_t
// ---------------------------------------------------------------------------------------------------------------
template <bool WBF, int MT, int U, int UW>
__global__ __launch_bounds__(256 * UW) void lstm_bwd_persist_kernel(BwdDir d0, BwdDir d1, int T, int B, int b_lo, int b_cnt, unsigned char* xbuf, int xmap) {
  typedef typename XE<WBF>::type E;
  constexpr int ES = sizeof(E), BT = 16 * MT, NSW = U / (16 * UW), NT = 256 * UW, K = 4 * U;
  constexpr int LDA = K + 16 / ES;
  constexpr int NCH = BT * K * ES / 16;
  constexpr int KC = WBF ? U / 32 : U / 16;              // k-chunks of this wave's K quarter
  constexpr int CPR = 4 * ES;                            // 16-byte chunks per published row (4 x 16 elements)
  __shared__ __attribute__((aligned(16))) E As[BT * LDA];
  __shared__ __attribute__((aligned(16))) float red[UW][4][MT][256];
  __shared__ __attribute__((aligned(16))) E zout[UW][BT * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int kq = wave & 3, ug = wave >> 2;
  const int bid = cluster_block_id(blockIdx.x, NSW, xmap);
  const int sl = bid % NSW, cl = bid / NSW, dir = cl & 1, bt = cl >> 1;
  const int nbt = (b_cnt + BT - 1) / BT;
  const BwdDir d = dir ? d1 : d0;
  const int sg = sl * UW + ug;
  const int b0 = b_lo + bt * BT, b_end = b_lo + b_cnt, j0 = sg * 16;
  unsigned* status = reinterpret_cast<unsigned*>(xbuf);
  E* xdata = reinterpret_cast<E*>(xbuf + kStatusBytes);
  const long tile_elems = (long)BT * K;
  bool dead = false;
  const bool local = xmap && cluster_shares_xcd(xbuf, cl, sl, NSW, tid, status, dead);   // plain (L2-resident) exchange stores

  u32x4 breg[KC];   // U[j0 + r][kq*u + k]: this wave's K quarter (= gate kq) of the 16 output units
  {
    const long rowoff = (long)(j0 + r) * K + (long)kq * U;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if constexpr (WBF) breg[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(d.uw) + rowoff + 32 * kc + 8 * q);
      else breg[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(d.uw) + rowoff + 16 * kc + 4 * q);
    }
  }
  const int tl = tid & 255, row = tl >> 4, col = tl & 15, j = j0 + col;
  float dcin[MT], bs[MT][4];                          // bs: this thread's (row, unit) share of the bias gradient, summed over the steps
#pragma unroll
  for (int m = 0; m < MT; ++m) { dcin[m] = 0.f; bs[m][0] = bs[m][1] = bs[m][2] = bs[m][3] = 0.f; }

#pragma unroll 1
  for (int sb = 0; sb < T; ++sb) {
    const int sp = T - 1 - sb;                       // processing index of this time in the forward pass
    const int t = dir ? T - 1 - sp : sp;
    const int tnext = dir ? t - 1 : t + 1;           // processed after t in forward order (already back-propagated)
    const int tprev = dir ? t + 1 : t - 1;
    float gv[MT][4], ctv[MT], cpv[MT], dov[MT];      // epilogue operands, requested before the wait
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = b0 + 16 * m + row;
      const long bb = (b < b_end) ? b : b_lo;
      const float* gt = d.gates + ((long)t * B + bb) * K;
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[m][g] = (CRNN_RNN_EXP & 1) ? 0.3f + 0.01f * (float)g : gt[g * U + j];
      ctv[m] = (CRNN_RNN_EXP & 1) ? 0.2f : d.c[((long)t * B + bb) * U + j];
      cpv[m] = (CRNN_RNN_EXP & 1) ? 0.1f : (sp > 0) ? d.c[((long)tprev * B + bb) * U + j] : 0.f;
      dov[m] = (CRNN_RNN_EXP & 1) ? 0.01f * (float)col : d.dout[((long)t * B + bb) * d.ldo + j];
    }
    if (sb > 0) {
      const E* tile = xdata + (((long)dir * kRing + ((sb - 1) & (kRing - 1))) * nbt + bt) * tile_elems;
      gather_tile<NCH, NT>(tile, tid, status, dead, [&](int idx, const u32x4& v) {
        // tile layout [unit group][row][gate][16]  ->  A[row][gate*u + group*16 + jj]
        const int e0 = idx * (16 / ES), sg2 = e0 / (BT * 64), rem = e0 % (BT * 64);
        const int rw = rem >> 6, g = (rem >> 4) & 3, jj = rem & 15;
        *reinterpret_cast<u32x4*>(&As[rw * LDA + g * U + sg2 * 16 + jj]) = v;
      });
      __syncthreads();
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (WBF) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(&As[(16 * m + r) * LDA + kq * U + 32 * kc + 8 * q]);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, breg[kc]), acc[m], 0, 0, 0);
          } else {
            const float4 av = *reinterpret_cast<const float4*>(&As[(16 * m + r) * LDA + kq * U + 16 * kc + 4 * q]);
            const float4 bv = __builtin_bit_cast(float4, breg[kc]);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[m], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[ug][kq][m][(q * 4 + e) * 16 + r] = acc[m][e];
      __syncthreads();
    }
    LstmBwdOut o[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = b0 + 16 * m + row;
      const float part = (sb > 0) ? ((red[ug][0][m][tl] + red[ug][1][m][tl]) + (red[ug][2][m][tl] + red[ug][3][m][tl])) : 0.f;
      o[m] = lstm_cell_bwd(part + dov[m], gv[m][0], gv[m][1], gv[m][2], gv[m][3], ctv[m], cpv[m], dcin[m]);
      if (b >= b_end) { o[m].dz[0] = o[m].dz[1] = o[m].dz[2] = o[m].dz[3] = 0.f; o[m].dc = 0.f; }
      dcin[m] = o[m].dc;
#pragma unroll
      for (int g = 0; g < 4; ++g) bs[m][g] += o[m].dz[g];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (WBF) zout[ug][((16 * m + row) * 4 + g) * 16 + col] = (bf16_t)(pack2_bf16(o[m].dz[g], 0.f) & 0xffffu);
        else zout[ug][((16 * m + row) * 4 + g) * 16 + col] = o[m].dz[g];
      }
    }
    if (sb + 1 < T) {
      E* tile = xdata + (((long)dir * kRing + (sb & (kRing - 1))) * nbt + bt) * tile_elems + (long)sg * BT * 64;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(tile, BT * 64 * ES);
      constexpr int NPUB = MT * 4 * CPR;               // chunks this wave publishes (its 4 rows of every 16-row sub-tile)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c0 = 0; c0 < NPUB; c0 += 64) {
        const int id = c0 + lane;
        if (id < NPUB) {
          const int part = id % CPR, rl = (id / CPR) % 4, m = id / (4 * CPR);
          const int eoff = (16 * m + 4 * kq + rl) * 64 + part * (16 / ES);
          const u32x4 v = *reinterpret_cast<const u32x4*>(&zout[ug][eoff]);
          xstore(v, rs, eoff * ES, local);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = b0 + 16 * m + row;
      if (b < b_end) {
        float* dz = d.dz + ((long)t * B + b) * K;
        dz[j] = o[m].dz[0]; dz[U + j] = o[m].dz[1]; dz[2 * U + j] = o[m].dz[2]; dz[3 * U + j] = o[m].dz[3];
      }
    }
    if (sb + 1 < T) {   // re-poison this wave's chunks of the slot step sb+2 will use
      E* stale = xdata + (((long)dir * kRing + ((sb + 2) & (kRing - 1))) * nbt + bt) * tile_elems + (long)sg * BT * 64;
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(stale, BT * 64 * ES);
      constexpr int NPUB = MT * 4 * CPR;
#pragma unroll
      for (int c0 = 0; c0 < NPUB; c0 += 64) {
        const int id = c0 + lane;
        if (id < NPUB) {
          const int part = id % CPR, rl = (id / CPR) % 4, m = id / (4 * CPR);
          const int eoff = (16 * m + 4 * kq + rl) * 64 + part * (16 / ES);
          xstore((u32x4){kSentinel, kSentinel, kSentinel, kSentinel}, rp, eoff * ES, local);
        }
      }
    }
  }
  // bias gradient of the layer (Keras' recurrent bias: db = column sums of dz over time and batch): the rows of a 16-row tile are 4 lanes apart in 4 waves --
  // shuffles, then the waves through LDS in a fixed order; one partial row per 16-row batch tile (the stand-alone column reduction read dz again for it)
  if (d.dbp) {
    __syncthreads();                                 // (`red` is free: the last step's sums were consumed before the last barrier-free epilogue)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = bs[m][g];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 16) red[ug][kq][m][g * 16 + lane] = v;
      }
    __syncthreads();
    if (kq == 0) {
      const int g = lane >> 4, cc = lane & 15;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (b0 + 16 * m < b_end)
          d.dbp[(long)((b0 + 16 * m) >> 4) * K + g * U + j0 + cc] =
              (red[ug][0][m][g * 16 + cc] + red[ug][1][m][g * 16 + cc]) + (red[ug][2][m][g * 16 + cc] + red[ug][3][m][g * 16 + cc]);
    }
  }
}

constexpr size_t lds_fwd(int U, int MT, int UW, int ES) {
  return (size_t)16 * MT * (U + 16 / ES) * ES + (size_t)UW * MT * 4 * 256 * 4 + (size_t)UW * 16 * MT * 16 * ES;
}
constexpr size_t lds_bwd(int U, int MT, int UW, int ES) {
  return (size_t)16 * MT * (4 * U + 16 / ES) * ES + (size_t)UW * MT * 4 * 256 * 4 + (size_t)UW * 16 * MT * 64 * ES;
}
// instantiated variants: the unit groups must tile u, the LDS must fit, and 1024-thread workgroups must fit 128 VGPRs
constexpr bool fwd_ok(bool wbf, int MT, int U, int UW) {
  return U % (16 * UW) == 0 && lds_fwd(U, MT, UW, wbf ? 2 : 4) <= 150 * 1024 && !(UW == 4 && !wbf && U >= 256 && MT == 2) && !(UW == 4 && U >= 512 && MT == 2);
}
constexpr bool bwd_ok(bool wbf, int MT, int U, int UW) {
  return U % (16 * UW) == 0 && lds_bwd(U, MT, UW, wbf ? 2 : 4) <= 150 * 1024 && !(UW == 4 && (!wbf ? U >= 256 : (U >= 512 || (U >= 256 && MT == 2))));
}

}  // namespace

// Bytes of the exchange buffer `xbuf` the persistent recurrences need for (T, B, u): a ring of kRing step slots per
// direction, sized for the backward (4u values per row and step); the forward uses a quarter of it.
extern "C" size_t crnn_lstm_persist_xbuf_bytes(int T, int B, int u, int dt_u) {
  const size_t es = (dt_u == CRNN_BF16) ? 2 : 4;
  const size_t rows = (size_t)cdiv(B, 32) * 32;
  (void)T;
  return kStatusBytes + 2 * (size_t)kRing * rows * 4 * (size_t)u * es;
}

// 0 when (u, dt_u) has a persistent kernel; CRNN_ERR_UNSUPPORTED otherwise (callers fall back to the per-step kernels)
extern "C" int crnn_lstm_persist_supported(int u, int dt_u) {
  if (dt_u == CRNN_BF16) return (u == 128 || u == 256 || u == 512) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
  return (u == 64 || u == 128 || u == 256) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}

namespace {
template <bool WBF, int MT, int U, int UW>
int launch_fwd_v(const FwdDir& a, const FwdDir& b, int T, int B, void* xbuf, size_t xbuf_bytes, int xreq, hipStream_t stream) {
  if constexpr (!fwd_ok(WBF, MT, U, UW)) {
    return CRNN_ERR_UNSUPPORTED;
  } else {
    constexpr int ES = WBF ? 2 : 4, NSW = U / (16 * UW), BT = 16 * MT;
    const Chunking ck = chunking(T, B, U, MT, UW, ES, lds_fwd(U, MT, UW, ES), U, (const void*)lstm_fwd_persist_kernel<WBF, MT, U, UW>);
    for (int lo = 0; lo < B; lo += ck.rows_per_launch) {
      const int cnt = (B - lo < ck.rows_per_launch) ? B - lo : ck.rows_per_launch;
      CRNN_TRY(prep_xbuf(xbuf, xbuf_bytes, ck.xdata_bytes, stream));     // every slot is written once per launch: poison first
      const int ncl = 2 * cdiv(cnt, BT);
      // xreq & 2 (CRNN_RNN_DEBUG_DROP_MEMBER, tests only): the last workgroup is not launched -- its cluster waits, gives up and says so
      hipLaunchKernelGGL((lstm_fwd_persist_kernel<WBF, MT, U, UW>), dim3(ncl * NSW - ((xreq & 2) ? 1 : 0)), dim3(256 * UW), 0, stream, a, b, T, B, lo, cnt,
                         (unsigned char*)xbuf, ((xreq & 1) && ncl % 8 == 0) ? 1 : 0);
    }
    return CRNN_OK;
  }
}
template <bool WBF, int MT, int U, int UW>
int launch_bwd_v(const BwdDir& a, const BwdDir& b, int T, int B, void* xbuf, size_t xbuf_bytes, int xreq, hipStream_t stream) {
  if constexpr (!bwd_ok(WBF, MT, U, UW)) {
    return CRNN_ERR_UNSUPPORTED;
  } else {
    constexpr int ES = WBF ? 2 : 4, NSW = U / (16 * UW), BT = 16 * MT;
    const Chunking ck = chunking(T, B, U, MT, UW, ES, lds_bwd(U, MT, UW, ES), 4 * U, (const void*)lstm_bwd_persist_kernel<WBF, MT, U, UW>);
    for (int lo = 0; lo < B; lo += ck.rows_per_launch) {
      const int cnt = (B - lo < ck.rows_per_launch) ? B - lo : ck.rows_per_launch;
      CRNN_TRY(prep_xbuf(xbuf, xbuf_bytes, ck.xdata_bytes, stream));
      const int ncl = 2 * cdiv(cnt, BT);
      hipLaunchKernelGGL((lstm_bwd_persist_kernel<WBF, MT, U, UW>), dim3(ncl * NSW), dim3(256 * UW), 0, stream, a, b, T, B, lo, cnt,
                         (unsigned char*)xbuf, ((xreq & 1) && ncl % 8 == 0) ? 1 : 0);
    }
    return CRNN_OK;
  }
}
#define DISPATCH_MT_UW(FN, WBF, U, ...)                                              \
  (mt == 2 ? (uw == 4 ? FN<WBF, 2, U, 4>(__VA_ARGS__) : uw == 2 ? FN<WBF, 2, U, 2>(__VA_ARGS__) : FN<WBF, 2, U, 1>(__VA_ARGS__)) \
           : (uw == 4 ? FN<WBF, 1, U, 4>(__VA_ARGS__) : uw == 2 ? FN<WBF, 1, U, 2>(__VA_ARGS__) : FN<WBF, 1, U, 1>(__VA_ARGS__)))

// (mt, uw) = (batch rows per workgroup / 16, 16-unit groups per workgroup): explicit requests are tried first, then the
// automatic choice, then smaller workgroups
template <typename Try>
int with_fallback(int B, int u, int mt_req, int uw_req, Try attempt) {
  int mt = (mt_req == 1 || mt_req == 2) ? mt_req : 1;
  // default: 512-thread workgroups, two unit groups each (measured at u = 256, B = 256: 139 / 187 us forward / backward against
  // 164 / 267 with one group and 196 / 233 with four -- 16-wave barriers cost more than the smaller cluster saves)
  int uw = (uw_req == 1 || uw_req == 2 || uw_req == 4) ? uw_req : 2;
  while (uw > 1 && u % (16 * uw)) uw >>= 1;
  for (;;) {
    const int rc = attempt(mt, uw);
    if (rc != CRNN_ERR_UNSUPPORTED) return rc;
    if (uw > 1) uw >>= 1;
    else if (mt > 1) { mt = 1; uw = (uw_req == 1 || uw_req == 2 || uw_req == 4) ? uw_req : 2; }
    else return rc;
  }
}
}  // namespace

// Zero the sticky give-up counter at the head of an exchange buffer (once after allocation; see crnn_lstm_fwd_persist).
extern "C" int crnn_rnn_status_reset(void* xbuf, hipStream_t stream) {
  if (!xbuf || ((uintptr_t)xbuf & 15)) return CRNN_ERR_ARG;
  hipError_t e = hipMemsetAsync(xbuf, 0, 4, stream);
  return e == hipSuccess ? CRNN_OK : (int)e;
}

// Forward recurrence of one Bidirectional(LSTM) layer in ONE launch.  Arguments as crnn_lstm_fwd_ex; `xbuf` is
// caller-owned scratch of crnn_lstm_persist_xbuf_bytes() bytes (16-byte aligned, its first 4 bytes zeroed ONCE by the caller after
// allocation).  Status words: the unsigned at byte 0 is a sticky counter of give-ups since the caller zeroed it (no launch resets it);
// the unsigned at byte 16 is the per-launch status, 0xFFFFFFFF after a clean launch, anything else means a bounded wait gave up
// (results invalid).  crnn_rnn_status_reset(xbuf) zeroes the counter for callers that do not allocate with a zero fill (the earlier
// contract filled xbuf[0] with 0xFFFFFFFF per launch: such a buffer must be reset once before it is used with this version).
// mt: batch rows per workgroup / 16 (1 | 2), uw: 16-unit groups per workgroup (1 | 2 | 4); 0 = automatic.  uw_req | CRNN_RNN_XCD_LOCAL
// (0x100) asks for the XCD-local workgroup -> cluster map (same results).
extern "C" int crnn_lstm_fwd_persist(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1,
                                     int ldh, float* c0, float* c1, float* g0, float* g1, int T, int B, int u, int dt_u,
                                     void* xbuf, size_t xbuf_bytes, int mt_req, int uw_req, hipStream_t stream) {
  CRNN_TRY(crnn_lstm_persist_supported(u, dt_u));
  if (T < 1 || B < 1 || (((uintptr_t)ut0 | (uintptr_t)ut1) & 15)) return CRNN_ERR_ARG;
  FwdDir a{xw0, ut0, h0, ldh, c0, g0}, b{xw1, ut1, h1, ldh, c1, g1};
  const int xreq = ((uw_req & CRNN_RNN_XCD_LOCAL) ? 1 : 0) | ((uw_req & CRNN_RNN_DEBUG_DROP_MEMBER) ? 2 : 0); uw_req &= 0xff;
  const int rc = with_fallback(B, u, mt_req, uw_req, [&](int mt, int uw) {
    if (dt_u == CRNN_BF16) {
      if (u == 128) return DISPATCH_MT_UW(launch_fwd_v, true, 128, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
      if (u == 256) return DISPATCH_MT_UW(launch_fwd_v, true, 256, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
      return DISPATCH_MT_UW(launch_fwd_v, true, 512, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    }
    if (u == 64) return DISPATCH_MT_UW(launch_fwd_v, false, 64, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    if (u == 128) return DISPATCH_MT_UW(launch_fwd_v, false, 128, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    return DISPATCH_MT_UW(launch_fwd_v, false, 256, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  });
  CRNN_TRY(rc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// BPTT of one Bidirectional(LSTM) layer in ONE launch: fills dz[d] [T][B][4u] from dout[d].  Arguments as
// crnn_lstm_bwd_ex without the dc scratch (the cell-gradient carry stays in registers).
// crnn_lstm_bwd_persist_db: the same launch also leaves the layer's bias-gradient partials, db_partials0 / 1 [crnn_rnn_db_rows(B)][4u] = column sums of dz0 / dz1 over
// time for every 16-row batch tile (finish with crnn_partials_sum over the rows) -- the stand-alone column reduction reads dz (54 MB per direction) once more.
extern "C" int crnn_rnn_db_rows(int B) { return cdiv(B, 16); }
extern "C" int crnn_lstm_bwd_persist_db(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0,
                                        const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0,
                                        float* db_partials1, int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes, int mt_req, int uw_req,
                                        hipStream_t stream);
extern "C" int crnn_lstm_bwd_persist(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0,
                                     const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1,
                                     int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes, int mt_req, int uw_req, hipStream_t stream) {
  return crnn_lstm_bwd_persist_db(u0, u1, c0, c1, g0, g1, dout0, dout1, ldo, dz0, dz1, nullptr, nullptr, T, B, u, dt_u, xbuf, xbuf_bytes, mt_req, uw_req, stream);
}
extern "C" int crnn_lstm_bwd_persist_db(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0,
                                        const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0,
                                        float* db_partials1, int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes, int mt_req, int uw_req,
                                        hipStream_t stream) {
  CRNN_TRY(crnn_lstm_persist_supported(u, dt_u));
  if (T < 1 || B < 1 || (((uintptr_t)u0 | (uintptr_t)u1) & 15) || (!db_partials0) != (!db_partials1)) return CRNN_ERR_ARG;
  BwdDir a{u0, c0, g0, dout0, ldo, dz0, db_partials0}, b{u1, c1, g1, dout1, ldo, dz1, db_partials1};
  const int xreq = (uw_req & CRNN_RNN_XCD_LOCAL) ? 1 : 0; uw_req &= 0xff;
  const int rc = with_fallback(B, u, mt_req, uw_req, [&](int mt, int uw) {
    if (dt_u == CRNN_BF16) {
      if (u == 128) return DISPATCH_MT_UW(launch_bwd_v, true, 128, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
      if (u == 256) return DISPATCH_MT_UW(launch_bwd_v, true, 256, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
      return DISPATCH_MT_UW(launch_bwd_v, true, 512, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    }
    if (u == 64) return DISPATCH_MT_UW(launch_bwd_v, false, 64, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    if (u == 128) return DISPATCH_MT_UW(launch_bwd_v, false, 128, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
    return DISPATCH_MT_UW(launch_bwd_v, false, 256, a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  });
  CRNN_TRY(rc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
