// Persistent Bidirectional-GRU recurrence (Keras 2.2.2 GRUCell, reset_after=False, gate order z,r,h; utils.py:80-82 -- the cell
// the reference's train.py really builds, train.py:119): ONE launch per layer and pass instead of 2 T dependent launches (rnn.hip).
//
// Same decomposition as the persistent LSTM (rnn_persist.hip): the chain of one 16-row batch tile of one direction is run by a
// CLUSTER of u/(16 UW) workgroups; a workgroup owns 16 UW hidden units, keeps its slices of the recurrent weights in registers for
// all T steps (wave (ug, kq) = unit group ug, K quarter kq -- the K split, the k order and the ((q0+q1)+(q2+q3)) + x combination of
// the step kernels), and the hidden state / gradient carry of its (row, unit) pairs never leaves registers.
//
// The GRU needs TWO all-gathers per step where the LSTM needs one: the candidate's recurrent product takes r * h_prev of ALL units
//   forward   gather h_{t-1}            -> z, r of the own units -> publish r*h_prev  ->  gather r*h_prev -> hh, h_t -> publish h_t
//   backward  gather [dz|dr]_{t_next}   -> dh_t, dz_t, dhh_t     -> publish dhh_t     ->  gather dhh_t    -> dr_t, carry -> publish [dz|dr]_t
// Both go through ONE ring of kRing = 4 slots indexed by the linear exchange number e (forward: e = 2s for r*h, 2s+1 for h_s;
// backward: e = 2sb for dhh, 2sb+1 for [dz|dr]); a workgroup publishes e only after it gathered e-1, which is all the slot-reuse
// argument of rnn_persist.hip needs: after publishing e it re-poisons ITS slice of slot (e+2) % 4 (last used by e-2, which every
// member has finished reading) and drains its stores before publishing e+1.  Even and odd slots keep their tile shape.
// Sentinel: valid r*h, h (|.| < 1) and finite gradients never have an all-ones bf16 pair / fp32 pattern.
//
// Numerics: bit-identical to gru_*_kernel of rnn.hip in both modes (shared cell arithmetic rnn_cell.h, contraction off).
#include "common.h"
#include "rnn_cell.h"
#include "rnn_exchange.h"

namespace {

struct GruFwdDir {
  const float* xw;   // [T][B][3u]  x*W + b
  const void* ut;    // U^T [3u][u], fp32 or bf16
  float* h; int ldh; // h(t,b,j) = h[(t*B+b)*ldh + j]
  float* gates;      // [T][B][3u] z, r, hh
  float* rh;         // [T][B][u]  r * h_prev
};
struct GruBwdDir {
  const void* uw;    // U [u][3u], fp32 or bf16
  const float* h; int ldh;
  const float* gates;
  const float* dout; int ldo;
  float* dz;         // [T][B][3u]
  float* dbp;        // may be null: [ceil(B / 16)][3u] bias-gradient partials, the column sums of dz over t of every 16-row batch tile
};

template <bool WBF> __device__ __forceinline__ typename XE<WBF>::type to_e(float v);
template <> __device__ __forceinline__ bf16_t to_e<true>(float v) { return (bf16_t)(pack2_bf16(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ float to_e<false>(float v) { return v; }

// one K-quarter chain of a 16x16 tile: acc += A[r][k0 + ...] * Bfrag over NKC k-chunks, ascending (the step kernels' order)
template <bool WBF, int NKC, typename E>
__device__ __forceinline__ f32x4 quarter_chain(const E* As, int lda, int k0, const u32x4 (&b)[NKC], int r, int q) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    if constexpr (WBF) {
      const u32x4 av = *reinterpret_cast<const u32x4*>(&As[r * lda + k0 + 32 * kc + 8 * q]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, b[kc]), acc, 0, 0, 0);
    } else {
      const float4 av = *reinterpret_cast<const float4*>(&As[r * lda + k0 + 16 * kc + 4 * q]);
      const float4 bv = __builtin_bit_cast(float4, b[kc]);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
  }
  return acc;
}
// NKC fragments of weight row `wrow` (k contiguous) starting at column k0: lane (r, q) holds k = k0 + chunk + (8|4) q ...
template <bool WBF, int NKC>
__device__ __forceinline__ void load_frags(const void* w, long row_elems_off, int k0, int q, u32x4 (&b)[NKC]) {
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    if constexpr (WBF) b[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(w) + row_elems_off + k0 + 32 * kc + 8 * q);
    else b[kc] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(w) + row_elems_off + k0 + 16 * kc + 4 * q);
  }
}
// a wave publishes the rows it produced (4 of the 16-row tile: rows 4 kq .. 4 kq + 3), RE elements each, from its LDS staging
// area to its slice of an exchange tile: 16 bytes per lane (write-through, or plain inside a verified one-XCD cluster)
template <int RE, typename E>
__device__ __forceinline__ void publish_rows(const E* stage, E* slice, int kq, int lane, bool local) {
  constexpr int ES = sizeof(E), CPR = RE * ES / 16;                 // 16-byte chunks per row
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(slice, 16 * RE * ES);
  if (lane < 4 * CPR) {
    const int part = lane % CPR, rl = lane / CPR;
    const int eoff = (4 * kq + rl) * RE + part * (16 / ES);
    const u32x4 v = *reinterpret_cast<const u32x4*>(&stage[eoff]);
    xstore(v, rs, eoff * ES, local);
  }
}
template <int RE, typename E>
__device__ __forceinline__ void poison_rows(E* slice, int kq, int lane, bool local) {
  constexpr int ES = sizeof(E), CPR = RE * ES / 16;
  if (lane < 4 * CPR) {
    const int part = lane % CPR, rl = lane / CPR;
    const int eoff = (4 * kq + rl) * RE + part * (16 / ES);
    xstore((u32x4){kSentinel, kSentinel, kSentinel, kSentinel}, make_rsrc(slice, 16 * RE * ES), eoff * ES, local);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <bool WBF, int U, int UW>
__global__ __launch_bounds__(256 * UW) void gru_fwd_persist_kernel(GruFwdDir d0, GruFwdDir d1, int T, int B, int b_lo, int b_cnt, unsigned char* xbuf, int xmap) {
  typedef typename XE<WBF>::type E;
  constexpr int ES = sizeof(E), BT = 16, NSW = U / (16 * UW), NT = 256 * UW;
  constexpr int LDA = U + 16 / ES;                       // +16 bytes per row
  constexpr int NCH = BT * U * ES / 16;                  // 16-byte chunks of one exchange tile (h or r*h: BT x U)
  constexpr int KQ = WBF ? U / 128 : U / 64;             // k-chunks per K quarter (one bf16 MFMA = 32 k; four fp32 MFMAs = 16 k)
  __shared__ __attribute__((aligned(16))) E As[BT * LDA];
  __shared__ __attribute__((aligned(16))) float red[UW][4][2][256];
  __shared__ __attribute__((aligned(16))) E pub[UW][BT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int kq = wave & 3, ug = wave >> 2;
  const int bid = cluster_block_id(blockIdx.x, NSW, xmap);
  const int sl = bid % NSW, cl = bid / NSW, dir = cl & 1, bt = cl >> 1;
  const int nbt = (b_cnt + BT - 1) / BT;
  const GruFwdDir d = dir ? d1 : d0;
  const int sg = sl * UW + ug;                           // this wave's unit group within the layer
  const int b0 = b_lo + bt * BT, b_end = b_lo + b_cnt, j0 = sg * 16;
  unsigned* status = reinterpret_cast<unsigned*>(xbuf);
  E* xdata = reinterpret_cast<E*>(xbuf + kStatusBytes);
  const long tile_elems = (long)BT * U;
  bool dead = false;
  const bool local = xmap && cluster_shares_xcd(xbuf, cl, sl, NSW, tid, status, dead);   // plain (L2-resident) exchange stores
  auto slot_tile = [&](int e) { return xdata + (((long)dir * kRing + (e & (kRing - 1))) * nbt + bt) * tile_elems; };

  // this wave's K quarter of the z, r and candidate columns j0 .. j0+15 of U (rows of U^T), resident for all T steps
  u32x4 bz[KQ], br[KQ], bh[KQ];
  load_frags<WBF, KQ>(d.ut, ((long)0 * U + j0 + r) * U, kq * (U / 4), q, bz);
  load_frags<WBF, KQ>(d.ut, ((long)1 * U + j0 + r) * U, kq * (U / 4), q, br);
  load_frags<WBF, KQ>(d.ut, ((long)2 * U + j0 + r) * U, kq * (U / 4), q, bh);

  const int tl = tid & 255, row = tl >> 4, col = tl & 15, j = j0 + col;
  const int b = b0 + row;
  const bool live = b < b_end;
  float hprev = 0.f;
  auto gather_to_As = [&](int e) {
    gather_tile<NCH, NT>(slot_tile(e), tid, status, dead, [&](int idx, const u32x4& v) {
      const int e0 = idx * (16 / ES), sg2 = e0 / (BT * 16), rem = e0 % (BT * 16);
      *reinterpret_cast<u32x4*>(&As[(rem >> 4) * LDA + sg2 * 16 + (rem & 15)]) = v;
    });
  };

#pragma unroll 1
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const float* xw = d.xw + ((long)t * B + (live ? b : b_lo)) * 3 * U;
    const float xz = xw[j], xr = xw[U + j], xh = xw[2 * U + j];     // requested before the wait
    float sz = 0.f, sr = 0.f, sh = 0.f;
    if (s > 0) {
      gather_to_As(2 * s - 1);                                       // h_{s-1} of the whole cluster
      __syncthreads();
      const f32x4 az = quarter_chain<WBF, KQ>(As, LDA, kq * (U / 4), bz, r, q);
      const f32x4 ar = quarter_chain<WBF, KQ>(As, LDA, kq * (U / 4), br, r, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // C/D: row = 4q+e, col = r
        red[ug][kq][0][(q * 4 + e) * 16 + r] = az[e];
        red[ug][kq][1][(q * 4 + e) * 16 + r] = ar[e];
      }
      __syncthreads();
      sz = (red[ug][0][0][tl] + red[ug][1][0][tl]) + (red[ug][2][0][tl] + red[ug][3][0][tl]);
      sr = (red[ug][0][1][tl] + red[ug][1][1][tl]) + (red[ug][2][1][tl] + red[ug][3][1][tl]);
    }
    GruZR o = gru_cell_zr(sz + xz, sr + xr, hprev);
    if (!live) o.rh = 0.f;
    if (s > 0) {
      // publish r * h_prev of this unit group (exchange 2s): the candidate product of every member waits for it
      pub[ug][row * 16 + col] = to_e<WBF>(o.rh);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the re-poisoning after the previous exchange has landed
      publish_rows<16>(pub[ug], slot_tile(2 * s) + (long)sg * BT * 16, kq, lane, local);
    }
    if (live) {                                                      // what the backward pass reads: off the critical path
      float* gt = d.gates + ((long)t * B + b) * 3 * U;
      gt[j] = o.zg; gt[U + j] = o.rg;
      d.rh[((long)t * B + b) * U + j] = o.rh;
    }
    if (s > 0) {
      poison_rows<16>(slot_tile(2 * s + 2) + (long)sg * BT * 16, kq, lane, local);
      gather_to_As(2 * s);                                           // r * h_prev of the whole cluster
      __syncthreads();
      const f32x4 ah = quarter_chain<WBF, KQ>(As, LDA, kq * (U / 4), bh, r, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) red[ug][kq][0][(q * 4 + e) * 16 + r] = ah[e];
      __syncthreads();
      sh = (red[ug][0][0][tl] + red[ug][1][0][tl]) + (red[ug][2][0][tl] + red[ug][3][0][tl]);
    }
    GruH g = gru_cell_h(sh + xh, o.zg, hprev);
    if (!live) g.hn = 0.f;
    hprev = g.hn;
    if (s + 1 < T) {
      pub[ug][row * 16 + col] = to_e<WBF>(g.hn);                     // exchange 2s+1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_rows<16>(pub[ug], slot_tile(2 * s + 1) + (long)sg * BT * 16, kq, lane, local);
    }
    if (live) {
      d.gates[((long)t * B + b) * 3 * U + 2 * U + j] = g.hh;
      d.h[((long)t * B + b) * d.ldh + j] = g.hn;
    }
    if (s + 1 < T) poison_rows<16>(slot_tile(2 * s + 3) + (long)sg * BT * 16, kq, lane, local);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT): dh_t = dout_t + [dz|dr]_{t_next} U[:, 0:2u]^T + carry ;  d(r h)_t = dhh_t U[:, 2u:3u]^T
// ---------------------------------------------------------------------------------------------------------------
template <bool WBF, int U, int UW>
__global__ __launch_bounds__(256 * UW) void gru_bwd_persist_kernel(GruBwdDir d0, GruBwdDir d1, int T, int B, int b_lo, int b_cnt, unsigned char* xbuf, int xmap) {
  typedef typename XE<WBF>::type E;
  constexpr int ES = sizeof(E), BT = 16, NSW = U / (16 * UW), NT = 256 * UW, K2 = 2 * U, G = 3 * U;
  constexpr int LDA = K2 + 16 / ES;
  constexpr int NCH_ZR = BT * K2 * ES / 16, NCH_H = BT * U * ES / 16;
  constexpr int KQB = WBF ? U / 64 : U / 32;             // k-chunks per quarter of K = 2u
  constexpr int KQA = WBF ? U / 128 : U / 64;            // k-chunks per quarter of K = u
  __shared__ __attribute__((aligned(16))) E As[BT * LDA];
  __shared__ __attribute__((aligned(16))) float red[UW][4][256];
  __shared__ __attribute__((aligned(16))) E pub[UW][BT * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int kq = wave & 3, ug = wave >> 2;
  const int bid = cluster_block_id(blockIdx.x, NSW, xmap);
  const int sl = bid % NSW, cl = bid / NSW, dir = cl & 1, bt = cl >> 1;
  const int nbt = (b_cnt + BT - 1) / BT;
  const GruBwdDir d = dir ? d1 : d0;
  const int sg = sl * UW + ug;
  const int b0 = b_lo + bt * BT, b_end = b_lo + b_cnt, j0 = sg * 16;
  unsigned* status = reinterpret_cast<unsigned*>(xbuf);
  E* xdata = reinterpret_cast<E*>(xbuf + kStatusBytes);
  const long tile_elems = (long)BT * K2;                 // slot stride (the dhh tiles use half of it)
  bool dead = false;
  const bool local = xmap && cluster_shares_xcd(xbuf, cl, sl, NSW, tid, status, dead);   // plain (L2-resident) exchange stores
  auto slot_tile = [&](int e) { return xdata + (((long)dir * kRing + (e & (kRing - 1))) * nbt + bt) * tile_elems; };

  // U[j0 + r][.]: this wave's quarter of the z|r columns (K = 2u) and of the candidate columns (K = u)
  u32x4 bb[KQB], ba[KQA];
  load_frags<WBF, KQB>(d.uw, (long)(j0 + r) * G, kq * (U / 2), q, bb);
  load_frags<WBF, KQA>(d.uw, (long)(j0 + r) * G + 2 * U, kq * (U / 4), q, ba);

  const int tl = tid & 255, row = tl >> 4, col = tl & 15, j = j0 + col;
  const int b = b0 + row;
  const bool live = b < b_end;
  const long bbx = live ? b : b_lo;
  float dhp = 0.f, bsz = 0.f, bsr = 0.f, bsh = 0.f;       // bs*: this thread's (row, unit) share of the bias gradient (z | r | candidate), summed over the steps

#pragma unroll 1
  for (int sb = 0; sb < T; ++sb) {
    const int sp = T - 1 - sb;                       // processing index of this time in the forward pass
    const int t = dir ? T - 1 - sp : sp;
    const int tprev = dir ? t + 1 : t - 1;
    const float* gt = d.gates + ((long)t * B + bbx) * G;              // epilogue operands, requested before the wait
    const float zg = gt[j], rg = gt[U + j], hh = gt[2 * U + j];
    const float hprev = (sp > 0) ? d.h[((long)tprev * B + bbx) * d.ldh + j] : 0.f;
    float dh = d.dout[((long)t * B + bbx) * d.ldo + j];
    if (sb > 0) {
      gather_tile<NCH_ZR, NT>(slot_tile(2 * sb - 1), tid, status, dead, [&](int idx, const u32x4& v) {
        // tile layout [unit group][row][gate z|r][16]  ->  A[row][gate*u + group*16 + jj]
        const int e0 = idx * (16 / ES), sg2 = e0 / (BT * 32), rem = e0 % (BT * 32);
        const int rw = rem >> 5, g = (rem >> 4) & 1, jj = rem & 15;
        *reinterpret_cast<u32x4*>(&As[rw * LDA + g * U + sg2 * 16 + jj]) = v;
      });
      __syncthreads();
      const f32x4 acc = quarter_chain<WBF, KQB>(As, LDA, kq * (U / 2), bb, r, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) red[ug][kq][(q * 4 + e) * 16 + r] = acc[e];
      __syncthreads();
      dh += ((red[ug][0][tl] + red[ug][1][tl]) + (red[ug][2][tl] + red[ug][3][tl])) + dhp;
    }
    GruBwdB ob = gru_cell_bwd_b(dh, zg, hh, hprev);
    if (!live) { ob.dzz = 0.f; ob.dhh = 0.f; }
    bsz += ob.dzz; bsh += ob.dhh;
    // publish dhh_t of this unit group (exchange 2sb)
    pub[ug][row * 16 + col] = to_e<WBF>(ob.dhh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_rows<16>(pub[ug], slot_tile(2 * sb) + (long)sg * BT * 16, kq, lane, local);
    if (live) {
      float* dz = d.dz + ((long)t * B + b) * G;
      dz[j] = ob.dzz; dz[2 * U + j] = ob.dhh;
    }
    poison_rows<16>(slot_tile(2 * sb + 2) + (long)sg * BT * 16, kq, lane, local);
    gather_tile<NCH_H, NT>(slot_tile(2 * sb), tid, status, dead, [&](int idx, const u32x4& v) {
      const int e0 = idx * (16 / ES), sg2 = e0 / (BT * 16), rem = e0 % (BT * 16);
      *reinterpret_cast<u32x4*>(&As[(rem >> 4) * LDA + sg2 * 16 + (rem & 15)]) = v;
    });
    __syncthreads();
    {
      const f32x4 acc = quarter_chain<WBF, KQA>(As, LDA, kq * (U / 4), ba, r, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) red[ug][kq][(q * 4 + e) * 16 + r] = acc[e];
    }
    __syncthreads();
    const float drh = (red[ug][0][tl] + red[ug][1][tl]) + (red[ug][2][tl] + red[ug][3][tl]);
    GruBwdA oa = gru_cell_bwd_a(drh, dh, zg, rg, hprev);
    if (!live) { oa.dzr = 0.f; oa.dhp = 0.f; }
    bsr += oa.dzr;
    dhp = oa.dhp;
    if (sb + 1 < T) {
      pub[ug][(row * 2 + 0) * 16 + col] = to_e<WBF>(ob.dzz);         // exchange 2sb+1: [dz | dr]_t
      pub[ug][(row * 2 + 1) * 16 + col] = to_e<WBF>(oa.dzr);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_rows<32>(pub[ug], slot_tile(2 * sb + 1) + (long)sg * BT * 32, kq, lane, local);
    }
    if (live) d.dz[((long)t * B + b) * G + U + j] = oa.dzr;
    if (sb + 1 < T) poison_rows<32>(slot_tile(2 * sb + 3) + (long)sg * BT * 32, kq, lane, local);
  }
  // bias gradient of the layer (db = column sums of dz over time and batch), as in the LSTM kernel: the 16 rows of the tile are 4 lanes apart in 4 waves
  if (d.dbp) {
    __syncthreads();
    float v[3] = {bsz, bsr, bsh};
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      v[g] += __shfl_xor(v[g], 16, 64); v[g] += __shfl_xor(v[g], 32, 64);
      if (lane < 16) red[ug][kq][g * 16 + lane] = v[g];
    }
    __syncthreads();
    if (kq == 0 && lane < 48 && b0 < b_end) {
      const int g = lane >> 4, cc = lane & 15;
      d.dbp[(long)(b0 >> 4) * G + g * U + j0 + cc] = (red[ug][0][g * 16 + cc] + red[ug][1][g * 16 + cc]) + (red[ug][2][g * 16 + cc] + red[ug][3][g * 16 + cc]);
    }
  }
}

constexpr size_t gru_lds_fwd(int U, int UW, int ES) { return (size_t)16 * (U + 16 / ES) * ES + (size_t)UW * 4 * 2 * 256 * 4 + (size_t)UW * 16 * 16 * ES; }
constexpr size_t gru_lds_bwd(int U, int UW, int ES) { return (size_t)16 * (2 * U + 16 / ES) * ES + (size_t)UW * 4 * 256 * 4 + (size_t)UW * 16 * 32 * ES; }

template <bool WBF, int U, int UW>
int gru_launch_fwd(const GruFwdDir& a, const GruFwdDir& b, int T, int B, void* xbuf, size_t xbuf_bytes, int xreq, hipStream_t stream) {
  constexpr int ES = WBF ? 2 : 4, NSW = U / (16 * UW);
  const Chunking ck = chunking(T, B, U, 1, UW, ES, gru_lds_fwd(U, UW, ES), U, (const void*)gru_fwd_persist_kernel<WBF, U, UW>);
  for (int lo = 0; lo < B; lo += ck.rows_per_launch) {
    const int cnt = (B - lo < ck.rows_per_launch) ? B - lo : ck.rows_per_launch;
    CRNN_TRY(prep_xbuf(xbuf, xbuf_bytes, ck.xdata_bytes, stream));     // every slot is written per launch: poison first
    const int ncl = 2 * cdiv(cnt, 16);
    hipLaunchKernelGGL((gru_fwd_persist_kernel<WBF, U, UW>), dim3(ncl * NSW), dim3(256 * UW), 0, stream, a, b, T, B, lo, cnt, (unsigned char*)xbuf,
                       (xreq && ncl % 8 == 0) ? 1 : 0);
  }
  return CRNN_OK;
}
template <bool WBF, int U, int UW>
int gru_launch_bwd(const GruBwdDir& a, const GruBwdDir& b, int T, int B, void* xbuf, size_t xbuf_bytes, int xreq, hipStream_t stream) {
  constexpr int ES = WBF ? 2 : 4, NSW = U / (16 * UW);
  const Chunking ck = chunking(T, B, U, 1, UW, ES, gru_lds_bwd(U, UW, ES), 2 * U, (const void*)gru_bwd_persist_kernel<WBF, U, UW>);
  for (int lo = 0; lo < B; lo += ck.rows_per_launch) {
    const int cnt = (B - lo < ck.rows_per_launch) ? B - lo : ck.rows_per_launch;
    CRNN_TRY(prep_xbuf(xbuf, xbuf_bytes, ck.xdata_bytes, stream));
    const int ncl = 2 * cdiv(cnt, 16);
    hipLaunchKernelGGL((gru_bwd_persist_kernel<WBF, U, UW>), dim3(ncl * NSW), dim3(256 * UW), 0, stream, a, b, T, B, lo, cnt, (unsigned char*)xbuf,
                       (xreq && ncl % 8 == 0) ? 1 : 0);
  }
  return CRNN_OK;
}

}  // namespace

// 0 when (u, dt_u) has a persistent GRU kernel (fp32: u in {64,128,256}; bf16: u in {128,256,512}), else -3 (use crnn_gru_*_ex)
extern "C" int crnn_gru_persist_supported(int u, int dt_u) {
  if (dt_u == CRNN_BF16) return (u == 128 || u == 256 || u == 512) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
  return (u == 64 || u == 128 || u == 256) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}

// Forward recurrence of one Bidirectional(GRU) layer in ONE launch.  Arguments as crnn_gru_fwd_ex; xbuf as for crnn_lstm_fwd_persist
// (crnn_lstm_persist_xbuf_bytes(T, B, u, dt_u) bytes, same status words).  flags: 0 or CRNN_RNN_XCD_LOCAL.
extern "C" int crnn_gru_fwd_persist(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1, int ldh,
                                    float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, int dt_u, void* xbuf,
                                    size_t xbuf_bytes, int flags, hipStream_t stream) {
  CRNN_TRY(crnn_gru_persist_supported(u, dt_u));
  if (T < 1 || B < 1 || (((uintptr_t)ut0 | (uintptr_t)ut1) & 15)) return CRNN_ERR_ARG;
  GruFwdDir a{xw0, ut0, h0, ldh, g0, rh0}, b{xw1, ut1, h1, ldh, g1, rh1};
  const int xreq = (flags & CRNN_RNN_XCD_LOCAL) ? 1 : 0;
  int rc;
  if (dt_u == CRNN_BF16) rc = u == 128 ? gru_launch_fwd<true, 128, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                            : u == 256 ? gru_launch_fwd<true, 256, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                                       : gru_launch_fwd<true, 512, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  else rc = u == 64 ? gru_launch_fwd<false, 64, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
          : u == 128 ? gru_launch_fwd<false, 128, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                     : gru_launch_fwd<false, 256, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  CRNN_TRY(rc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// BPTT of one Bidirectional(GRU) layer in ONE launch: fills dz[d] [T][B][3u] from dout[d].  Arguments as crnn_gru_bwd_ex without
// the dh / dhp scratch (both stay in registers).
// crnn_gru_bwd_persist_db: the same launch also leaves the bias-gradient partials db_partials0 / 1 [crnn_rnn_db_rows(B)][3u] (column sums of dz over time per 16-row tile)
extern "C" int crnn_gru_bwd_persist_db(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0, const float* g1,
                                       const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0, float* db_partials1,
                                       int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes, int flags, hipStream_t stream);
extern "C" int crnn_gru_bwd_persist(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0, const float* g1,
                                    const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, int T, int B, int u, int dt_u,
                                    void* xbuf, size_t xbuf_bytes, int flags, hipStream_t stream) {
  return crnn_gru_bwd_persist_db(u0, u1, h0, h1, ldh, g0, g1, dout0, dout1, ldo, dz0, dz1, nullptr, nullptr, T, B, u, dt_u, xbuf, xbuf_bytes, flags, stream);
}
extern "C" int crnn_gru_bwd_persist_db(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0, const float* g1,
                                       const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1, float* db_partials0, float* db_partials1,
                                       int T, int B, int u, int dt_u, void* xbuf, size_t xbuf_bytes, int flags, hipStream_t stream) {
  CRNN_TRY(crnn_gru_persist_supported(u, dt_u));
  if (T < 1 || B < 1 || (((uintptr_t)u0 | (uintptr_t)u1) & 15) || (!db_partials0) != (!db_partials1)) return CRNN_ERR_ARG;
  GruBwdDir a{u0, h0, ldh, g0, dout0, ldo, dz0, db_partials0}, b{u1, h1, ldh, g1, dout1, ldo, dz1, db_partials1};
  const int xreq = (flags & CRNN_RNN_XCD_LOCAL) ? 1 : 0;
  int rc;
  if (dt_u == CRNN_BF16) rc = u == 128 ? gru_launch_bwd<true, 128, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                            : u == 256 ? gru_launch_bwd<true, 256, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                                       : gru_launch_bwd<true, 512, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  else rc = u == 64 ? gru_launch_bwd<false, 64, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
          : u == 128 ? gru_launch_bwd<false, 128, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream)
                     : gru_launch_bwd<false, 256, 2>(a, b, T, B, xbuf, xbuf_bytes, xreq, stream);
  CRNN_TRY(rc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
