// Bidirectional LSTM recurrence (Keras 2.2.2 LSTMCell: hard_sigmoid gates, tanh, gate order i,f,c,o;
// utils.py:77-79), time-major [T][B][.] internally.
//
// One launch per timestep, both directions in the same launch (blockIdx.z).  The per-step work is the
// recurrent gate GEMM  z[B,4u] = h_{t-1}[B,u] * U[u,4u]  on the matrix cores (v_mfma_f32_16x16x4_f32),
// with the gate non-linearities / cell update fused as its epilogue.  A workgroup owns a 16(batch) x
// 16(units) x 4(gates) output tile so the whole cell update is thread-local; its 4 waves split K and
// combine through LDS.  Operands are read straight from L2 (U is 1 MB/direction and shared by every
// workgroup) as 16-byte loads along K: lane (r,q) loads k = kb+4q..4q+3 and feeds 4 MFMAs, i.e. MFMA e
// covers k in {kb+e, kb+4+e, kb+8+e, kb+12+e} for both operands.
// The input projection x*W+b for all timesteps is hoisted into one big GEMM (gemm.hip).
// Backward (BPTT) mirrors it: dh_{t-1} = dz_t * U^T as the per-step GEMM, the gate-gradient math as its
// epilogue; dW/dU/dx/db are whole-sequence GEMMs afterwards.
#include "common.h"
#include "rnn_cell.h"

// ----------------------------------------------------------------------------------------------------------
// Shared step-GEMM tile: acc[g] (16x16, MFMA C/D layout) = sum_k A[b0+r][k] * Brow_g[j0+r'][k] over this wave's
// quarter of K, then reduced across the 4 waves into red[.][g][256] (row-major 16x16).  arow / brow are this
// lane's row pointers (k contiguous).  The loads of PD k-chunks (16 k each) are all issued before their MFMAs, so
// a wave pays the L2 latency once per group instead of once per chunk (the step kernels are latency-bound).
// ----------------------------------------------------------------------------------------------------------
template <int NG, int PD, int MT = 1>
__device__ __forceinline__ void step_tile_gemm(const float* (&arow)[MT], const bool (&valid)[MT], const float* (&brow)[NG], int K, bool skip,
                                               float (*red)[MT * NG][256], int wave, int r, int q) {
  // MT batch tiles (16 rows each) share every B fragment: the recurrent weights are read once per MT*16 batch rows
  f32x4 acc[MT][NG];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!skip) {
    const int kw = K >> 2, kbeg = wave * kw, kend = kbeg + kw;
    int kb = kbeg;
    for (; kb + 16 * PD <= kend; kb += 16 * PD) {
      float4 a4[MT][PD], b4[NG][PD];
#pragma unroll
      for (int c = 0; c < PD; ++c) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a4[m][c] = valid[m] ? *reinterpret_cast<const float4*>(arow[m] + kb + 16 * c + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < NG; ++g) b4[g][c] = *reinterpret_cast<const float4*>(brow[g] + kb + 16 * c + 4 * q);
      }
#pragma unroll
      for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][c].x, b4[g][c].x, acc[m][g], 0, 0, 0);
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][c].y, b4[g][c].y, acc[m][g], 0, 0, 0);
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][c].z, b4[g][c].z, acc[m][g], 0, 0, 0);
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m][c].w, b4[g][c].w, acc[m][g], 0, 0, 0);
          }
    }
    for (; kb < kend; kb += 16) {
      float4 a4[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a4[m] = valid[m] ? *reinterpret_cast<const float4*>(arow[m] + kb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float4 b4 = *reinterpret_cast<const float4*>(brow[g] + kb + 4 * q);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].x, b4.x, acc[m][g], 0, 0, 0);
          acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].y, b4.y, acc[m][g], 0, 0, 0);
          acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].z, b4.z, acc[m][g], 0, 0, 0);
          acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[m].w, b4.w, acc[m][g], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][m * NG + g][(q * 4 + e) * 16 + r] = acc[m][g][e];  // C/D: row = 4q+e, col = r
  __syncthreads();
}

// XCD-aware placement of the step tiles.  Workgroup id -> XCD is id % 8, and an XCD's L2 does not keep the
// recurrent operands across launches, so every step each XCD refetches whatever its tiles touch.  The 8 XCDs are
// therefore arranged as SJ column groups x (8/SJ) batch groups: an XCD only ever sees 1/SJ of the recurrent
// weights and SJ/8 of the state rows (e.g. 20 MB -> 6 MB of L2 fills per LSTM forward step at u = 256, B = 256).
__host__ __device__ inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }
struct StepTile { int dir, bt, jt; };
__host__ __device__ inline int step_grid(int gx, int gy, int SJ) { return 8 * 2 * cdiv_i(gy, 8 / SJ) * (gx / SJ); }
__device__ __forceinline__ StepTile step_tile_map(int gx, int gy, int SJ) {
  const int id = blockIdx.x, xcd = id & 7, loc = id >> 3;
  const int jpg = gx / SJ, bpg = cdiv_i(gy, 8 / SJ);
  const int xj = xcd % SJ, xb = xcd / SJ;
  StepTile t;
  t.jt = xj * jpg + loc % jpg;
  const int rest = loc / jpg;
  t.bt = xb * bpg + rest % bpg;
  t.dir = rest / bpg;
  return t;
}
static inline int step_sj(int gx) { return (gx % 4 == 0) ? 4 : (gx % 2 == 0 ? 2 : 1); }

// bf16-MFMA variant for the throughput modes: the recurrent weights come from a bf16 copy (half the L2 bytes), the
// fp32 state rows are rounded to bf16 while they are packed (v_cvt_pk_bf16_f32) and one v_mfma_f32_16x16x32_bf16
// covers 32 k (8 of the fp32 MFMAs).  Lane (r,q) holds k = kb+8q..8q+7 of row r for both operands.  Requires the
// wave's K quarter to be a multiple of 32.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t pack8_bf16(const float4& lo, const float4& hi) {
  uint4 w = make_uint4(pack2_bf16(lo.x, lo.y), pack2_bf16(lo.z, lo.w), pack2_bf16(hi.x, hi.y), pack2_bf16(hi.z, hi.w));
  return __builtin_bit_cast(bf16x8_t, w);
}
template <int NG, int PD, int MT = 1>
__device__ __forceinline__ void step_tile_gemm_bf16(const float* (&arow)[MT], const bool (&valid)[MT], const bf16_t* (&brow)[NG], int K, bool skip,
                                                    float (*red)[MT * NG][256], int wave, int r, int q) {
  f32x4 acc[MT][NG];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!skip) {
    const int kw = K >> 2, kbeg = wave * kw, kend = kbeg + kw;
    for (int kb = kbeg; kb < kend; kb += 32 * PD) {
      float4 alo[MT][PD], ahi[MT][PD]; uint4 b8[NG][PD];
#pragma unroll
      for (int c = 0; c < PD; ++c) {
        const int k = kb + 32 * c + 8 * q;
        const bool in = kb + 32 * c < kend;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          alo[m][c] = (valid[m] && in) ? *reinterpret_cast<const float4*>(arow[m] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          ahi[m][c] = (valid[m] && in) ? *reinterpret_cast<const float4*>(arow[m] + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) b8[g][c] = in ? *reinterpret_cast<const uint4*>(brow[g] + k) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8_t av = pack8_bf16(alo[m][c], ahi[m][c]);
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, b8[g][c]), acc[m][g], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][m * NG + g][(q * 4 + e) * 16 + r] = acc[m][g][e];  // C/D: row = 4q+e, col = r
  __syncthreads();
}

// MT = batch tiles (x16 rows) per workgroup in the LSTM step kernels: 1 while the step is latency-bound (<= 1 workgroup
// per CU), 2 for large batches where the recurrent weights' L2 traffic per batch row matters (inference at batch 1024)

struct LstmDir {
  const float* xw;   // [T][B][4u]  x*W + b
  const float* wt;   // fwd: U^T [4u][u] ; bwd: U [u][4u]
  float* h; int ldh; // h(t,b,j) = h[(t*B+b)*ldh + j]
  float* c;          // [T][B][u]
  float* gates;      // [T][B][4u] activated i,f,g,o
  const float* dout; int ldo;  // bwd: upstream gradient w.r.t. h, same indexing as h
  float* dz;         // bwd: [T][B][4u] pre-activation gradients
  float* dc;         // bwd: [B][u] cell-gradient carry
};

template <bool WBF, int MT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmDir d0, LstmDir d1, int s, int T, int B, int u, int SJ) {
  __shared__ float red[4][MT * 4][256];
  const StepTile st = step_tile_map(u / 16, cdiv_i(B, 16 * MT), SJ);
  if (st.bt * 16 * MT >= B) return;
  const LstmDir d = st.dir ? d1 : d0;
  const int dir = st.dir;
  const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
  const int b0 = st.bt * (16 * MT), j0 = st.jt * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  float xwv[MT][4], cpv[MT];   // epilogue operands, requested before the GEMM so their latency overlaps it
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int b = b0 + 16 * m + (tid >> 4), j = j0 + (tid & 15);
    const bool ok = b < B;
    const float* xw = d.xw + ((long)t * B + (ok ? b : 0)) * 4 * u;
#pragma unroll
    for (int g = 0; g < 4; ++g) xwv[m][g] = xw[g * u + j];
    cpv[m] = (s > 0) ? d.c[((long)tp * B + (ok ? b : 0)) * u + j] : 0.f;
  }
  {
    bool valid[MT]; const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      valid[m] = (b0 + 16 * m + r) < B;
      arow[m] = d.h + ((long)(s > 0 ? tp : t) * B + (valid[m] ? b0 + 16 * m + r : 0)) * d.ldh;
    }
    if constexpr (WBF) {
      const bf16_t* wb = reinterpret_cast<const bf16_t*>(d.wt);
      const bf16_t* brow[4] = {wb + (long)(j0 + r) * u, wb + (long)(u + j0 + r) * u, wb + (long)(2 * u + j0 + r) * u, wb + (long)(3 * u + j0 + r) * u};
      step_tile_gemm_bf16<4, 2, MT>(arow, valid, brow, u, s == 0, red, wave, r, q);
    } else {
      const float* brow[4] = {d.wt + (long)(j0 + r) * u, d.wt + (long)(u + j0 + r) * u, d.wt + (long)(2 * u + j0 + r) * u,
                              d.wt + (long)(3 * u + j0 + r) * u};
      step_tile_gemm<4, 4, MT>(arow, valid, brow, u, s == 0, red, wave, r, q);
    }
  }
  const int row = tid >> 4, col = tid & 15, j = j0 + col;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int b = b0 + 16 * m + row;
    if (b < B) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        z[g] = ((red[0][m * 4 + g][tid] + red[1][m * 4 + g][tid]) + (red[2][m * 4 + g][tid] + red[3][m * 4 + g][tid])) + xwv[m][g];
      const LstmFwdOut o = lstm_cell_fwd(z, cpv[m]);
      float* gt = d.gates + ((long)t * B + b) * 4 * u;
      gt[j] = o.ig; gt[u + j] = o.fg; gt[2 * u + j] = o.gg; gt[3 * u + j] = o.og;
      d.c[((long)t * B + b) * u + j] = o.cn;
      d.h[((long)t * B + b) * d.ldh + j] = o.hn;
    }
  }
}

// sb = 0..T-1 counts backward steps; the time handled is the (T-1-sb)-th in processing order
template <bool WBF, int MT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmDir d0, LstmDir d1, int sb, int T, int B, int u, int SJ) {
  __shared__ float red[4][MT][256];
  const StepTile st = step_tile_map(u / 16, cdiv_i(B, 16 * MT), SJ);
  if (st.bt * 16 * MT >= B) return;
  const LstmDir d = st.dir ? d1 : d0;
  const int dir = st.dir;
  const int sp = T - 1 - sb;                       // processing index of this time in the forward pass
  const int t = dir ? T - 1 - sp : sp;
  const int tnext = dir ? t - 1 : t + 1;           // processed after t in forward order (already back-propagated)
  const int tprev = dir ? t + 1 : t - 1;           // processed before t in forward order
  const int b0 = st.bt * (16 * MT), j0 = st.jt * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int K = 4 * u;
  float gv[MT][4], ctv[MT], cpv[MT], dcv[MT], dov[MT];   // epilogue operands, requested before the GEMM
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int b = b0 + 16 * m + (tid >> 4), j = j0 + (tid & 15);
    const long bb = (b < B) ? b : 0;
    const float* gt = d.gates + ((long)t * B + bb) * K;
#pragma unroll
    for (int g = 0; g < 4; ++g) gv[m][g] = gt[g * u + j];
    ctv[m] = d.c[((long)t * B + bb) * u + j];
    cpv[m] = (sp > 0) ? d.c[((long)tprev * B + bb) * u + j] : 0.f;
    dcv[m] = (sb > 0) ? d.dc[bb * u + j] : 0.f;
    dov[m] = d.dout[((long)t * B + bb) * d.ldo + j];
  }
  {
    bool valid[MT]; const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      valid[m] = (b0 + 16 * m + r) < B;
      arow[m] = d.dz + ((long)(sb > 0 ? tnext : t) * B + (valid[m] ? b0 + 16 * m + r : 0)) * K;
    }
    if constexpr (WBF) {
      const bf16_t* brow[1] = {reinterpret_cast<const bf16_t*>(d.wt) + (long)(j0 + r) * K};
      step_tile_gemm_bf16<1, 4, MT>(arow, valid, brow, K, sb == 0, red, wave, r, q);
    } else {
      const float* brow[1] = {d.wt + (long)(j0 + r) * K};
      step_tile_gemm<1, 8, MT>(arow, valid, brow, K, sb == 0, red, wave, r, q);
    }
  }
  const int row = tid >> 4, col = tid & 15, j = j0 + col;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int b = b0 + 16 * m + row;
    if (b < B) {
      float dh = ((red[0][m][tid] + red[1][m][tid]) + (red[2][m][tid] + red[3][m][tid])) + dov[m];
      const LstmBwdOut o = lstm_cell_bwd(dh, gv[m][0], gv[m][1], gv[m][2], gv[m][3], ctv[m], cpv[m], dcv[m]);
      float* dz = d.dz + ((long)t * B + b) * K;
      dz[j] = o.dz[0]; dz[u + j] = o.dz[1]; dz[2 * u + j] = o.dz[2]; dz[3 * u + j] = o.dz[3];
      d.dc[(long)b * u + j] = o.dc;
    }
  }
}

static int check_units(int u) { return (u >= 64 && u % 64 == 0) ? 0 : CRNN_ERR_UNSUPPORTED; }

// Forward recurrence of one Bidirectional(LSTM) layer (both directions).  Pointers per direction d in
// {0: forward-in-time, 1: backward-in-time}: xw[d] [T][B][4u], ut[d] = U^T [4u][u], h[d] (row stride ldh),
// c[d] [T][B][u], gates[d] [T][B][4u].  T launches on `stream`.
extern "C" int crnn_lstm_fwd_ex(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1,
                                int ldh, float* c0, float* c1, float* g0, float* g1, int T, int B, int u, int dt_u, hipStream_t stream) {
  CRNN_TRY(check_units(u));
  if (ldh % 4 != 0) return CRNN_ERR_ARG;
  if (dt_u == CRNN_BF16 && (u % 128 != 0 || ((uintptr_t)ut0 | (uintptr_t)ut1) & 15)) return CRNN_ERR_UNSUPPORTED;   // 32-wide bf16 MFMA per K quarter
  LstmDir a{xw0, (const float*)ut0, h0, ldh, c0, g0, nullptr, 0, nullptr, nullptr};
  LstmDir b{xw1, (const float*)ut1, h1, ldh, c1, g1, nullptr, 0, nullptr, nullptr};
  const int SJ = step_sj(u / 16);
  const int MT = (B >= 512) ? 2 : 1;
  dim3 grid(step_grid(u / 16, cdiv(B, 16 * MT), SJ));
  for (int s = 0; s < T; ++s) {
    if (dt_u == CRNN_BF16) {
      if (MT == 2) hipLaunchKernelGGL((lstm_fwd_step_kernel<true, 2>), grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
      else hipLaunchKernelGGL((lstm_fwd_step_kernel<true, 1>), grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
    } else {
      if (MT == 2) hipLaunchKernelGGL((lstm_fwd_step_kernel<false, 2>), grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
      else hipLaunchKernelGGL((lstm_fwd_step_kernel<false, 1>), grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
    }
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_lstm_fwd(const float* xw0, const float* xw1, const float* ut0, const float* ut1, float* h0, float* h1,
                             int ldh, float* c0, float* c1, float* g0, float* g1, int T, int B, int u, hipStream_t stream) {
  return crnn_lstm_fwd_ex(xw0, xw1, ut0, ut1, h0, h1, ldh, c0, c1, g0, g1, T, B, u, CRNN_F32, stream);
}

// BPTT of one Bidirectional(LSTM) layer: fills dz[d] [T][B][4u] (gradients w.r.t. the gate pre-activations)
// from dout[d] (gradient w.r.t. h, row stride ldo).  u_[d] = U [u][4u]; dc[d] = [B][u] scratch.
extern "C" int crnn_lstm_bwd_ex(const void* u0, const void* u1, const float* c0, const float* c1, const float* g0,
                                const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1,
                                float* dc0, float* dc1, int T, int B, int u, int dt_u, hipStream_t stream) {
  CRNN_TRY(check_units(u));
  if (dt_u == CRNN_BF16 && (u % 32 != 0 || ((uintptr_t)u0 | (uintptr_t)u1) & 15)) return CRNN_ERR_UNSUPPORTED;
  LstmDir a{nullptr, (const float*)u0, nullptr, 0, const_cast<float*>(c0), const_cast<float*>(g0), dout0, ldo, dz0, dc0};
  LstmDir b{nullptr, (const float*)u1, nullptr, 0, const_cast<float*>(c1), const_cast<float*>(g1), dout1, ldo, dz1, dc1};
  const int SJ = step_sj(u / 16) > 2 ? 2 : step_sj(u / 16);   // the dz rows (4u wide) outweigh the weights here: favour batch groups
  const int MT = (B >= 512) ? 2 : 1;
  dim3 grid(step_grid(u / 16, cdiv(B, 16 * MT), SJ));
  for (int sb = 0; sb < T; ++sb) {
    if (dt_u == CRNN_BF16) {
      if (MT == 2) hipLaunchKernelGGL((lstm_bwd_step_kernel<true, 2>), grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
      else hipLaunchKernelGGL((lstm_bwd_step_kernel<true, 1>), grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
    } else {
      if (MT == 2) hipLaunchKernelGGL((lstm_bwd_step_kernel<false, 2>), grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
      else hipLaunchKernelGGL((lstm_bwd_step_kernel<false, 1>), grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
    }
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_lstm_bwd(const float* u0, const float* u1, const float* c0, const float* c1, const float* g0,
                             const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1,
                             float* dc0, float* dc1, int T, int B, int u, hipStream_t stream) {
  return crnn_lstm_bwd_ex(u0, u1, c0, c1, g0, g1, dout0, dout1, ldo, dz0, dz1, dc0, dc1, T, B, u, CRNN_F32, stream);
}

// =====================================================================================================
// Bidirectional GRU (Keras 2.2.2 GRUCell, reset_after=False, gate order z,r,h; utils.py:81-82 -- the cell
// train.py really builds):  z = hs(xWz + h Uz), r = hs(xWr + h Ur), hh = tanh(xWh + (r*h) Uh),
// h' = z*h + (1-z)*hh.  The candidate needs r first, so a timestep is two dependent MFMA step-GEMMs:
//   fwd  K1: [z|r] tile  = h_{t-1} * U[:, 0:2u]      epilogue: gates z,r ; rh = r*h_{t-1}
//        K2: hh tile     = rh * U[:, 2u:3u]          epilogue: hh, h_t
//   bwd  KB: dh_t        = dout_t + dhp + [dz|dr]_{t+1} * U[:, 0:2u]^T    epilogue: dz_t, dhh_t
//        KA: d(rh)_t     = dhh_t * U[:, 2u:3u]^T                          epilogue: dr_t ; dhp = dh_t*z + d(rh)*r
// Same 16x16 tile / 4-wave split-K / 16-byte L2 streaming structure as the LSTM kernels above.
// =====================================================================================================
struct GruDir {
  const float* xw;    // [T][B][3u]
  const float* w;     // fwd: U^T [3u][u] ; bwd: U [u][3u]
  float* h; int ldh;  // h(t,b,j)
  float* gates;       // [T][B][3u] z, r, hh
  float* rh;          // [T][B][u]  r * h_prev (kept for dU_h)
  const float* dout; int ldo;
  float* dz;          // [T][B][3u]
  float* dh;          // [B][u] gradient w.r.t. h_t of the step in flight
  float* dhp;         // [B][u] dh*z + d(rh)*r carried to the previous step
};

#define STEP_TILE()                                                                                  \
  const StepTile st = step_tile_map(u / 16, cdiv_i(B, 16), SJ);                                      \
  if (st.bt * 16 >= B) return
#define STEP_IDS()                                                                                   \
  const int b0 = st.bt * 16, j0 = st.jt * 16;                                                        \
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;       \
  const int row = tid >> 4, col = tid & 15, b = b0 + row, j = j0 + col;                              \
  const bool valid1 = (b0 + r) < B;                                                                  \
  const bool valid[1] = {valid1};                                                                    \
  const int ar = valid1 ? b0 + r : 0

template <bool WBF>
__global__ __launch_bounds__(256) void gru_fwd_zr_kernel(GruDir d0, GruDir d1, int s, int T, int B, int u, int SJ) {
  __shared__ float red[4][2][256];
  STEP_TILE();
  const GruDir d = st.dir ? d1 : d0;
  const int dir = st.dir, t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
  STEP_IDS();
  const float* arow[1] = {d.h + ((long)(s > 0 ? tp : t) * B + ar) * d.ldh};
  if constexpr (WBF) {
    const bf16_t* wb = reinterpret_cast<const bf16_t*>(d.w);
    const bf16_t* brow[2] = {wb + (long)(j0 + r) * u, wb + (long)(u + j0 + r) * u};
    step_tile_gemm_bf16<2, 2>(arow, valid, brow, u, s == 0, red, wave, r, q);
  } else {
    const float* brow[2] = {d.w + (long)(j0 + r) * u, d.w + (long)(u + j0 + r) * u};
    step_tile_gemm<2, 4>(arow, valid, brow, u, s == 0, red, wave, r, q);
  }
  if (b < B) {
    const float* xw = d.xw + ((long)t * B + b) * 3 * u;
    float zz = ((red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid])) + xw[j];
    float rr = ((red[0][1][tid] + red[1][1][tid]) + (red[2][1][tid] + red[3][1][tid])) + xw[u + j];
    float hprev = (s > 0) ? d.h[((long)tp * B + b) * d.ldh + j] : 0.f;
    const GruZR o = gru_cell_zr(zz, rr, hprev);
    float* gt = d.gates + ((long)t * B + b) * 3 * u;
    gt[j] = o.zg; gt[u + j] = o.rg;
    d.rh[((long)t * B + b) * u + j] = o.rh;
  }
}

template <bool WBF>
__global__ __launch_bounds__(256) void gru_fwd_h_kernel(GruDir d0, GruDir d1, int s, int T, int B, int u, int SJ) {
  __shared__ float red[4][1][256];
  STEP_TILE();
  const GruDir d = st.dir ? d1 : d0;
  const int dir = st.dir, t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
  STEP_IDS();
  const float* arow[1] = {d.rh + ((long)t * B + ar) * u};
  if constexpr (WBF) {
    const bf16_t* brow[1] = {reinterpret_cast<const bf16_t*>(d.w) + (long)(2 * u + j0 + r) * u};
    step_tile_gemm_bf16<1, 2>(arow, valid, brow, u, s == 0, red, wave, r, q);
  } else {
    const float* brow[1] = {d.w + (long)(2 * u + j0 + r) * u};
    step_tile_gemm<1, 4>(arow, valid, brow, u, s == 0, red, wave, r, q);
  }
  if (b < B) {
    float* gt = d.gates + ((long)t * B + b) * 3 * u;
    float pre = ((red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid])) + d.xw[((long)t * B + b) * 3 * u + 2 * u + j];
    float zg = gt[j];
    float hprev = (s > 0) ? d.h[((long)tp * B + b) * d.ldh + j] : 0.f;
    const GruH o = gru_cell_h(pre, zg, hprev);
    gt[2 * u + j] = o.hh;
    d.h[((long)t * B + b) * d.ldh + j] = o.hn;
  }
}

template <bool WBF>
__global__ __launch_bounds__(256) void gru_bwd_b_kernel(GruDir d0, GruDir d1, int sb, int T, int B, int u, int SJ) {
  __shared__ float red[4][1][256];
  STEP_TILE();
  const GruDir d = st.dir ? d1 : d0;
  const int dir = st.dir, sp = T - 1 - sb, t = dir ? T - 1 - sp : sp;
  const int tnext = dir ? t - 1 : t + 1, tprev = dir ? t + 1 : t - 1;
  STEP_IDS();
  const float* arow[1] = {d.dz + ((long)(sb > 0 ? tnext : t) * B + ar) * 3 * u};
  if constexpr (WBF) {
    const bf16_t* brow[1] = {reinterpret_cast<const bf16_t*>(d.w) + (long)(j0 + r) * 3 * u};
    step_tile_gemm_bf16<1, 4>(arow, valid, brow, 2 * u, sb == 0, red, wave, r, q);
  } else {
    const float* brow[1] = {d.w + (long)(j0 + r) * 3 * u};
    step_tile_gemm<1, 8>(arow, valid, brow, 2 * u, sb == 0, red, wave, r, q);
  }
  if (b < B) {
    float dh = d.dout[((long)t * B + b) * d.ldo + j];
    if (sb > 0) dh += ((red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid])) + d.dhp[(long)b * u + j];
    const float* gt = d.gates + ((long)t * B + b) * 3 * u;
    float zg = gt[j], hh = gt[2 * u + j];
    float hprev = (sp > 0) ? d.h[((long)tprev * B + b) * d.ldh + j] : 0.f;
    float* dz = d.dz + ((long)t * B + b) * 3 * u;
    const GruBwdB o = gru_cell_bwd_b(dh, zg, hh, hprev);
    dz[j] = o.dzz;
    dz[2 * u + j] = o.dhh;
    d.dh[(long)b * u + j] = dh;
  }
}

template <bool WBF>
__global__ __launch_bounds__(256) void gru_bwd_a_kernel(GruDir d0, GruDir d1, int sb, int T, int B, int u, int SJ) {
  __shared__ float red[4][1][256];
  STEP_TILE();
  const GruDir d = st.dir ? d1 : d0;
  const int dir = st.dir, sp = T - 1 - sb, t = dir ? T - 1 - sp : sp;
  const int tprev = dir ? t + 1 : t - 1;
  STEP_IDS();
  const float* arow[1] = {d.dz + ((long)t * B + ar) * 3 * u + 2 * u};
  if constexpr (WBF) {
    const bf16_t* brow[1] = {reinterpret_cast<const bf16_t*>(d.w) + (long)(j0 + r) * 3 * u + 2 * u};
    step_tile_gemm_bf16<1, 2>(arow, valid, brow, u, false, red, wave, r, q);
  } else {
    const float* brow[1] = {d.w + (long)(j0 + r) * 3 * u + 2 * u};
    step_tile_gemm<1, 4>(arow, valid, brow, u, false, red, wave, r, q);
  }
  if (b < B) {
    float drh = (red[0][0][tid] + red[1][0][tid]) + (red[2][0][tid] + red[3][0][tid]);
    const float* gt = d.gates + ((long)t * B + b) * 3 * u;
    float zg = gt[j], rg = gt[u + j];
    float hprev = (sp > 0) ? d.h[((long)tprev * B + b) * d.ldh + j] : 0.f;
    const GruBwdA o = gru_cell_bwd_a(drh, d.dh[(long)b * u + j], zg, rg, hprev);
    d.dz[((long)t * B + b) * 3 * u + u + j] = o.dzr;
    d.dhp[(long)b * u + j] = o.dhp;
  }
}

// Forward recurrence of one Bidirectional(GRU) layer.  xw[d] [T][B][3u] (x*W+b), ut[d] = U^T [3u][u], h[d] (row
// stride ldh), gates[d] [T][B][3u], rh[d] [T][B][u].  2T launches.
extern "C" int crnn_gru_fwd_ex(const float* xw0, const float* xw1, const void* ut0, const void* ut1, float* h0, float* h1,
                               int ldh, float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, int dt_u, hipStream_t stream) {
  CRNN_TRY(check_units(u));
  if (ldh % 4 != 0) return CRNN_ERR_ARG;
  if (dt_u == CRNN_BF16 && (u % 128 != 0 || ((uintptr_t)ut0 | (uintptr_t)ut1) & 15)) return CRNN_ERR_UNSUPPORTED;
  GruDir a{xw0, (const float*)ut0, h0, ldh, g0, rh0, nullptr, 0, nullptr, nullptr, nullptr};
  GruDir b{xw1, (const float*)ut1, h1, ldh, g1, rh1, nullptr, 0, nullptr, nullptr, nullptr};
  const int SJ = step_sj(u / 16);
  dim3 grid(step_grid(u / 16, cdiv(B, 16), SJ));
  for (int s = 0; s < T; ++s) {
    if (dt_u == CRNN_BF16) {
      hipLaunchKernelGGL(gru_fwd_zr_kernel<true>, grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
      hipLaunchKernelGGL(gru_fwd_h_kernel<true>, grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
    } else {
      hipLaunchKernelGGL(gru_fwd_zr_kernel<false>, grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
      hipLaunchKernelGGL(gru_fwd_h_kernel<false>, grid, dim3(256), 0, stream, a, b, s, T, B, u, SJ);
    }
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_gru_fwd(const float* xw0, const float* xw1, const float* ut0, const float* ut1, float* h0, float* h1,
                            int ldh, float* g0, float* g1, float* rh0, float* rh1, int T, int B, int u, hipStream_t stream) {
  return crnn_gru_fwd_ex(xw0, xw1, ut0, ut1, h0, h1, ldh, g0, g1, rh0, rh1, T, B, u, CRNN_F32, stream);
}

// BPTT of one Bidirectional(GRU) layer: fills dz[d] [T][B][3u] (gradients w.r.t. the z, r, hh pre-activations).
// u_[d] = U [u][3u]; h[d] as in the forward; dh[d], dhp[d] = [B][u] scratch.
extern "C" int crnn_gru_bwd_ex(const void* u0, const void* u1, const float* h0, const float* h1, int ldh, const float* g0,
                               const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1,
                               float* dh0, float* dh1, float* dhp0, float* dhp1, int T, int B, int u, int dt_u, hipStream_t stream) {
  CRNN_TRY(check_units(u));
  if (dt_u == CRNN_BF16 && (u % 128 != 0 || ((uintptr_t)u0 | (uintptr_t)u1) & 15)) return CRNN_ERR_UNSUPPORTED;
  GruDir a{nullptr, (const float*)u0, const_cast<float*>(h0), ldh, const_cast<float*>(g0), nullptr, dout0, ldo, dz0, dh0, dhp0};
  GruDir b{nullptr, (const float*)u1, const_cast<float*>(h1), ldh, const_cast<float*>(g1), nullptr, dout1, ldo, dz1, dh1, dhp1};
  const int SJ = step_sj(u / 16) > 2 ? 2 : step_sj(u / 16);
  dim3 grid(step_grid(u / 16, cdiv(B, 16), SJ));
  for (int sb = 0; sb < T; ++sb) {
    if (dt_u == CRNN_BF16) {
      hipLaunchKernelGGL(gru_bwd_b_kernel<true>, grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
      hipLaunchKernelGGL(gru_bwd_a_kernel<true>, grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
    } else {
      hipLaunchKernelGGL(gru_bwd_b_kernel<false>, grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
      hipLaunchKernelGGL(gru_bwd_a_kernel<false>, grid, dim3(256), 0, stream, a, b, sb, T, B, u, SJ);
    }
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_gru_bwd(const float* u0, const float* u1, const float* h0, const float* h1, int ldh, const float* g0,
                            const float* g1, const float* dout0, const float* dout1, int ldo, float* dz0, float* dz1,
                            float* dh0, float* dh1, float* dhp0, float* dhp1, int T, int B, int u, hipStream_t stream) {
  return crnn_gru_bwd_ex(u0, u1, h0, h1, ldh, g0, g1, dout0, dout1, ldo, dz0, dz1, dh0, dh1, dhp0, dhp1, T, B, u, CRNN_F32, stream);
}

// out[c][r] = in[r][c]  (U -> U^T once per weight update; tiny); TO = float or bf16_t
template <typename TO>
__global__ void transpose_kernel(const float* __restrict__ in, TO* __restrict__ out, int R, int C) {
  __shared__ float tile[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r = blockIdx.y * 32 + threadIdx.y;
  for (int k = 0; k < 32; k += 8) if (r + k < R && c < C) tile[threadIdx.y + k][threadIdx.x] = in[(long)(r + k) * C + c];
  __syncthreads();
  int oc = blockIdx.y * 32 + threadIdx.x, orow = blockIdx.x * 32 + threadIdx.y;
  for (int k = 0; k < 32; k += 8) if (orow + k < C && oc < R) st1(&out[(long)(orow + k) * R + oc], tile[threadIdx.x][threadIdx.y + k]);
}
extern "C" int crnn_transpose_ex(const float* in, void* out, int R, int C, int dt_out, hipStream_t stream) {
  if (dt_out == CRNN_BF16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream, in, (bf16_t*)out, R, C);
  else hipLaunchKernelGGL(transpose_kernel<float>, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, stream, in, (float*)out, R, C);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_transpose(const float* in, float* out, int R, int C, hipStream_t stream) {
  return crnn_transpose_ex(in, out, R, C, CRNN_F32, stream);
}

// several small transposes in one launch (blockIdx.z = matrix): the W^T copies of the pointwise-conv weights
// (one-dimensional grid: matrix z owns the blocks first[z] .. first[z + 1] - 1, cdiv(C, 32) per block row.  A grid sized by the largest matrix in every
//  dimension launched 18 432 workgroups for the eight of the forward once dense1's 4608 x 128 weight joined them, seven eighths of them empty: 9 us)
struct TransTable { long in_off[8], out_off[8]; int R[8], C[8]; int first[9]; };
template <typename TO>
__global__ void transpose_batch_kernel(const float* __restrict__ src, TO* __restrict__ dst, TransTable tab) {
  __shared__ float tile[32][33];
  int z = 0;
  while (z < 7 && (int)blockIdx.x >= tab.first[z + 1]) ++z;
  const int R = tab.R[z], C = tab.C[z];
  const float* in = src + tab.in_off[z]; TO* out = dst + tab.out_off[z];
  const int nbx = (C + 31) / 32, lb = blockIdx.x - tab.first[z], bx = lb % nbx, by = lb / nbx;
  int c = bx * 32 + threadIdx.x, r = by * 32 + threadIdx.y;
  for (int k = 0; k < 32; k += 8) if (r + k < R && c < C) tile[threadIdx.y + k][threadIdx.x] = in[(long)(r + k) * C + c];
  __syncthreads();
  int oc = by * 32 + threadIdx.x, orow = bx * 32 + threadIdx.y;
  for (int k = 0; k < 32; k += 8) if (orow + k < C && oc < R) st1(&out[(long)(orow + k) * R + oc], tile[threadIdx.x][threadIdx.y + k]);
}
extern "C" int crnn_transpose_batch(const float* src, void* dst, int n, const long* in_off, const long* out_off, const int* R, const int* C,
                                    int dt_out, hipStream_t stream) {
  if (n <= 0 || n > 8) return CRNN_ERR_ARG;
  TransTable tab; int blocks = 0;
  for (int i = 0; i < 8; ++i) {
    int j = i < n ? i : 0;
    if (R[j] <= 0 || C[j] <= 0) return CRNN_ERR_ARG;
    tab.in_off[i] = in_off[j]; tab.out_off[i] = out_off[j]; tab.R[i] = R[j]; tab.C[i] = C[j];
    tab.first[i] = blocks;
    if (i < n) blocks += cdiv(C[j], 32) * cdiv(R[j], 32);
  }
  tab.first[8] = blocks;
  dim3 grid(blocks);
  if (dt_out == CRNN_BF16) hipLaunchKernelGGL(transpose_batch_kernel<bf16_t>, grid, dim3(32, 8), 0, stream, src, (bf16_t*)dst, tab);
  else hipLaunchKernelGGL(transpose_batch_kernel<float>, grid, dim3(32, 8), 0, stream, src, (float*)dst, tab);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
