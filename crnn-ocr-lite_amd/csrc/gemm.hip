// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, == an fmaf chain).
//
// One kernel template serves every matmul-shaped op of the CRNN hot path:
//   NN  C[M,N] = A[M,K]  * B[K,N]      pointwise 1x1 conv fwd (utils.py:47), Dense fwd (utils.py:74,85,253,256),
//                                      RNN input GEMM, STN im2col convs (utils.py:249,251)
//   NT  C[M,N] = A[M,K]  * Bt[N,K]^T   data gradients (dX = dY * W^T)
//   TN  C[M,N] = At[K,M]^T * B[K,N]    weight gradients (dW = X^T * dY), reduction over the huge row
//                                      dimension split across workgroups (deterministic 2-stage sum)
// 128 x {128,64} x 32 block tile, 4 waves (64 lanes each), each wave owns 2x2 / 1x2 MFMA 32x32
// accumulators.  Operands are staged through LDS; "row-major-in-k" operands are stored
// [rows][BK+4] and read as ds_read_b128 (conflict-free: 144-B row stride), "k-major" operands are
// stored [BK][rows] and read as ds_read_b32.  The k order inside a BK chunk is permuted
// (k = 8*kk8 + 4*half + e) identically for A and B so one b128 read feeds four MFMAs.
// Workgroup ids are remapped so that tiles sharing the same A rows run on the same XCD (L2).
#include "common.h"
#include <stdlib.h>

#define GBK 32
#define GLDM (GBK + 4)

struct GemmParams {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  const float* bias;
  int act;         // 0 none, 1 relu
  int accumulate;  // C += result
  int permP;       // 0: none; else out_row = (m % P) * (M / P) + m / P
  int klen;        // K range per split (blockIdx.y); nsplit = gridDim.y
  int vecA, vecB;  // 16-byte vector loads legal for the operand
  int vecC;        // 16-byte stores legal for C (and the split scratch)
  int dtA, dtB, dtC;  // storage of the operands / result (CRNN_F32 | CRNN_BF16); the fp32 kernel requires all CRNN_F32
  int tilesN;
  int xsplit;      // > 0: 1-D grid of tiles * xsplit workgroups, K split xsplit-fold with split s on XCD s % 8 (all tiles of one K range
                   // share an L2): id -> xcd = id & 7, tile = (id >> 3) % tiles, split = ((id >> 3) / tiles) * 8 + xcd.  0: 2-D grid (tile, split)
  float* stats;    // optional [tilesM][2][N]: per-tile column sums / sums of squares of the result as stored (BatchNorm statistics)
  const float* cscale; const float* cshift;   // optional per-column epilogue  C = ReLU6(C * cscale[n] + cshift[n])  (inference BatchNorm folded in)
  // optional producer prologue on A (bf16 kernel, bf16 A): the operand the MFMA sees is ReLU6(A * ascale[ch] + ashift[ch])
  // rounded to bf16, ch = the reduction index (modes 0/1) or the A row (mode 2): the BatchNorm + ReLU6 between a depthwise
  // and a pointwise convolution, applied while the tile is staged instead of in a pass of its own
  const float* ascale; const float* ashift;
  // optional BatchNorm-BACKWARD statistics from the epilogue (bf16-family tile kernels, fp32 result, whole tiles): C is the gradient da that arrives at
  // a ReLU6(BatchNorm(d)); bnpart [tilesM][2][N] = per-tile column sums of gy and gy * xhat, gy = C where 0 < d * scale + shift < 6,
  // xhat = (d - mean) / sqrt(var + eps); bnD [M][ldd] fp32 = d, bnstate = [mean | var | scale | shift] x N
  const float* bnD; int ldd; const float* bnstate; float* bnpart;
  // three-plane kernel: an operand handed over as its three bf16 planes (crnn_split3_planes; plane pl of element i at Xpl[pl * xpls + i], the leading
  // dimension of the fp32 operand it stands for) -- null: split from fp32 while staging
  const unsigned short* Apl; const unsigned short* Bpl; long apls, bpls;
#ifdef CRNN_GEMM_EXP
  unsigned long long* trace;   // ablation build only: s_memrealtime stamps of workgroup 300, thread 0
  int exp;         // ablation build only (scripts/gemm_ablate.py): 1 no C stores, 2 no MFMA, 4 B loaded once, 8 A loaded once
#endif
};

// ---- statistics epilogue: per-tile column sums / sums of squares of the result as it will be stored, taken straight
// from the MFMA accumulators (a lane owns one column of each 32x32 block: 16 rows x TM blocks per column block), then
// combined across the two lane halves (shuffle) and the waves stacked along M (LDS), all in a fixed order.
typedef float f32x16_stats __attribute__((ext_vector_type(16)));
template <int TM, int TN>
__device__ __forceinline__ void tile_stats_regs(const f32x16_stats (&acc)[TM][TN], int row_base, int M, int dtC, int half,
                                                float (&ssum)[TN], float (&ssq)[TN]) {
  const bool full = row_base + TM * 32 <= M;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][j][e];
        if (dtC == CRNN_BF16) v = __uint_as_float(pack2_bf16(v, 0.f) << 16);   // the value the consumer reads back
        if (!full && row_base + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half >= M) v = 0.f;
        if (e & 1) { s1 += v; q1 = fmaf(v, v, q1); } else { s0 += v; q0 = fmaf(v, v, q0); }
      }
    float sv = s0 + s1, qv = q0 + q1;
    sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64);
    ssum[j] = sv; ssq[j] = qv;
  }
}
// smem: [2][WAVES_M][BN]; lanes of half 0 deposit their wave's column sums, then one thread per (stat, column) adds the waves
template <int BN, int TN, int WAVES_M>
__device__ __forceinline__ void tile_stats_finish(float* smem, float* stats, int tm, int n0, int N, int tid, int wmi, int wn0,
                                                  int half, int l31, const float (&ssum)[TN], const float (&ssq)[TN]) {
  if (half == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      smem[(0 * WAVES_M + wmi) * BN + wn0 + j * 32 + l31] = ssum[j];
      smem[(1 * WAVES_M + wmi) * BN + wn0 + j * 32 + l31] = ssq[j];
    }
  }
  __syncthreads();
  if (tid < 2 * BN) {
    const int v = tid / BN, c = tid % BN;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES_M; ++q) a += smem[(v * WAVES_M + q) * BN + c];
    if (n0 + c < N) stats[((long)tm * 2 + v) * N + n0 + c] = a;
  }
}

template <bool KM, int ROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, int ld, int row0, int nrows_total,
                                          int k0, int kend, int vec, int tid, float4 (&r)[ROWS / 32]) {
  // ROWS x GBK tile -> ROWS*8 float4 -> ROWS/32 per thread (256 threads)
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    int idx = tid + it * 256;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!KM) {
      int row = idx >> 3, k4 = idx & 7;
      int gr = row0 + row, gk = k0 + 4 * k4;
      if (gr < nrows_total && gk < kend) {
        const float* p = X + (long)gr * ld + gk;
        if (vec) v = *reinterpret_cast<const float4*>(p);
        else {
          v.x = p[0];
          if (gk + 1 < kend) v.y = p[1];
          if (gk + 2 < kend) v.z = p[2];
          if (gk + 3 < kend) v.w = p[3];
        }
      }
    } else {
      int krow = idx / (ROWS / 4), c4 = idx % (ROWS / 4);
      int gk = k0 + krow, gc = row0 + 4 * c4;
      if (gk < kend && gc < nrows_total) {
        const float* p = X + (long)gk * ld + gc;
        if (vec) v = *reinterpret_cast<const float4*>(p);
        else {
          v.x = p[0];
          if (gc + 1 < nrows_total) v.y = p[1];
          if (gc + 2 < nrows_total) v.z = p[2];
          if (gc + 3 < nrows_total) v.w = p[3];
        }
      }
    }
    r[it] = v;
  }
}

template <bool KM, int ROWS>
__device__ __forceinline__ void store_tile(float* Xs, int tid, const float4 (&r)[ROWS / 32]) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    int idx = tid + it * 256;
    if (!KM) {
      int row = idx >> 3, k4 = idx & 7;
      *reinterpret_cast<float4*>(&Xs[row * GLDM + 4 * k4]) = r[it];
    } else {
      int krow = idx / (ROWS / 4), c4 = idx % (ROWS / 4);
      *reinterpret_cast<float4*>(&Xs[krow * ROWS + 4 * c4]) = r[it];
    }
  }
}

template <bool KM, int ROWS>
__device__ __forceinline__ void read_frag(const float* Xs, int r0, int kk8, int half, int l31, float (&f)[4]) {
  if (!KM) {
    float4 v = *reinterpret_cast<const float4*>(&Xs[(r0 + l31) * GLDM + kk8 * 8 + 4 * half]);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = Xs[(kk8 * 8 + 4 * half + e) * ROWS + r0 + l31];
  }
}

template <int BN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
  constexpr int BM = 128;
  constexpr int WAVES_N = (BN == 128) ? 2 : 1;
  constexpr int WM = (BN == 128) ? 64 : 32;  // rows per wave
  constexpr int TM = WM / 32, TN = 2;        // MFMA tiles per wave (wave covers WM x 64)
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * GLDM];
  float* As = smem;
  float* Bs = smem + BM * GLDM;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 64;

  // XCD-aware bijective remap of the tile id (blocks b, b+8, ... share an L2)
  int nwg = gridDim.x, bid = blockIdx.x;
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int tm = lid / p.tilesN, tn = lid % p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kbeg = blockIdx.y * p.klen;
  const int kend = min(p.K, kbeg + p.klen);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[BM / 32], rb[BN / 32];
  load_tile<A_KM, BM>(p.A, p.lda, m0, p.M, kbeg, kend, p.vecA, tid, ra);
  load_tile<B_KM, BN>(p.B, p.ldb, n0, p.N, kbeg, kend, p.vecB, tid, rb);

  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    store_tile<A_KM, BM>(As, tid, ra);
    store_tile<B_KM, BN>(Bs, tid, rb);
    __syncthreads();
    if (k0 + GBK < kend) {  // prefetch the next chunk; in flight during the MFMAs below
      load_tile<A_KM, BM>(p.A, p.lda, m0, p.M, k0 + GBK, kend, p.vecA, tid, ra);
      load_tile<B_KM, BN>(p.B, p.ldb, n0, p.N, k0 + GBK, kend, p.vecB, tid, rb);
    }
#pragma unroll
    for (int kk8 = 0; kk8 < 4; ++kk8) {
      float fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<A_KM, BM>(As, wm0 + i * 32, kk8, half, l31, fa[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<B_KM, BN>(Bs, wn0 + j * 32, kk8, half, l31, fb[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  const bool split = gridDim.y > 1;
  float* Cout = split ? p.C + (long)blockIdx.y * p.M * p.N : p.C;
  const int ldc = split ? p.N : p.ldc;
  const int Q = p.permP ? p.M / p.permP : 0;
  // ---- epilogue: stage the accumulators through LDS (two 64-row halves) so that every lane stores 16
  // contiguous bytes (512 B per output row per wave) instead of 64 separate 4-byte-per-lane stores.
  constexpr int CLD = BN + 4;                      // padded row stride of the staged C half-tile
  constexpr int ROWS_PER_IT = 256 / (BN / 4);      // rows covered by the 256 threads per store iteration
  const bool vecC = p.vecC && !((ldc & 3) | (n0 & 3));
  float st_sum[TN], st_sq[TN];
  if (p.stats) tile_stats_regs<TM, TN>(acc, m0 + wm0, p.M, CRNN_F32, half, st_sum, st_sq);
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
    if (wm0 >= 64 * hp && wm0 < 64 * hp + 64) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            smem[(wm0 - 64 * hp + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * CLD + wn0 + j * 32 + l31] = acc[i][j][e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 64 / ROWS_PER_IT; ++it) {
      const int rl = it * ROWS_PER_IT + tid / (BN / 4), c4 = tid % (BN / 4);
      const int gm = m0 + 64 * hp + rl, gn = n0 + 4 * c4;
      if (gm < p.M && gn < p.N) {
        float4 v = *reinterpret_cast<const float4*>(&smem[rl * CLD + 4 * c4]);
        int orow = gm;
        if (!split && p.permP) orow = (gm % p.permP) * Q + gm / p.permP;
        float* dst = Cout + (long)orow * ldc + gn;
        float vv[4] = {v.x, v.y, v.z, v.w};
        if (!split) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gn + e < p.N) {
              if (p.bias) vv[e] += p.bias[gn + e];
              if (p.act == 1) vv[e] = fmaxf(vv[e], 0.f);
              if (p.cscale) vv[e] = relu6f(fmaf(vv[e], p.cscale[gn + e], p.cshift[gn + e]));
            }
        }
        if (vecC && gn + 3 < p.N) {
          float4 o = make_float4(vv[0], vv[1], vv[2], vv[3]);
          if (!split && p.accumulate) { float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
          *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gn + e < p.N) dst[e] = vv[e] + ((!split && p.accumulate) ? dst[e] : 0.f);
        }
      }
    }
    __syncthreads();
  }
  if (p.stats) tile_stats_finish<BN, TN, 4 / WAVES_N>(smem, p.stats, tm, n0, p.N, tid, wave / WAVES_N, wn0, half, l31, st_sum, st_sq);
}

// second stage of a split reduction: C = act(sum_z part[z] + bias) (+ C); fixed summation order (deterministic).
template <int LY>
__global__ __launch_bounds__(32 * LY) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int nsplit, GemmParams p) {
  // blockDim (32, LY): 32 consecutive outputs x LY split-lanes (8 for short split lists, 32 for long ones), four
  // independent loads in flight per lane
  __shared__ float red[LY][33];
  const long total = (long)p.M * p.N;
  const long i = (long)blockIdx.x * 32 + threadIdx.x;
  float s = 0.f;
  if (i < total) {
    int z = threadIdx.y;
    for (; z + 3 * LY < nsplit; z += 4 * LY) {
      float v0 = part[(long)z * total + i], v1 = part[(long)(z + LY) * total + i];
      float v2 = part[(long)(z + 2 * LY) * total + i], v3 = part[(long)(z + 3 * LY) * total + i];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; z < nsplit; z += LY) s += part[(long)z * total + i];
  }
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0 && i < total) {
    s = 0.f;
#pragma unroll
    for (int r = 0; r < LY; r += 4)
      s += (red[r][threadIdx.x] + red[r + 1][threadIdx.x]) + (red[r + 2][threadIdx.x] + red[r + 3][threadIdx.x]);
    int m = (int)(i / p.N), n = (int)(i % p.N);
    if (p.bias) s += p.bias[n];
    if (p.act == 1) s = fmaxf(s, 0.f);
    int orow = m;
    if (p.permP) orow = (m % p.permP) * (p.M / p.permP) + m / p.permP;
    if (p.dtC == CRNN_BF16) {
      bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + (long)orow * p.ldc + n;
      if (p.accumulate) s += ld1(dst);
      st1(dst, s);
    } else {
      float* dst = p.C + (long)orow * p.ldc + n;
      if (p.accumulate) s += *dst;
      *dst = s;
    }
  }
}
// The same second stage for the common case -- at most 8 partial results, fp32 result, N % 4 == 0 and 16-byte aligned rows: one thread sums
// four consecutive outputs over the splits in registers (the partial lists are 7 - 55 MB: the LDS-combining form above moved 32 outputs
// per 256-thread workgroup and ran dense1's forward reduction at a quarter of the memory bandwidth).  Fixed order: ((p0+p1)+(p2+p3)) +
// ((p4+p5)+(p6+p7)), absent partials count as zero.
__global__ __launch_bounds__(256) void gemm_splitk_reduce8_kernel(const float* __restrict__ part, int nsplit, GemmParams p) {
  const long total4 = ((long)p.M * p.N) >> 2;
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const float4* src = reinterpret_cast<const float4*>(part);
  float4 v[8];
#pragma unroll
  for (int z = 0; z < 8; ++z) v[z] = z < nsplit ? src[(long)z * total4 + i4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float s[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a0 = (&v[0].x)[e], a1 = (&v[1].x)[e], a2 = (&v[2].x)[e], a3 = (&v[3].x)[e];
    const float a4 = (&v[4].x)[e], a5 = (&v[5].x)[e], a6 = (&v[6].x)[e], a7 = (&v[7].x)[e];
    s[e] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  const long i = i4 << 2;
  const int m = (int)(i / p.N), n = (int)(i % p.N);
  if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); s[0] += b.x; s[1] += b.y; s[2] += b.z; s[3] += b.w; }
  if (p.act == 1) { s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f); }
  int orow = m;
  if (p.permP) orow = (m % p.permP) * (p.M / p.permP) + m / p.permP;
  float4* dst = reinterpret_cast<float4*>(p.C + (long)orow * p.ldc + n);
  if (p.accumulate) { const float4 c = *dst; s[0] += c.x; s[1] += c.y; s[2] += c.z; s[3] += c.w; }
  *dst = make_float4(s[0], s[1], s[2], s[3]);
}
// launches the matching second stage
static inline void launch_splitk_reduce(const float* scratch, int nsplit, const GemmParams& p, hipStream_t stream) {
  const long total = (long)p.M * p.N;
  const bool vec = nsplit <= 8 && p.dtC == CRNN_F32 && (p.N & 3) == 0 && (p.ldc & 3) == 0 && !(((uintptr_t)p.C | (uintptr_t)scratch) & 15) &&
                   (!p.bias || !((uintptr_t)p.bias & 15));
  if (vec) hipLaunchKernelGGL(gemm_splitk_reduce8_kernel, dim3(cdiv(total >> 2, 256)), dim3(256), 0, stream, scratch, nsplit, p);
  else if (nsplit > 32) hipLaunchKernelGGL(gemm_splitk_reduce_kernel<32>, dim3(cdiv(total, 32)), dim3(32, 32), 0, stream, scratch, nsplit, p);
  else hipLaunchKernelGGL(gemm_splitk_reduce_kernel<8>, dim3(cdiv(total, 32)), dim3(32, 8), 0, stream, scratch, nsplit, p);
}

static inline int aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// mode: 0 = NN, 1 = NT, 2 = TN.  `scratch` (scratch_bytes) is needed only when the reduction is split
// (mode 2 with few output tiles); pass nullptr/0 to forbid splitting.
static int gemm_f32_impl(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                         int ldc, const float* bias, int act, int accumulate, int permP, float* scratch,
                         size_t scratch_bytes, float* stats, hipStream_t stream, const float* cscale = nullptr, const float* cshift = nullptr) {
  if (M <= 0 || N <= 0 || K <= 0) return CRNN_ERR_ARG;
  if (permP && (M % permP) != 0) return CRNN_ERR_ARG;
  if (stats && (bias || act || accumulate || permP || scratch || cscale)) return CRNN_ERR_ARG;   // statistics of the plain product only
  if (cscale && (scratch || accumulate || !cshift)) return CRNN_ERR_ARG;                          // no split reduction with the folded BatchNorm
  GemmParams p;
  p.xsplit = 0;
  p.stats = stats; p.cscale = cscale; p.cshift = cshift; p.ascale = nullptr; p.ashift = nullptr;
  p.bnD = nullptr; p.ldd = 0; p.bnstate = nullptr; p.bnpart = nullptr;
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.bias = bias; p.act = act; p.accumulate = accumulate; p.permP = permP;
  p.dtA = p.dtB = p.dtC = CRNN_F32;
  const bool a_km = (mode == 2), b_km = (mode != 1);
  // contiguous extent of each operand: A: K (m-major) or M (k-major); B: N (k-major) or K (n-major)
  p.vecA = aligned16(A) && (lda % 4 == 0) && ((a_km ? M : K) % 4 == 0);
  p.vecB = aligned16(B) && (ldb % 4 == 0) && ((b_km ? N : K) % 4 == 0);
  p.vecC = aligned16(C) && (ldc % 4 == 0) && (N % 4 == 0) && (!scratch || aligned16(scratch));
  const int BN = (N <= 64) ? 64 : 128;
  const int tilesM = cdiv(M, 128), tilesN = cdiv(N, BN);
  p.tilesN = tilesN;
  int tiles = tilesM * tilesN;
  int nsplit = 1;
  const int split_target = crnn_knob("CRNN_SPLIT_WGS", 768);   // workgroups a split reduction aims for: 3 resident per CU
  if (scratch && ((tiles < 256 && K >= 2048) || (tiles <= 16 && K >= 512))) {
    nsplit = cdiv(split_target, tiles);
    int maxs = K / (K >= 2048 ? 512 : 128); if (maxs < 1) maxs = 1;
    if (nsplit > maxs) nsplit = maxs;
    size_t per = (size_t)M * N * sizeof(float);
    size_t fit = scratch_bytes / per;
    if ((size_t)nsplit > fit) nsplit = (int)fit;
    if (nsplit < 1) nsplit = 1;
  }
  int klen = cdiv(K, nsplit);
  klen = ((klen + GBK - 1) / GBK) * GBK;
  nsplit = cdiv(K, klen);
  p.klen = klen;
  GemmParams pk = p;
  if (nsplit > 1) pk.C = scratch;
  dim3 grid(tiles, nsplit), block(256);
#define LAUNCH(BNV, AK, BKM) hipLaunchKernelGGL((gemm_f32_kernel<BNV, AK, BKM>), grid, block, 0, stream, pk)
  if (BN == 128) {
    if (mode == 0) LAUNCH(128, false, true); else if (mode == 1) LAUNCH(128, false, false); else LAUNCH(128, true, true);
  } else {
    if (mode == 0) LAUNCH(64, false, true); else if (mode == 1) LAUNCH(64, false, false); else LAUNCH(64, true, true);
  }
#undef LAUNCH
  CRNN_LAUNCH_CHECK();
  if (nsplit > 1) {
    launch_splitk_reduce(scratch, nsplit, p, stream);
    CRNN_LAUNCH_CHECK();
  }
  return CRNN_OK;
}

extern "C" int crnn_gemm_f32(int mode, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                             int ldc, const float* bias, int act, int accumulate, int permP, float* scratch,
                             size_t scratch_bytes, hipStream_t stream) {
  return gemm_f32_impl(mode, A, B, C, M, N, K, lda, ldb, ldc, bias, act, accumulate, permP, scratch, scratch_bytes, nullptr, stream);
}

#include "gemm_bf16.inc"

// ---- pointwise 1x1 convolution = GEMM over the pixels, with the next BatchNorm's statistics from the epilogue
extern "C" int crnn_pwconv_stat_rows(long M) { return cdiv(M, 128); }
extern "C" int crnn_pwconv_fwd(const void* a, const void* w, void* q, long M, int N, int K, float* stat_partials,
                               const float* out_bnstate, int bf16_products, int dt_a, int dt_w, int dt_q, int w_transposed,
                               hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL) return CRNN_ERR_ARG;
  // w_transposed: the weights are given as W^T [N][K] (both operands then contiguous along the reduction: the staging
  // needs no k-pair interleave and the fragments are single 16-byte LDS reads)
  const int mode = w_transposed ? 1 : 0, ldw = w_transposed ? K : N;
  // out_bnstate ([mean|var|scale|shift] of the BatchNorm after the conv, inference): q = ReLU6(product * scale + shift)
  const float* cs = out_bnstate ? out_bnstate + 2L * N : nullptr; const float* ch = out_bnstate ? out_bnstate + 3L * N : nullptr;
  if (bf16_products == 2 || bf16_products == 3)   // fp32 tensors: 2 = fp32-accurate three-plane bf16 products (crnn_gemm_f32x3), 3 = two planes (crnn_gemm_f32x2)
    return gemm_bf16_impl(mode, a, w, q, (int)M, N, K, K, ldw, N, nullptr, 0, 0, 0, nullptr, 0, dt_a, dt_w, dt_q, stat_partials, stream, cs, ch,
                          nullptr, nullptr, true, nullptr, nullptr, bf16_products == 2 ? 3 : 2);
  if (bf16_products)
    return gemm_bf16_impl(mode, a, w, q, (int)M, N, K, K, ldw, N, nullptr, 0, 0, 0, nullptr, 0, dt_a, dt_w, dt_q, stat_partials, stream, cs, ch);
  if (dt_a != CRNN_F32 || dt_w != CRNN_F32 || dt_q != CRNN_F32) return CRNN_ERR_ARG;
  return gemm_f32_impl(mode, (const float*)a, (const float*)w, (float*)q, (int)M, N, K, K, ldw, N, nullptr, 0, 0, 0, nullptr, 0, stat_partials, stream, cs, ch);
}

// The same convolution fed by the PRE-BatchNorm depthwise output d (bf16): the operand is ReLU6(BN(d)) (utils.py:45-46),
// formed per element while the tile is staged (GemmParams::ascale/ashift), so the activated tensor is never written.
extern "C" int crnn_pwconv_bnrelu6_fwd(const void* d, const float* in_bnstate, const void* w, void* q, long M, int N, int K,
                                       float* stat_partials, int dt_q, int w_transposed, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  const int mode = w_transposed ? 1 : 0, ldw = w_transposed ? K : N;
  return gemm_bf16_impl(mode, d, w, q, (int)M, N, K, K, ldw, N, nullptr, 0, 0, 0, nullptr, 0, CRNN_BF16, CRNN_BF16, dt_q, stat_partials, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K);
}
// ... and its weight gradient dw[K][N] = ReLU6(BN(d))^T [K][M] * g[M][N] (fp32 result; g bf16)
extern "C" int crnn_pwconv_bnrelu6_wgrad(const void* d, const float* in_bnstate, const void* g, float* dw, long M, int N, int K,
                                         float* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  return gemm_bf16_impl(2, d, g, dw, K, N, (int)M, K, N, N, nullptr, 0, 0, 0, scratch, scratch_bytes, CRNN_BF16, CRNN_BF16, CRNN_F32, nullptr, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K);
}
// Parity mode (fp32 tensors, three-plane products): the same two GEMMs fed by the PRE-BatchNorm depthwise output d [M][K] fp32 -- the staging waves of
// the three-plane kernel apply ReLU6(d * scale[ch] + shift[ch]) (the arithmetic of crnn_bn_act_pool_drop_ex, bit for bit) to the raw items before the
// plane split, so the activated tensor is never written.  Results equal crnn_bn_act_pool_drop_ex + crnn_pwconv_fwd(bf16_products = 2) / crnn_gemm_f32x3
// mode 2 bit for bit.  w [K][N] fp32 (K <= 512); -3 for shapes outside the kernel's rules.
extern "C" int crnn_pwconv_bnrelu6_fwd_f32x3(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K,
                                             float* stat_partials, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  return gemm_bf16_impl(0, d, w, q, (int)M, N, K, K, N, N, nullptr, 0, 0, 0, nullptr, 0, CRNN_F32, CRNN_F32, CRNN_F32, stat_partials, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K, true);
}
// Two-plane forms of the two entry points below / above (hi*hi + hi*mid + mid*hi: 16 significant bits per factor; crnn_gemm_f32x2_bnstats)
extern "C" int crnn_pwconv_bnrelu6_fwd_f32x2(const float* d, const float* in_bnstate, const float* w, float* q, long M, int N, int K,
                                             float* stat_partials, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  return gemm_bf16_impl(0, d, w, q, (int)M, N, K, K, N, N, nullptr, 0, 0, 0, nullptr, 0, CRNN_F32, CRNN_F32, CRNN_F32, stat_partials, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K, true, nullptr, nullptr, 2);
}
extern "C" int crnn_pwconv_bnrelu6_wgrad_f32x2(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K,
                                               float* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  return gemm_bf16_impl(2, d, g, dw, K, N, (int)M, K, N, N, nullptr, 0, 0, 0, scratch, scratch_bytes, CRNN_F32, CRNN_F32, CRNN_F32, nullptr, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K, true, nullptr, nullptr, 2);
}
// ... with the weights as planes (crnn_split3_planes of w, plane stride w_plane_stride elements; null: the entry point above)
extern "C" int crnn_pwconv_bnrelu6_fwd_f32x3_pl(const float* d, const float* in_bnstate, const float* w, const void* w_planes, long w_plane_stride, float* q,
                                                long M, int N, int K, float* stat_partials, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  const X3Planes pl{nullptr, 0, (const unsigned short*)w_planes, w_plane_stride};
  return gemm_bf16_impl(0, d, w, q, (int)M, N, K, K, N, N, nullptr, 0, 0, 0, nullptr, 0, CRNN_F32, CRNN_F32, CRNN_F32, stat_partials, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K, true, nullptr, w_planes ? &pl : nullptr);
}
extern "C" int crnn_pwconv_bnrelu6_wgrad_f32x3(const float* d, const float* in_bnstate, const float* g, float* dw, long M, int N, int K,
                                               float* scratch, size_t scratch_bytes, hipStream_t stream) {
  if (M <= 0 || M > 0x7fffffffL || !in_bnstate) return CRNN_ERR_ARG;
  return gemm_bf16_impl(2, d, g, dw, K, N, (int)M, K, N, N, nullptr, 0, 0, 0, scratch, scratch_bytes, CRNN_F32, CRNN_F32, CRNN_F32, nullptr, stream,
                        nullptr, nullptr, in_bnstate + 2L * K, in_bnstate + 3L * K, true);
}
