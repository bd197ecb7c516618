// Softmax, CTC loss + gradient, greedy and beam CTC decoding -- one 64-lane wavefront per sample.
//   K.ctc_batch_cost(labels, y_pred[:, 2:, :], input_length, label_length)   (utils.py:98-103)
//   K.ctc_decode(greedy=True|False, beam_width, top_paths=1)                 (utils.py:347-357)
// The extended label sequence (S = 2L+1 <= 64) lives one state per lane; the alpha/beta recursions use
// wavefront shuffles for the s-1 / s-2 neighbours, log-space fp32 throughout.
#include "common.h"

#ifndef CRNN_CTC_EXP
#define CRNN_CTC_EXP 0      // timing experiments (scripts/ctc_bench.py): 1 = stop after the log-softmax phase, 2 = after the recursions, 3 = before the log-softmax
#endif
#define CTC_EPS 1e-7f
#define NEG_INF (-INFINITY)

__device__ __forceinline__ float lse2(float a, float b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}
// log(e^a + e^b + e^c) in one go, branch-free (absent terms are -inf: e^-inf = 0): three exponentials and one logarithm on the recursions' dependent chain
// where two nested lse2 took four and two
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  const float mm = (m == NEG_INF) ? 0.f : m;                  // (all three absent: the sum below is 0 and its logarithm -inf)
  return mm + logf(expf(a - mm) + expf(b - mm) + expf(c - mm));
}

// ---- row softmax over C (<= 64) classes: one wave per row -------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ z, float* __restrict__ p, long rows, int C) {
  long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v = lane < C ? z[row * C + lane] : NEG_INF;
  float m = wave_max(v);
  float e = lane < C ? expf(v - m) : 0.f;
  float s = wave_sum(e);
  if (lane < C) p[row * C + lane] = e / s;
}
extern "C" int crnn_softmax_rows(const float* z, float* p, long rows, int C, hipStream_t stream) {
  if (C > 64) return CRNN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, z, p, rows, C);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// dense2's epilogue in one pass (round 5): z [rows][ldz] are the raw products of the streaming GEMM over a padded weight matrix (columns >= C are never read);
// logits = z + bias go out in permuted row order (out row = (m % P) * (rows / P) + m / P: time-major rows back to batch-major, as the tile GEMM's epilogue
// does) together with their softmax p1 (workspace) and p2 (the caller's y_pred, may be NULL) -- the softmax and copy launches of the unfused path.
__global__ __launch_bounds__(256) void softmax_rows_perm_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ bias, float* __restrict__ logits,
                                                                float* __restrict__ p1, float* __restrict__ p2, long rows, int C, int P) {
  const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= rows) return;
  const long orow = P ? (m % P) * (rows / P) + m / P : m;
  const float v = lane < C ? z[m * ldz + lane] + (bias ? bias[lane] : 0.f) : NEG_INF;
  const float mx = wave_max(v);
  const float e = lane < C ? expf(v - mx) : 0.f;
  const float sum = wave_sum(e);
  if (lane < C) {
    logits[orow * C + lane] = v;
    const float pr = e / sum;
    p1[orow * C + lane] = pr;
    if (p2) p2[orow * C + lane] = pr;
  }
}
extern "C" int crnn_softmax_rows_perm(const float* z, int ldz, const float* bias, float* logits, float* p1, float* p2, long rows, int C, int permP,
                                      hipStream_t stream) {
  if (!z || !logits || !p1 || rows <= 0 || C < 1 || ldz < C || permP < 0 || (permP && rows % permP)) return CRNN_ERR_ARG;
  if (C > 64) return CRNN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(softmax_rows_perm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, z, ldz, bias, logits, p1, p2, rows, C, permP);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- CTC loss + gradient w.r.t. the dense2 logits -------------------------------------------------------
// y [B][T][C] softmax (batch-major), labels [B][Lmax] int32, lengths int32.
// loss[b] = -log p(label | y[:, skip:skip+Tb]);  dlogits [T][B][C] TIME-major, = grad_scale * d loss_b / d logits
// (through log(y+eps), TF's internal re-softmax, and the model's softmax); rows outside [skip, skip+Tb) are 0.
// One workgroup of CTC_WAVES wavefronts per sample; the extended label (S = 2L+1 <= 64 states) lives one state per lane:
//   phase 1  all waves : posteriors -> LDS (coalesced), log-softmax of log(y+eps) (a time step per wave, a class per lane)
//   phase 2  wave 0    : alpha recursion (t ascending)   ||   wave 1 : beta recursion (t descending)  -> LDS
//   phase 3  all waves : time steps dealt round-robin to the waves: w_s = exp(alpha+beta-lsm-ll) per state lane, then
//                        class lane k adds the w_s of its states (bit mask, ascending s; the blank class by a wave reduction) and chains to the logits
// Round 5 (scripts/ctc_bench.py, phase-ablation builds): 61 -> 28 us per launch at batch 256 -- the gradient phase was 30 us of it (the blank lane walking
// its 24-state mask through LDS with the wave waiting, two workgroup barriers per time step), the log-softmax phase ran on 50 lanes of 256.
// LDS: lsm [Tb][C], alpha [Tb][64], beta [Tb][64], ys [Tb][C], ab [4][64].
#ifndef CTC_WAVES
#define CTC_WAVES 16     // round 5: 4 -> 16 (phases 1 and 3 deal the time steps over the waves: 41 -> 28 us at batch 256; 8 waves: 31.5)
#endif
__global__ __launch_bounds__(64 * CTC_WAVES) void ctc_loss_grad_kernel(const float* __restrict__ y, const int* __restrict__ labels,
                                                                       const int* __restrict__ input_len, const int* __restrict__ label_len,
                                                                       float* __restrict__ loss, float* __restrict__ dlogits, int B, int T,
                                                                       int C, int Lmax, int skip, float grad_scale) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blank = C - 1;
  const int Tmax = T - skip;
  int Tb = input_len[b]; if (Tb > Tmax) Tb = Tmax; if (Tb < 0) Tb = 0;
  const int L = label_len[b];
  const int S = 2 * L + 1;
  float* lsm = sm;                     // [Tmax][C]
  float* alpha = lsm + Tmax * C;       // [Tmax][64]
  float* beta = alpha + Tmax * 64;     // [Tmax][64]
  float* ys = beta + Tmax * 64;        // [Tmax][C]
  float* ab = ys + Tmax * C;           // [CTC_WAVES][64]
  float& ll_sh = ab[64 * CTC_WAVES];   // log-likelihood, handed from the alpha wave to everyone
  const float* yg = y + ((long)b * T + skip) * C;
  for (int i = tid; i < Tb * C; i += 64 * CTC_WAVES) ys[i] = yg[i];
  // zero gradient rows outside the valid window
  for (int t = wave; t < T; t += CTC_WAVES) {
    bool inside = (t >= skip && t < skip + Tb);
    if (!inside && lane < C) dlogits[((long)t * B + b) * C + lane] = 0.f;
  }
  if (Tb == 0 || S > 64) {             // degenerate sample (uniform across the workgroup)
    if (tid == 0) loss[b] = (Tb == 0 && L == 0) ? 0.f : INFINITY;
    for (int t = skip + wave; t < skip + Tb; t += CTC_WAVES) if (lane < C) dlogits[((long)t * B + b) * C + lane] = 0.f;
    return;
  }
  __syncthreads();
#if CRNN_CTC_EXP == 3
  return;
#endif
  // phase 1: log-softmax of z = log(y + eps): a time step per wave, a class per lane (round 5: was one time step per THREAD -- 50 busy lanes of 256, each
  // evaluating 3 C logarithms and C exponentials in sequence: a fifth of the kernel)
  for (int t = wave; t < Tb; t += CTC_WAVES) {
    const float z = lane < C ? logf(ys[t * C + lane] + CTC_EPS) : NEG_INF;
    const float m = wave_max(z);
    const float e = lane < C ? expf(z - m) : 0.f;
    const float lz = m + logf(wave_sum(e));
    if (lane < C) lsm[t * C + lane] = z - lz;
  }
  __syncthreads();
#if CRNN_CTC_EXP == 1
  return;
#endif
  // extended label of this lane (every wave holds its own copy)
  const int s = lane;
  int ext = blank;
  if (s < S && (s & 1)) ext = labels[(long)b * Lmax + (s >> 1)];
  // phase 2: the two recursions on two waves
  if (wave == 0) {
    int ext2 = __shfl_up(ext, 2, 64);
    const bool can_skip = (s >= 2) && (s < S) && (ext != blank) && (ext != ext2);
    float a = NEG_INF;
    if (s == 0) a = lsm[ext];
    else if (s == 1 && S > 1) a = lsm[ext];
    alpha[s] = a;
    for (int t = 1; t < Tb; ++t) {
      const float em = lsm[t * C + ext];                        // (independent of the chain: issued ahead of it)
      const float a1 = __shfl_up(a, 1, 64), a2 = __shfl_up(a, 2, 64);
      const float v = lse3(a, s >= 1 ? a1 : NEG_INF, can_skip ? a2 : NEG_INF);
      a = (s < S && v != NEG_INF) ? v + em : NEG_INF;
      alpha[t * 64 + s] = a;
    }
    float aL = __shfl(a, S - 1, 64);
    float aL2 = (S > 1) ? __shfl(a, S - 2, 64) : NEG_INF;
    if (lane == 0) ll_sh = lse2(aL, aL2);
  } else if (wave == 1) {
    // beta includes the emission at t, like alpha
    int extn2 = __shfl_down(ext, 2, 64);
    const bool can_skip_b = (s + 2 < S) && (ext != blank) && (ext != extn2);
    float bt = NEG_INF;
    for (int t = Tb - 1; t >= 0; --t) {
      if (t == Tb - 1) {
        bt = (s == S - 1 || (s == S - 2 && S > 1)) ? lsm[t * C + ext] : NEG_INF;
      } else {
        const float em = lsm[t * C + ext];
        const float b1 = __shfl_down(bt, 1, 64), b2 = __shfl_down(bt, 2, 64);
        const float v = lse3(bt, s + 1 < S ? b1 : NEG_INF, can_skip_b ? b2 : NEG_INF);
        bt = (s < S && v != NEG_INF) ? v + em : NEG_INF;
      }
      beta[t * 64 + s] = bt;
    }
  }
  __syncthreads();
#if CRNN_CTC_EXP == 2
  return;
#endif
  const float ll = ll_sh;
  if (tid == 0) loss[b] = -ll;
  if (ll == NEG_INF) {  // no valid path: TF reports inf loss, zero gradient
    for (int t = skip + wave; t < skip + Tb; t += CTC_WAVES) if (lane < C) dlogits[((long)t * B + b) * C + lane] = 0.f;
    return;
  }
  // which extended-label states carry class `lane` (bit s set <=> ext_s == lane).  The BLANK class sits on every even state (L + 1 of them): its lane takes
  // the sum from a wave reduction of the state lanes' own registers instead of walking its mask (round 5: 24 dependent LDS reads per time step on one lane
  // -- with the whole wave waiting for it -- were 30 of the kernel's 55 us; phase timings in scripts/ctc_bench.py)
  unsigned long long occ_mask = 0ull;
  for (int s2 = 0; s2 < S; ++s2) {
    int e2 = __shfl(ext, s2, 64);
    if (e2 == lane && lane != blank) occ_mask |= 1ull << s2;
  }
  // phase 3: gradient, time steps round-robin over the waves.  A wave works on its own ab row only: the hand-over from the state lanes to the class lanes is
  // inside the wave (LDS operations of one wave complete in order), no workgroup barrier.
  float* abw = ab + wave * 64;
  for (int t = wave; t < Tb; t += CTC_WAVES) {
    const float v = (s < S) ? alpha[t * 64 + s] + beta[t * 64 + s] : NEG_INF;   // both contain lsm[t][ext] once
    const float w = (v != NEG_INF) ? expf(v - lsm[t * C + ext] - ll) : 0.f;
    const float wblank = wave_sum((s < S && ext == blank) ? w : 0.f);
    abw[s] = w;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float gyk = 0.f, pk = 0.f;
    if (lane < C) {
      const float l = lsm[t * C + lane];
      float occ = (lane == blank) ? wblank : 0.f;
      for (unsigned long long m = occ_mask; m; m &= m - 1) occ += abw[__ffsll((long long)m) - 1];
      const float gz = expf(l) - occ;
      pk = ys[t * C + lane];
      gyk = gz / (pk + CTC_EPS);          // d loss / d y_pred[t][k]
    }
    const float dot = wave_sum(gyk * pk);
    if (lane < C) dlogits[((long)(t + skip) * B + b) * C + lane] = grad_scale * pk * (gyk - dot);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();       // (the class lanes are done with this row before the state lanes overwrite it)
  }
}

extern "C" int crnn_ctc_loss_grad(const float* y, const int* labels, const int* input_len, const int* label_len, float* loss,
                                  float* dlogits, int B, int T, int C, int Lmax, int skip, float grad_scale, hipStream_t stream) {
  if (C > 64 || C < 2 || T <= skip) return CRNN_ERR_UNSUPPORTED;
  if (Lmax < 0 || 2 * Lmax + 1 > 64) return CRNN_ERR_UNSUPPORTED;   // the extended label (2L+1 states) lives on the 64 lanes of one wavefront
  if (B <= 0) return CRNN_ERR_ARG;
  size_t lds = (2 * (size_t)(T - skip) * C + 2 * (size_t)(T - skip) * 64 + 64 * CTC_WAVES + 4) * sizeof(float);
  if (lds > 160 * 1024) return CRNN_ERR_UNSUPPORTED;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)ctc_loss_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ctc_loss_grad_kernel, dim3(B), dim3(64 * CTC_WAVES), lds, stream, y, labels, input_len, label_len, loss, dlogits, B, T, C, Lmax, skip, grad_scale);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- greedy decode (tf.nn.ctc_greedy_decoder, merge_repeated=True) --------------------------------------
// y [B][T][C]; out [B][T] int32 padded with -1; out_len [B].  argmax = first maximum.
__global__ __launch_bounds__(64) void ctc_greedy_kernel(const float* __restrict__ y, const int* __restrict__ input_len,
                                                        int* __restrict__ out, int* __restrict__ out_len, int T, int C) {
  extern __shared__ int am[];
  const int b = blockIdx.x, lane = threadIdx.x;
  int Tb = input_len ? input_len[b] : T; if (Tb > T) Tb = T;
  const float* yb = y + (long)b * T * C;
  for (int t = lane; t < Tb; t += 64) {
    float m = yb[t * C]; int k0 = 0;
    for (int k = 1; k < C; ++k) { float v = yb[t * C + k]; if (v > m) { m = v; k0 = k; } }
    am[t] = k0;
  }
  for (int t = lane; t < T; t += 64) out[(long)b * T + t] = -1;
  __syncthreads();
  if (lane == 0) {
    int n = 0, prev = -1;
    for (int t = 0; t < Tb; ++t) {
      int k = am[t];
      if (k != C - 1 && k != prev) out[(long)b * T + n++] = k;
      prev = k;
    }
    out_len[b] = n;
  }
}
extern "C" int crnn_ctc_greedy_decode(const float* y, const int* input_len, int* out, int* out_len, int B, int T, int C,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(64), (size_t)T * sizeof(int), stream, y, input_len, out, out_len, T, C);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
