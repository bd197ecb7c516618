// Softmax, CTC loss + gradient, greedy and beam CTC decoding -- one 64-lane wavefront per sample.
//   K.ctc_batch_cost(labels, y_pred[:, 2:, :], input_length, label_length)   (utils.py:98-103)
//   K.ctc_decode(greedy=True|False, beam_width, top_paths=1)                 (utils.py:347-357)
// The extended label sequence (S = 2L+1 <= 64) lives one state per lane; the alpha/beta recursions use
// wavefront shuffles for the s-1 / s-2 neighbours, log-space fp32 throughout.
#include "common.h"

#define CTC_EPS 1e-7f
#define NEG_INF (-INFINITY)

__device__ __forceinline__ float lse2(float a, float b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
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

// ---- CTC loss + gradient w.r.t. the dense2 logits -------------------------------------------------------
// y [B][T][C] softmax (batch-major), labels [B][Lmax] int32, lengths int32.
// loss[b] = -log p(label | y[:, skip:skip+Tb]);  dlogits [T][B][C] TIME-major, = grad_scale * d loss_b / d logits
// (through log(y+eps), TF's internal re-softmax, and the model's softmax); rows outside [skip, skip+Tb) are 0.
// One workgroup of 4 wavefronts per sample; the extended label (S = 2L+1 <= 64 states) lives one state per lane:
//   phase 1  all waves : posteriors -> LDS (coalesced), log-softmax of log(y+eps) (one time step per thread)
//   phase 2  wave 0    : alpha recursion (t ascending)   ||   wave 1 : beta recursion (t descending)  -> LDS
//   phase 3  all waves : time steps dealt round-robin to the waves: w_s = exp(alpha+beta-lsm-ll) per state lane, then
//                        class lane k adds the w_s of its states (bit mask, ascending s) and chains to the logits
// LDS: lsm [Tb][C], alpha [Tb][64], beta [Tb][64], ys [Tb][C], ab [4][64].
#define CTC_WAVES 4
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
  // phase 1: log-softmax of z = log(y + eps), one time step per thread
  for (int t = tid; t < Tb; t += 64 * CTC_WAVES) {
    float m = NEG_INF;
    for (int k = 0; k < C; ++k) m = fmaxf(m, logf(ys[t * C + k] + CTC_EPS));
    float sacc = 0.f;
    for (int k = 0; k < C; ++k) sacc += expf(logf(ys[t * C + k] + CTC_EPS) - m);
    float lz = m + logf(sacc);
    for (int k = 0; k < C; ++k) lsm[t * C + k] = logf(ys[t * C + k] + CTC_EPS) - lz;
  }
  __syncthreads();
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
      float a1 = __shfl_up(a, 1, 64), a2 = __shfl_up(a, 2, 64);
      float v = a;
      if (s >= 1) v = lse2(v, a1);
      if (can_skip) v = lse2(v, a2);
      a = (s < S && v != NEG_INF) ? v + lsm[t * C + ext] : NEG_INF;
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
        float b1 = __shfl_down(bt, 1, 64), b2 = __shfl_down(bt, 2, 64);
        float v = bt;
        if (s + 1 < S) v = lse2(v, b1);
        if (can_skip_b) v = lse2(v, b2);
        bt = (s < S && v != NEG_INF) ? v + lsm[t * C + ext] : NEG_INF;
      }
      beta[t * 64 + s] = bt;
    }
  }
  __syncthreads();
  const float ll = ll_sh;
  if (tid == 0) loss[b] = -ll;
  if (ll == NEG_INF) {  // no valid path: TF reports inf loss, zero gradient
    for (int t = skip + wave; t < skip + Tb; t += CTC_WAVES) if (lane < C) dlogits[((long)t * B + b) * C + lane] = 0.f;
    return;
  }
  // which extended-label states carry class `lane` (bit s set <=> ext_s == lane)
  unsigned long long occ_mask = 0ull;
  for (int s2 = 0; s2 < S; ++s2) {
    int e2 = __shfl(ext, s2, 64);
    if (e2 == lane) occ_mask |= 1ull << s2;
  }
  // phase 3: gradient, time steps round-robin over the waves (each wave has its own ab row)
  float* abw = ab + wave * 64;
  for (int t0 = 0; t0 < Tb; t0 += CTC_WAVES) {
    const int t = t0 + wave;
    const bool act = t < Tb;
    if (act) {
      float v = (s < S) ? alpha[t * 64 + s] + beta[t * 64 + s] : NEG_INF;   // both contain lsm[t][ext] once
      abw[s] = (v != NEG_INF) ? expf(v - lsm[t * C + ext] - ll) : 0.f;
    }
    __syncthreads();
    if (act) {
      float gyk = 0.f, pk = 0.f;
      if (lane < C) {
        float l = lsm[t * C + lane];
        float occ = 0.f;
        for (unsigned long long m = occ_mask; m; m &= m - 1) occ += abw[__ffsll((long long)m) - 1];
        float gz = expf(l) - occ;
        pk = ys[t * C + lane];
        gyk = gz / (pk + CTC_EPS);          // d loss / d y_pred[t][k]
      }
      float dot = wave_sum(gyk * pk);
      if (lane < C) dlogits[((long)(t + skip) * B + b) * C + lane] = grad_scale * pk * (gyk - dot);
    }
    __syncthreads();
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
