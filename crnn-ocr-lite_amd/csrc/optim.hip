// Fused optimizer step over the flat fp32 parameter buffer (train.py:187-190, Keras 2.2.2 formulas):
// global-norm gradient clipping (clipnorm) + Adam (epsilon outside the sqrt, bias correction folded into
// lr_t) or SGD with Nesterov momentum and time-based decay.
// The global norm is a deterministic two-stage reduction and stays on the device (no host sync).
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partials) {
  __shared__ double red[256];
  double a = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) { double v = g[i]; a += v * v; }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
// norm_out[0] = sqrt(sum), norm_out[1] = clip multiplier (clipnorm/norm if norm >= clipnorm else 1)
__global__ __launch_bounds__(256) void norm_finalize_kernel(const double* __restrict__ partials, int nparts, float clipnorm,
                                                            float* __restrict__ norm_out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += partials[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    double nrm = sqrt(red[0]);
    norm_out[0] = (float)nrm;
    norm_out[1] = (clipnorm > 0.f && nrm >= (double)clipnorm) ? (float)((double)clipnorm / nrm) : 1.f;
  }
}

#define NORM_BLOCKS 512
// scratch: NORM_BLOCKS doubles; norm_out: 2 floats (device)
extern "C" int crnn_global_norm(const float* g, long n, float clipnorm, void* scratch, float* norm_out, hipStream_t stream) {
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, g, n, (double*)scratch);
  CRNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double*)scratch, NORM_BLOCKS, clipnorm, norm_out);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// Adam (Keras 2.2.2): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr_t, float b1, float b2, float eps, const float* __restrict__ norm_out) {
  const float cs = norm_out ? norm_out[1] : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * cs;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}
extern "C" int crnn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                              float eps, const float* norm_out, hipStream_t stream) {
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, n, lr_t, beta1, beta2, eps, norm_out);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// SGD (Keras 2.2.2): vel = mom*vel - lr*g ; p += nesterov ? mom*vel - lr*g : vel   (lr already decayed by the host)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ vel, long n, float lr,
                           float mom, int nesterov, const float* __restrict__ norm_out) {
  const float cs = norm_out ? norm_out[1] : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * cs;
    float vi = mom * vel[i] - lr * gi;
    vel[i] = vi;
    p[i] = p[i] + (nesterov ? (mom * vi - lr * gi) : vi);
  }
}
extern "C" int crnn_sgd_step(float* p, const float* g, float* vel, long n, float lr, float momentum, int nesterov,
                             const float* norm_out, hipStream_t stream) {
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, stream, p, g, vel, n, lr, momentum, nesterov, norm_out);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

__global__ void scale_kernel(float* __restrict__ x, long n, float s) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
}
extern "C" int crnn_scale(float* x, long n, float s, hipStream_t stream) {
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, stream, x, n, s);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// bf16 shadow of a flat fp32 buffer (round-to-nearest-even), 4 elements per thread; n % 4 == 0
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) st4(y + 4 * i, ld4(x + 4 * i));
}
extern "C" int crnn_convert_f32_to_bf16(const float* x, void* y, long n, hipStream_t stream) {
  if (n % 4) return CRNN_ERR_ARG;
  int blocks = cdiv(n / 4, 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, (bf16_t*)y, n / 4);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
