// Spatial transformer kernels (utils.py:116-258): plain MaxPool (fwd / first-argmax bwd), 5x5 'valid'
// conv lowering (im2col / col2im; the matmuls run on gemm.hip), and the reference's bilinear grid
// sampler with all of its quirks (x = .5(x+1)*W, truncation, clip-before-weights) fused with the
// ZeroPadding2D((2,2)) that follows it (utils.py:63), plus its gradient w.r.t. theta.
#include "common.h"

// ---- MaxPool2D(ph,pw), NHWC, floor ('valid') ----------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int ph, int pw) {
  int Ho = H / ph, Wo = W / pw;
  long total = (long)B * Ho * Wo * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long pix = i / C;
    int wo = (int)(pix % Wo); long r = pix / Wo; int ho = (int)(r % Ho); long b = r / Ho;
    float m = -INFINITY;
    for (int ii = 0; ii < ph; ++ii)
      for (int j = 0; j < pw; ++j) m = fmaxf(m, x[((b * H + ho * ph + ii) * W + wo * pw + j) * C + c]);
    y[i] = m;
  }
}

// gx[b,h,w,c] = gy[b,h/ph,w/pw,c] if (h,w) is the FIRST maximum of its window (scan order) else 0
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, int B,
                                   int H, int W, int C, int ph, int pw) {
  // 32-bit index arithmetic (the launcher refuses tensors of 2^31 elements or more): 64-bit div/mod chains cost more
  // than the memory traffic of these small localisation-net tensors
  const int Ho = H / ph, Wo = W / pw;
  const int total = B * H * W * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int c = i % C, pix = i / C;
    int w = pix % W, r = pix / W, h = r % H, b = r / H;
    int ho = h / ph, wo = w / pw;
    float out = 0.f;
    if (ho < Ho && wo < Wo) {
      float v = x[i];
      int si = h - ho * ph, sj = w - wo * pw;
      bool is_arg = true;
      for (int ii = 0; ii < ph && is_arg; ++ii)
        for (int j = 0; j < pw; ++j) {
          if (ii == si && j == sj) continue;
          float o = x[((b * H + ho * ph + ii) * W + wo * pw + j) * C + c];
          bool earlier = (ii < si) || (ii == si && j < sj);
          if (earlier ? (o >= v) : (o > v)) { is_arg = false; break; }
        }
      if (is_arg) out = gy[((b * Ho + ho) * Wo + wo) * C + c];
    }
    gx[i] = out;
  }
}

extern "C" int crnn_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int ph, int pw, hipStream_t s) {
  long total = (long)B * (H / ph) * (W / pw) * C;
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, ph, pw);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_maxpool_bwd(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int ph, int pw, hipStream_t s) {
  long total = (long)B * H * W * C;
  if (total >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;   // 32-bit index arithmetic
  int blocks = cdiv(total, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, s, x, gy, gx, B, H, W, C, ph, pw);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- im2col / col2im for a KxK 'valid' stride-1 conv; column = (i*K + j)*C + c (HWIO order) -------
__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int C, int K) {
  const int Ho = H - K + 1, Wo = W - K + 1;
  const int KKC = K * K * C;
  const int total = B * Ho * Wo * KKC;             // < 2^31 (checked by the launcher)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int q = i % KKC, row = i / KKC;
    int c = q % C, ij = q / C, j = ij % K, ii = ij / K;
    int wo = row % Wo, r = row / Wo, ho = r % Ho, b = r / Ho;
    col[i] = x[((b * H + ho + ii) * W + wo + j) * C + c];
  }
}
__global__ void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W, int C, int K) {
  const int Ho = H - K + 1, Wo = W - K + 1;
  const int KKC = K * K * C;
  const int total = B * H * W * C;                 // dcol has < 2^31 elements too (checked by the launcher)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int c = i % C, pix = i / C;
    int w = pix % W, r = pix / W, h = r % H, b = r / H;
    float a = 0.f;
    for (int ii = 0; ii < K; ++ii) {
      int ho = h - ii; if (ho < 0 || ho >= Ho) continue;
      for (int j = 0; j < K; ++j) {
        int wo = w - j; if (wo < 0 || wo >= Wo) continue;
        a += dcol[((b * Ho + ho) * Wo + wo) * KKC + (ii * K + j) * C + c];
      }
    }
    dx[i] = a;
  }
}
extern "C" int crnn_im2col(const float* x, float* col, int B, int H, int W, int C, int K, hipStream_t s) {
  long total = (long)B * (H - K + 1) * (W - K + 1) * K * K * C;
  if (total >= (1L << 31) || (long)B * H * W * C >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;   // 32-bit index arithmetic
  int blocks = cdiv(total, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, s, x, col, B, H, W, C, K);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_col2im(const float* dcol, float* dx, int B, int H, int W, int C, int K, hipStream_t s) {
  long total = (long)B * H * W * C;
  if (total >= (1L << 31) || (long)B * (H - K + 1) * (W - K + 1) * K * K * C >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  int blocks = cdiv(total, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(col2im_kernel, dim3(blocks), dim3(256), 0, s, dcol, dx, B, H, W, C, K);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- bilinear sampler (C = 1, output size == input size, as the model uses it) --------------------
struct SamplePoint { int x0, x1, y0, y1; float x, y; };

__device__ __forceinline__ SamplePoint sample_point(const float* th, int i, int j, int H, int W) {
  // grid: xs = linspace(-1,1,W)[j], ys = linspace(-1,1,H)[i]  (utils.py:209-213)
  float gx = (W > 1) ? (-1.f + (2.f * (float)j) / (float)(W - 1)) : -1.f;
  float gy = (H > 1) ? (-1.f + (2.f * (float)i) / (float)(H - 1)) : -1.f;
  if (j == W - 1 && W > 1) gx = 1.f;
  if (i == H - 1 && H > 1) gy = 1.f;
  float sx = th[0] * gx + th[1] * gy + th[2];
  float sy = th[3] * gx + th[4] * gy + th[5];
  SamplePoint p;
  p.x = (0.5f * (sx + 1.0f)) * (float)W;   // utils.py:150 (scaled by W, not W-1)
  p.y = (0.5f * (sy + 1.0f)) * (float)H;
  // clamp before the int cast so that huge |theta| cannot overflow; equals trunc+clip on the valid range
  float xc = fminf(fmaxf(p.x, -2.f), (float)W + 2.f), yc = fminf(fmaxf(p.y, -2.f), (float)H + 2.f);
  int x0 = (int)xc, y0 = (int)yc;            // truncation toward zero (utils.py:153-156)
  p.x1 = min(max(x0 + 1, 0), W - 1); p.x0 = min(max(x0, 0), W - 1);
  p.y1 = min(max(y0 + 1, 0), H - 1); p.y0 = min(max(y0, 0), H - 1);
  return p;
}

// out [B, H+2*pad, W+2*pad] (zero border), image [B,H,W], theta [B,6]
__global__ void sampler_fwd_kernel(const float* __restrict__ img, const float* __restrict__ theta, float* __restrict__ out,
                                   int B, int H, int W, int pad) {
  int Hp = H + 2 * pad, Wp = W + 2 * pad;
  long total = (long)B * Hp * Wp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int jp = (int)(idx % Wp); long r = idx / Wp; int ip = (int)(r % Hp); int b = (int)(r / Hp);
    int i = ip - pad, j = jp - pad;
    float v = 0.f;
    if (i >= 0 && i < H && j >= 0 && j < W) {
      SamplePoint p = sample_point(theta + b * 6, i, j, H, W);
      const float* im = img + (long)b * H * W;
      float Pa = im[p.y0 * W + p.x0], Pb = im[p.y1 * W + p.x0], Pc = im[p.y0 * W + p.x1], Pd = im[p.y1 * W + p.x1];
      float x0 = (float)p.x0, x1 = (float)p.x1, y0 = (float)p.y0, y1 = (float)p.y1;
      float wa = (x1 - p.x) * (y1 - p.y), wb = (x1 - p.x) * (p.y - y0), wc = (p.x - x0) * (y1 - p.y), wd = (p.x - x0) * (p.y - y0);
      v = ((wa * Pa + wb * Pb) + wc * Pc) + wd * Pd;  // utils.py:201-205 add order
    }
    out[idx] = v;
  }
}

// dtheta[b][6] = sum_i [dx_i * .5W ; dy_i * .5H] * G_i^T ; gout is the padded-map gradient [B,H+2p,W+2p]
__global__ __launch_bounds__(256) void sampler_bwd_kernel(const float* __restrict__ img, const float* __restrict__ theta,
                                                          const float* __restrict__ gout, float* __restrict__ dtheta, int H,
                                                          int W, int pad) {
  __shared__ float red[6][256];
  int b = blockIdx.x, tid = threadIdx.x;
  int Wp = W + 2 * pad, Hp = H + 2 * pad;
  const float* im = img + (long)b * H * W;
  const float* th = theta + b * 6;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int n = tid; n < H * W; n += 256) {
    int i = n / W, j = n % W;
    SamplePoint p = sample_point(th, i, j, H, W);
    float g = gout[((long)b * Hp + i + pad) * Wp + j + pad];
    float Pa = im[p.y0 * W + p.x0], Pb = im[p.y1 * W + p.x0], Pc = im[p.y0 * W + p.x1], Pd = im[p.y1 * W + p.x1];
    float x0 = (float)p.x0, x1 = (float)p.x1, y0 = (float)p.y0, y1 = (float)p.y1;
    float dx = g * (-(y1 - p.y) * Pa - (p.y - y0) * Pb + (y1 - p.y) * Pc + (p.y - y0) * Pd) * (0.5f * (float)W);
    float dy = g * (-(x1 - p.x) * Pa + (x1 - p.x) * Pb - (p.x - x0) * Pc + (p.x - x0) * Pd) * (0.5f * (float)H);
    float gx = (W > 1) ? (-1.f + (2.f * (float)j) / (float)(W - 1)) : -1.f;
    float gy = (H > 1) ? (-1.f + (2.f * (float)i) / (float)(H - 1)) : -1.f;
    acc[0] += dx * gx; acc[1] += dx * gy; acc[2] += dx;
    acc[3] += dy * gx; acc[4] += dy * gy; acc[5] += dy;
  }
  for (int e = 0; e < 6; ++e) red[e][tid] = acc[e];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) for (int e = 0; e < 6; ++e) red[e][tid] += red[e][tid + s];
    __syncthreads();
  }
  if (tid < 6) dtheta[b * 6 + tid] = red[tid][0];
}

extern "C" int crnn_sampler_fwd(const float* img, const float* theta, float* out, int B, int H, int W, int pad, hipStream_t s) {
  long total = (long)B * (H + 2 * pad) * (W + 2 * pad);
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sampler_fwd_kernel, dim3(blocks), dim3(256), 0, s, img, theta, out, B, H, W, pad);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_sampler_bwd(const float* img, const float* theta, const float* gout, float* dtheta, int B, int H, int W,
                                int pad, hipStream_t s) {
  hipLaunchKernelGGL(sampler_bwd_kernel, dim3(B), dim3(256), 0, s, img, theta, gout, dtheta, H, W, pad);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// zero-pad copy (STN disabled): out [B,H+2p,W+2p] <- img [B,H,W]
__global__ void pad_copy_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, int pad) {
  int Hp = H + 2 * pad, Wp = W + 2 * pad;
  long total = (long)B * Hp * Wp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int jp = (int)(idx % Wp); long r = idx / Wp; int ip = (int)(r % Hp); int b = (int)(r / Hp);
    int i = ip - pad, j = jp - pad;
    out[idx] = (i >= 0 && i < H && j >= 0 && j < W) ? img[((long)b * H + i) * W + j] : 0.f;
  }
}
extern "C" int crnn_pad_copy(const float* img, float* out, int B, int H, int W, int pad, hipStream_t s) {
  long total = (long)B * (H + 2 * pad) * (W + 2 * pad);
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pad_copy_kernel, dim3(blocks), dim3(256), 0, s, img, out, B, H, W, pad);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
