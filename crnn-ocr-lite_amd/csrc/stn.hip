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
  // one thread per pooling window and channel: reads the window once, routes gy to its FIRST maximum (scan order) and
  // zeroes the rest; rows / columns beyond the last full window were zeroed by the launcher.  32-bit index arithmetic
  // (the launcher refuses tensors of 2^31 elements or more).
  const int Ho = H / ph, Wo = W / pw;
  const int total = B * Ho * Wo * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int c = i % C, pix = i / C;
    int wo = pix % Wo, r = pix / Wo, ho = r % Ho, b = r / Ho;
    const int base = ((b * H + ho * ph) * W + wo * pw) * C + c;
    float best = x[base]; int arg = 0;
    for (int ii = 0; ii < ph; ++ii)
      for (int j = 0; j < pw; ++j) {
        float v = x[base + (ii * W + j) * C];
        if (v > best) { best = v; arg = ii * pw + j; }          // strict '>' keeps the first maximum
      }
    const float g = gy[i];
    for (int ii = 0; ii < ph; ++ii)
      for (int j = 0; j < pw; ++j) gx[base + (ii * W + j) * C] = (ii * pw + j == arg) ? g : 0.f;
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
  if (H % ph || W % pw) {                                  // elements outside every window get no gradient
    hipError_t e = hipMemsetAsync(gx, 0, (size_t)total * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  const long windows = (long)B * (H / ph) * (W / pw) * C;
  int blocks = cdiv(windows, 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
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

// =====================================================================================================
// Direct kernels for the localisation net (utils.py:248-256): two 5x5 'valid' convs with 20 filters and two small dense
// layers on a 50x16 map.  0.1 % of the step's FLOPs -- through im2col + the generic GEMM they were 4.7 % of its time
// (tiny, unaligned matrices on a handful of workgroups); these kernels keep the weights in LDS and the whole tensors
// in cache.  NHWC, fp32, CO = 20 filters, kernel order HWIO (as Keras stores it).
// =====================================================================================================
#define LOC_CO 20
#define LOC_K 5

// y[b,ho,wo,:] = bias + sum_{i,j,c} x[b,ho+i,wo+j,c] * k[i,j,c,:]      thread = (output pixel, group of 4 filters)
template <int CIN>
__global__ __launch_bounds__(256) void loc_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ k, const float* __restrict__ bias,
                                                           float* __restrict__ y, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ks[LOC_K * LOC_K * CIN * LOC_CO];
  for (int i = threadIdx.x; i < LOC_K * LOC_K * CIN * LOC_CO; i += 256) ks[i] = k[i];
  __syncthreads();
  const int Ho = H - LOC_K + 1, Wo = W - LOC_K + 1;
  const int total = B * Ho * Wo * (LOC_CO / 4);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int g = t % (LOC_CO / 4), pix = t / (LOC_CO / 4);
    const int wo = pix % Wo, r = pix / Wo, ho = r % Ho, b = r / Ho;
    float4 acc = *reinterpret_cast<const float4*>(bias + 4 * g);
    for (int i = 0; i < LOC_K; ++i)
      for (int j = 0; j < LOC_K; ++j) {
        const float* xp = x + ((long)(b * H + ho + i) * W + wo + j) * CIN;
        const float* kp = ks + ((i * LOC_K + j) * CIN) * LOC_CO + 4 * g;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float xv = xp[c];
          const float4 kv = *reinterpret_cast<const float4*>(kp + c * LOC_CO);
          acc.x = fmaf(xv, kv.x, acc.x); acc.y = fmaf(xv, kv.y, acc.y); acc.z = fmaf(xv, kv.z, acc.z); acc.w = fmaf(xv, kv.w, acc.w);
        }
      }
    *reinterpret_cast<float4*>(y + (long)pix * LOC_CO + 4 * g) = acc;
  }
}

// weight / bias gradient partials over a chunk of LOC_WG_PIX output pixels:
//   part[chunk][tap*CIN*20 + c*20 + o] = sum_pix x[pix + tap][c] * gy[pix][o] ;  part[chunk][25*CIN*20 + o] = sum_pix gy[pix][o]
// The chunk's gradients gs[pix][20] and the input values it pairs with xs[pix][XW] are staged in LDS once, then thread
// (e, o) walks the pixels with four independent accumulators.  CIN = 1: one block covers all 25 taps (XW = 25, e = tap);
// CIN = 20: grid.y = tap, XW = 20, e = input channel.
#define LOC_WG_PIX 128
template <int CIN>
__global__ __launch_bounds__(512) void loc_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part,
                                                             int B, int H, int W) {
  constexpr int XW = (CIN == 1) ? LOC_K * LOC_K : CIN;
  constexpr int NOUT = LOC_K * LOC_K * CIN * LOC_CO;
  __shared__ float gs[LOC_WG_PIX * LOC_CO];
  __shared__ float xs[LOC_WG_PIX * XW];
  __shared__ int base[LOC_WG_PIX];
  const int Ho = H - LOC_K + 1, Wo = W - LOC_K + 1, npix = B * Ho * Wo;
  const int p0 = blockIdx.x * LOC_WG_PIX;
  const int np = min(LOC_WG_PIX, npix - p0);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < LOC_WG_PIX * LOC_CO; i += nt) gs[i] = (i < np * LOC_CO) ? gy[(long)p0 * LOC_CO + i] : 0.f;
  if (tid < LOC_WG_PIX) {
    int pix = p0 + (tid < np ? tid : 0);
    int wo = pix % Wo, r = pix / Wo, ho = r % Ho, b = r / Ho;
    base[tid] = ((b * H + ho) * W + wo) * CIN;
  }
  __syncthreads();
  for (int i = tid; i < LOC_WG_PIX * XW; i += nt) {
    const int q = i / XW, e = i % XW;
    const int tap = (CIN == 1) ? e : blockIdx.y, c = (CIN == 1) ? 0 : e;
    xs[i] = (q < np) ? x[base[q] + ((tap / LOC_K) * W + tap % LOC_K) * CIN + c] : 0.f;
  }
  __syncthreads();
  if (tid < XW * LOC_CO) {
    const int e = tid / LOC_CO, o = tid % LOC_CO;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int q = 0; q < LOC_WG_PIX; q += 4) {          // rows beyond np are zero: no tail handling
      a0 = fmaf(xs[q * XW + e], gs[q * LOC_CO + o], a0);
      a1 = fmaf(xs[(q + 1) * XW + e], gs[(q + 1) * LOC_CO + o], a1);
      a2 = fmaf(xs[(q + 2) * XW + e], gs[(q + 2) * LOC_CO + o], a2);
      a3 = fmaf(xs[(q + 3) * XW + e], gs[(q + 3) * LOC_CO + o], a3);
    }
    const int tap = (CIN == 1) ? e : blockIdx.y, c = (CIN == 1) ? 0 : e;
    part[(long)blockIdx.x * (NOUT + LOC_CO) + (tap * CIN + c) * LOC_CO + o] = (a0 + a1) + (a2 + a3);
  }
  // bias gradient: one block row per chunk (the tap-0 block), last 20 of the first 64 idle-or-not threads
  if ((CIN == 1 || blockIdx.y == 0) && tid >= nt - LOC_CO) {
    const int o = tid - (nt - LOC_CO);
    float a0 = 0.f, a1 = 0.f;
    for (int q = 0; q < LOC_WG_PIX; q += 2) { a0 += gs[q * LOC_CO + o]; a1 += gs[(q + 1) * LOC_CO + o]; }
    part[(long)blockIdx.x * (NOUT + LOC_CO) + NOUT + o] = a0 + a1;
  }
}

// dx[b,h,w,c] = sum_{i,j,o} gy[b,h-i,w-j,o] * k[i,j,c,o]      thread = (input pixel, group of 4 input channels); CIN = 20
__global__ __launch_bounds__(256) void loc_conv_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ k, float* __restrict__ dx,
                                                             int B, int H, int W) {
  constexpr int CIN = 20;
  __shared__ __attribute__((aligned(16))) float ks[LOC_K * LOC_K * CIN * LOC_CO];
  for (int i = threadIdx.x; i < LOC_K * LOC_K * CIN * LOC_CO; i += 256) ks[i] = k[i];
  __syncthreads();
  const int Ho = H - LOC_K + 1, Wo = W - LOC_K + 1;
  const int total = B * H * W * (CIN / 4);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int g = t % (CIN / 4), pix = t / (CIN / 4);
    const int w = pix % W, r = pix / W, h = r % H, b = r / H;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < LOC_K; ++i) {
      const int ho = h - i; if (ho < 0 || ho >= Ho) continue;
      for (int j = 0; j < LOC_K; ++j) {
        const int wo = w - j; if (wo < 0 || wo >= Wo) continue;
        const float* gp = gy + ((long)(b * Ho + ho) * Wo + wo) * LOC_CO;
        const float* kp = ks + ((i * LOC_K + j) * CIN + 4 * g) * LOC_CO;
#pragma unroll
        for (int o4 = 0; o4 < LOC_CO / 4; ++o4) {
          const float4 gv = *reinterpret_cast<const float4*>(gp + 4 * o4);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float4 kv = *reinterpret_cast<const float4*>(kp + cc * LOC_CO + 4 * o4);
            acc[cc] = fmaf(gv.x, kv.x, fmaf(gv.y, kv.y, fmaf(gv.z, kv.z, fmaf(gv.w, kv.w, acc[cc]))));
          }
        }
      }
    }
    *reinterpret_cast<float4*>(dx + (long)pix * CIN + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

extern "C" int crnn_loc_conv_fwd(const float* x, const float* k, const float* bias, float* y, int B, int H, int W, int Cin, hipStream_t s) {
  if ((Cin != 1 && Cin != 20) || H < LOC_K || W < LOC_K) return CRNN_ERR_UNSUPPORTED;
  const long total = (long)B * (H - 4) * (W - 4) * (LOC_CO / 4);
  if (total >= (1L << 31) || (long)B * H * W * Cin >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  if (Cin == 1) hipLaunchKernelGGL(loc_conv_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, x, k, bias, y, B, H, W);
  else hipLaunchKernelGGL(loc_conv_fwd_kernel<20>, dim3(blocks), dim3(256), 0, s, x, k, bias, y, B, H, W);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_loc_conv_wgrad_chunks(int B, int H, int W) { return cdiv((long)B * (H - 4) * (W - 4), LOC_WG_PIX); }
// dk [5][5][Cin][20], db [20]; scratch: crnn_loc_conv_wgrad_chunks(B,H,W) * (25*Cin*20 + 20) floats
extern "C" int crnn_loc_conv_wgrad(const float* x, const float* gy, float* dk, float* db, float* scratch, int B, int H, int W, int Cin,
                                   hipStream_t s) {
  if ((Cin != 1 && Cin != 20) || H < LOC_K || W < LOC_K) return CRNN_ERR_UNSUPPORTED;
  if ((long)B * H * W * Cin >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  const int chunks = crnn_loc_conv_wgrad_chunks(B, H, W);
  const int nout = LOC_K * LOC_K * Cin * LOC_CO;
  if (Cin == 1) hipLaunchKernelGGL(loc_conv_wgrad_kernel<1>, dim3(chunks, 1), dim3(512), 0, s, x, gy, scratch, B, H, W);
  else hipLaunchKernelGGL(loc_conv_wgrad_kernel<20>, dim3(chunks, LOC_K * LOC_K), dim3(448), 0, s, x, gy, scratch, B, H, W);
  CRNN_LAUNCH_CHECK();
  // second stage over the chunk rows: the row is [dk | db] so one reduction produces both (they are adjacent in the
  // parameter layout as well: kernel then bias)
  if (db != dk + nout) {
    CRNN_TRY(crnn_partials_sum(scratch, chunks, nout + LOC_CO, scratch + (long)chunks * (nout + LOC_CO), 1.f, s));
    hipError_t e = hipMemcpyAsync(dk, scratch + (long)chunks * (nout + LOC_CO), nout * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(db, scratch + (long)chunks * (nout + LOC_CO) + nout, LOC_CO * sizeof(float), hipMemcpyDeviceToDevice, s);
    return e == hipSuccess ? CRNN_OK : (int)e;
  }
  return crnn_partials_sum(scratch, chunks, nout + LOC_CO, dk, 1.f, s);
}
extern "C" int crnn_loc_conv_dgrad(const float* gy, const float* k, float* dx, int B, int H, int W, hipStream_t s) {
  if (H < LOC_K || W < LOC_K) return CRNN_ERR_UNSUPPORTED;
  const long total = (long)B * H * W * 5;
  if (total >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loc_conv_dgrad_kernel, dim3(blocks), dim3(256), 0, s, gy, k, dx, B, H, W);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- the two dense layers: fc1 = relu(flat W1 + b1) [F -> 50], theta = fc1 W2 + b2 [50 -> 6]  (utils.py:254-255) ----------
#define LOC_H1 50
#define LOC_H2 6
#define LOC_KP 5            // k-parts per hidden unit in the forward: 50 x 5 = 250 threads per image
__global__ __launch_bounds__(256) void loc_fc_fwd_kernel(const float* __restrict__ flat, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ fc1,
                                                         float* __restrict__ theta, int B, int F) {
  // one workgroup per image; thread = (hidden unit j, k-part): the 760-long dot products are latency-bound, so they are
  // cut into 5 interleaved parts of 4 independent chains each and combined through LDS in a fixed order
  __shared__ float ps[LOC_KP][LOC_H1];
  __shared__ float hs[LOC_H1];
  const int img = blockIdx.x, tid = threadIdx.x, part = tid / LOC_H1, j = tid % LOC_H1;
  const float* xr = flat + (long)img * F;
  if (tid < LOC_KP * LOC_H1) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = part;
    for (; k + 3 * LOC_KP < F; k += 4 * LOC_KP) {
      a0 = fmaf(xr[k], w1[(long)k * LOC_H1 + j], a0);
      a1 = fmaf(xr[k + LOC_KP], w1[(long)(k + LOC_KP) * LOC_H1 + j], a1);
      a2 = fmaf(xr[k + 2 * LOC_KP], w1[(long)(k + 2 * LOC_KP) * LOC_H1 + j], a2);
      a3 = fmaf(xr[k + 3 * LOC_KP], w1[(long)(k + 3 * LOC_KP) * LOC_H1 + j], a3);
    }
    for (; k < F; k += LOC_KP) a0 = fmaf(xr[k], w1[(long)k * LOC_H1 + j], a0);
    ps[part][j] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (tid < LOC_H1) {
    float a = b1[tid];
#pragma unroll
    for (int q = 0; q < LOC_KP; ++q) a += ps[q][tid];
    a = fmaxf(a, 0.f);
    hs[tid] = a;
    fc1[(long)img * LOC_H1 + tid] = a;
  }
  __syncthreads();
  if (tid < LOC_H2) {
    float a = b2[tid];
    for (int q = 0; q < LOC_H1; ++q) a = fmaf(hs[q], w2[q * LOC_H2 + tid], a);
    theta[(long)img * LOC_H2 + tid] = a;
  }
}
// per image: dfc1 = (dtheta W2^T) * [fc1 > 0];  dflat = dfc1 W1^T      one workgroup per image
__global__ __launch_bounds__(256) void loc_fc_bwd_data_kernel(const float* __restrict__ dtheta, const float* __restrict__ fc1, const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ dfc1, float* __restrict__ dflat,
                                                              int B, int F) {
  __shared__ float ds[LOC_H1];
  const int img = blockIdx.x, tid = threadIdx.x;
  if (tid < LOC_H1) {
    float a = 0.f;
    for (int o = 0; o < LOC_H2; ++o) a = fmaf(dtheta[(long)img * LOC_H2 + o], w2[tid * LOC_H2 + o], a);
    a = (fc1[(long)img * LOC_H1 + tid] > 0.f) ? a : 0.f;
    ds[tid] = a;
    dfc1[(long)img * LOC_H1 + tid] = a;
  }
  __syncthreads();
  for (int k = tid; k < F; k += 256) {
    const float* wr = w1 + (long)k * LOC_H1;
    float a0 = 0.f, a1 = 0.f;
    for (int q = 0; q < LOC_H1; q += 2) { a0 = fmaf(ds[q], wr[q], a0); a1 = fmaf(ds[q + 1], wr[q + 1], a1); }
    dflat[(long)img * F + k] = a0 + a1;
  }
}
// weight gradients over the batch (fixed order over images: deterministic).  out[r][c] = sum_i L[i][r] * R[i][c] with the
// images staged through LDS in chunks of LOC_IC; a block owns LOC_RB rows r of one of two problems:
//   blocks [0, nb1):  L = flat [B][F] (+ a virtual all-ones row F -> db1), R = dfc1 [B][50]  -> dW1 [F][50], db1
//   last block:       L = fc1 [B][50] (+ ones row -> db2),                R = dtheta [B][6] -> dW2 [50][6], db2
#define LOC_RB 8
#define LOC_IC 64
__global__ __launch_bounds__(256) void loc_fc_bwd_weights_kernel(const float* __restrict__ flat, const float* __restrict__ fc1, const float* __restrict__ dfc1,
                                                                 const float* __restrict__ dtheta, float* __restrict__ dw1, float* __restrict__ db1,
                                                                 float* __restrict__ dw2, float* __restrict__ db2, int B, int F) {
  __shared__ float ls[LOC_IC][LOC_H1 + 1];     // left rows of this block (at most 51 for the dW2 block, 16 for dW1 blocks)
  __shared__ float rs[LOC_IC][LOC_H1];
  const int nb1 = (F + 1 + LOC_RB - 1) / LOC_RB, tid = threadIdx.x;
  const bool second = (int)blockIdx.x == nb1;
  const int r0 = second ? 0 : blockIdx.x * LOC_RB;
  const int nr = second ? LOC_H1 + 1 : min(LOC_RB, F + 1 - r0);     // rows incl. the ones-row
  const int nc = second ? LOC_H2 : LOC_H1;
  const int nlim = second ? LOC_H1 : F;                              // index of the virtual ones-row
  const float* Lm = second ? fc1 : flat; const int ldl = second ? LOC_H1 : F;
  const float* Rm = second ? dtheta : dfc1;
  // outputs of this thread: o = tid, tid + 256, ... < nr * nc  (<= 4 per thread: 8*50 = 400, 51*6 = 306)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = 0; i0 < B; i0 += LOC_IC) {
    const int ni = min(LOC_IC, B - i0);
    __syncthreads();
    for (int t = tid; t < LOC_IC * nr; t += 256) {
      const int i = t / nr, r = t % nr;
      ls[i][r] = (i < ni) ? ((r0 + r < nlim) ? Lm[(long)(i0 + i) * ldl + r0 + r] : 1.f) : 0.f;
    }
    for (int t = tid; t < LOC_IC * nc; t += 256) {
      const int i = t / nc, c = t % nc;
      rs[i][c] = (i < ni) ? Rm[(long)(i0 + i) * nc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = tid + 256 * u;
      if (o < nr * nc) {
        const int r = o / nc, c = o % nc;
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < LOC_IC; i += 2) { a0 = fmaf(ls[i][r], rs[i][c], a0); a1 = fmaf(ls[i + 1][r], rs[i + 1][c], a1); }
        acc[u] += a0 + a1;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = tid + 256 * u;
    if (o < nr * nc) {
      const int r = r0 + o / nc, c = o % nc;
      if (second) { if (r < LOC_H1) dw2[r * LOC_H2 + c] = acc[u]; else db2[c] = acc[u]; }
      else { if (r < F) dw1[(long)r * LOC_H1 + c] = acc[u]; else db1[c] = acc[u]; }
    }
  }
}
extern "C" int crnn_loc_fc_fwd(const float* flat, const float* w1, const float* b1, const float* w2, const float* b2, float* fc1, float* theta,
                               int B, int F, hipStream_t s) {
  if (B <= 0 || F <= 0) return CRNN_ERR_ARG;
  hipLaunchKernelGGL(loc_fc_fwd_kernel, dim3(B), dim3(256), 0, s, flat, w1, b1, w2, b2, fc1, theta, B, F);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_loc_fc_bwd(const float* flat, const float* fc1, const float* dtheta, const float* w1, const float* w2, float* dfc1,
                               float* dflat, float* dw1, float* db1, float* dw2, float* db2, int B, int F, hipStream_t s) {
  if (B <= 0 || F <= 0) return CRNN_ERR_ARG;
  hipLaunchKernelGGL(loc_fc_bwd_data_kernel, dim3(B), dim3(256), 0, s, dtheta, fc1, w1, w2, dfc1, dflat, B, F);
  CRNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(loc_fc_bwd_weights_kernel, dim3(cdiv(F + 1, LOC_RB) + 1), dim3(256), 0, s, flat, fc1, dfc1, dtheta, dw1, db1, dw2, db2, B, F);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
