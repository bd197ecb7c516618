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

// =====================================================================================================
// The localisation net of one sample in ONE workgroup (round 4; utils.py:248-256): MaxPool -> Conv2D(20, 5x5) -> MaxPool -> Conv2D(20, 5x5)
// -> Flatten -> Dense(50, relu) -> Dense(6), everything a sample needs (3 KB of pooled image, 11 KB of pooled features, the 40 KB of the second
// convolution's kernel) in LDS.  The five launches it replaces are 5..15 us each of mostly dispatch and drain for 0.7 MFLOP per sample; the
// arithmetic (fmaf chains, fmaxf scans, the dense layers' interleaved partial sums) is theirs, statement for statement: bit-identical outputs,
// and the saved intermediates (pool1, c1, pool2, flat, fc1) are still written for the backward pass.
// =====================================================================================================
#define LOC_NT 512          // threads of the per-sample forward kernel: two waves per SIMD -- one workgroup per CU has nothing else to hide LDS latency behind
#define LOC_NTB 1024        // ... of the backward kernel: the second convolution's data gradient and weight-gradient term run side by side on 512 threads each
struct LocDims { int H0, W0, Hs1, Ws1, Ho1, Wo1, Hs2, Ws2, Ho2, Wo2, F; };
static LocDims loc_dims(int H0, int W0) {
  LocDims d; d.H0 = H0; d.W0 = W0; d.Hs1 = H0 / 2; d.Ws1 = W0 / 2; d.Ho1 = d.Hs1 - 4; d.Wo1 = d.Ws1 - 4;
  d.Hs2 = d.Ho1 / 2; d.Ws2 = d.Wo1 / 2; d.Ho2 = d.Hs2 - 4; d.Wo2 = d.Ws2 - 4; d.F = d.Ho2 * d.Wo2 * LOC_CO;
  return d;
}
static size_t loc_fwd_lds(const LocDims& d) {
  return sizeof(float) * ((size_t)d.Hs1 * d.Ws1 + (size_t)d.Hs2 * d.Ws2 * LOC_CO + LOC_K * LOC_K * LOC_CO * LOC_CO + LOC_K * LOC_K * LOC_CO + d.F +
                          LOC_KP * LOC_H1 + LOC_H1 + 16);
}
static size_t loc_bwd_lds(const LocDims& d);
__global__ __launch_bounds__(LOC_NT) void loc_net_fwd_kernel(const float* __restrict__ x, const float* __restrict__ k1, const float* __restrict__ bc1,
                                                          const float* __restrict__ k2, const float* __restrict__ bc2, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ pool1, float* __restrict__ c1, float* __restrict__ pool2,
                                                          float* __restrict__ flat, float* __restrict__ fc1, float* __restrict__ theta, LocDims d) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  constexpr int NK2 = LOC_K * LOC_K * LOC_CO * LOC_CO, NK1 = LOC_K * LOC_K * LOC_CO;
  float* const k2s = lsm;                                        // [5][5][20][20]
  float* const k1s = k2s + NK2;                                  // [5][5][1][20]
  float* const p2 = k1s + NK1;                                   // [Hs2][Ws2][20]
  float* const fl = p2 + d.Hs2 * d.Ws2 * LOC_CO;                 // [F]
  float* const ps = fl + d.F;                                    // [LOC_KP][LOC_H1]
  float* const hs = ps + LOC_KP * LOC_H1;                        // [LOC_H1]
  float* const p1 = hs + LOC_H1 + 2;                             // [Hs1][Ws1]   (+2: keeps p1 off the float4-aligned part; scalar reads only)
  const int img = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < NK2 / 4; i += LOC_NT) reinterpret_cast<float4*>(k2s)[i] = reinterpret_cast<const float4*>(k2)[i];
  for (int i = tid; i < NK1 / 4; i += LOC_NT) reinterpret_cast<float4*>(k1s)[i] = reinterpret_cast<const float4*>(k1)[i];
  // ---- MaxPool2D(2,2) of the image (maxpool_fwd_kernel's scan)
  const float* xi = x + (long)img * d.H0 * d.W0;
  for (int i = tid; i < d.Hs1 * d.Ws1; i += LOC_NT) {
    const int w = i % d.Ws1, h = i / d.Ws1;
    float m = -INFINITY;
    for (int ii = 0; ii < 2; ++ii)
      for (int j = 0; j < 2; ++j) m = fmaxf(m, xi[(2 * h + ii) * d.W0 + 2 * w + j]);
    p1[i] = m;
    pool1[(long)img * d.Hs1 * d.Ws1 + i] = m;
  }
  __syncthreads();
  // ---- Conv2D(20, 5x5, valid) on the pooled image + MaxPool2D(2,2): a thread owns one pooling window and 4 filters (loc_conv_fwd_kernel<1>'s chain)
  for (int t = tid; t < d.Hs2 * d.Ws2 * (LOC_CO / 4); t += LOC_NT) {
    const int g = t % (LOC_CO / 4), pix = t / (LOC_CO / 4);
    const int pw = pix % d.Ws2, ph = pix / d.Ws2;
    const float4 bias = *reinterpret_cast<const float4*>(bc1 + 4 * g);
    float4 acc[4] = {bias, bias, bias, bias};
    for (int i = 0; i < LOC_K; ++i)
      for (int j = 0; j < LOC_K; ++j) {
        const float4 kv = *reinterpret_cast<const float4*>(k1s + (i * LOC_K + j) * LOC_CO + 4 * g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xv = p1[(2 * ph + (q >> 1) + i) * d.Ws1 + 2 * pw + (q & 1) + j];
          acc[q].x = fmaf(xv, kv.x, acc[q].x); acc[q].y = fmaf(xv, kv.y, acc[q].y); acc[q].z = fmaf(xv, kv.z, acc[q].z); acc[q].w = fmaf(xv, kv.w, acc[q].w);
        }
      }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(c1 + (((long)img * d.Ho1 + 2 * ph + (q >> 1)) * d.Wo1 + 2 * pw + (q & 1)) * LOC_CO + 4 * g) = acc[q];
      m.x = fmaxf(m.x, acc[q].x); m.y = fmaxf(m.y, acc[q].y); m.z = fmaxf(m.z, acc[q].z); m.w = fmaxf(m.w, acc[q].w);
    }
    *reinterpret_cast<float4*>(p2 + pix * LOC_CO + 4 * g) = m;
    *reinterpret_cast<float4*>(pool2 + ((long)img * d.Hs2 * d.Ws2 + pix) * LOC_CO + 4 * g) = m;
  }
  __syncthreads();
  // ---- Conv2D(20, 5x5, valid) on the pooled features = the flattened input of the dense layers (loc_conv_fwd_kernel<20>'s chain)
  for (int t = tid; t < d.Ho2 * d.Wo2 * (LOC_CO / 4); t += LOC_NT) {
    const int g = t % (LOC_CO / 4), pix = t / (LOC_CO / 4);
    const int wo = pix % d.Wo2, ho = pix / d.Wo2;
    float4 acc = *reinterpret_cast<const float4*>(bc2 + 4 * g);
    for (int i = 0; i < LOC_K; ++i)
      for (int j = 0; j < LOC_K; ++j) {
        const float* xp = p2 + ((ho + i) * d.Ws2 + wo + j) * LOC_CO;
        const float* kp = k2s + ((i * LOC_K + j) * LOC_CO) * LOC_CO + 4 * g;
#pragma unroll
        for (int c = 0; c < LOC_CO; ++c) {
          const float xv = xp[c];
          const float4 kv = *reinterpret_cast<const float4*>(kp + c * LOC_CO);
          acc.x = fmaf(xv, kv.x, acc.x); acc.y = fmaf(xv, kv.y, acc.y); acc.z = fmaf(xv, kv.z, acc.z); acc.w = fmaf(xv, kv.w, acc.w);
        }
      }
    *reinterpret_cast<float4*>(fl + pix * LOC_CO + 4 * g) = acc;
    *reinterpret_cast<float4*>(flat + (long)img * d.F + pix * LOC_CO + 4 * g) = acc;
  }
  __syncthreads();
  // ---- Dense(50, relu), Dense(6) (loc_fc_fwd_kernel's partial sums)
  const int F = d.F, part = tid / LOC_H1, j = tid % LOC_H1;
  if (tid < LOC_KP * LOC_H1) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int k = part;
#pragma unroll 8   // (one workgroup per CU: nothing else hides the weight loads' latency -- 32 of them in flight per thread)
    for (; k + 3 * LOC_KP < F; k += 4 * LOC_KP) {
      a0 = fmaf(fl[k], w1[(long)k * LOC_H1 + j], a0);
      a1 = fmaf(fl[k + LOC_KP], w1[(long)(k + LOC_KP) * LOC_H1 + j], a1);
      a2 = fmaf(fl[k + 2 * LOC_KP], w1[(long)(k + 2 * LOC_KP) * LOC_H1 + j], a2);
      a3 = fmaf(fl[k + 3 * LOC_KP], w1[(long)(k + 3 * LOC_KP) * LOC_H1 + j], a3);
    }
    for (; k < F; k += LOC_KP) a0 = fmaf(fl[k], w1[(long)k * LOC_H1 + j], a0);
    ps[part * LOC_H1 + j] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (tid < LOC_H1) {
    float a = b1[tid];
#pragma unroll
    for (int q = 0; q < LOC_KP; ++q) a += ps[q * LOC_H1 + tid];
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
// CRNN_OK when the fused localisation-net kernels take an H0 x W0 image: whole pooling windows over the first convolution's map, everything in LDS
extern "C" int crnn_loc_net_fused_supported(int H0, int W0) {
  if (H0 < 2 || W0 < 2) return CRNN_ERR_UNSUPPORTED;
  const LocDims d = loc_dims(H0, W0);
  if (d.Ho1 < 2 || d.Wo1 < 2 || (d.Ho1 & 1) || (d.Wo1 & 1) || d.Ho2 < 1 || d.Wo2 < 1) return CRNN_ERR_UNSUPPORTED;
  return (loc_fwd_lds(d) <= 120 * 1024 && loc_bwd_lds(d) <= 150 * 1024) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
// x [B][H0][W0] -> pool1 [B][H0/2][W0/2], c1 [B][Ho1][Wo1][20], pool2 [B][Hs2][Ws2][20], flat [B][F], fc1 [B][50], theta [B][6]: the outputs of
// crnn_maxpool_fwd + crnn_loc_conv_fwd + crnn_maxpool_fwd + crnn_loc_conv_fwd + crnn_loc_fc_fwd, bit for bit, in one launch.
extern "C" int crnn_loc_net_fwd(const float* x, const float* k1, const float* bc1, const float* k2, const float* bc2, const float* w1, const float* b1,
                                const float* w2, const float* b2, float* pool1, float* c1, float* pool2, float* flat, float* fc1, float* theta,
                                int B, int H0, int W0, hipStream_t s) {
  if (!x || !k1 || !bc1 || !k2 || !bc2 || !w1 || !b1 || !w2 || !b2 || !pool1 || !c1 || !pool2 || !flat || !fc1 || !theta || B <= 0) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_loc_net_fused_supported(H0, W0));
  if ((((uintptr_t)k1 | (uintptr_t)k2 | (uintptr_t)bc1 | (uintptr_t)bc2 | (uintptr_t)c1 | (uintptr_t)pool2 | (uintptr_t)flat) & 15)) return CRNN_ERR_UNSUPPORTED;
  const LocDims d = loc_dims(H0, W0);
  const size_t lds = loc_fwd_lds(d);
  CRNN_LDS_ATTR(loc_net_fwd_kernel, 120 * 1024);
  hipLaunchKernelGGL(loc_net_fwd_kernel, dim3(B), dim3(LOC_NT), lds, s, x, k1, bc1, k2, bc2, w1, b1, w2, b2, pool1, c1, pool2, flat, fc1, theta, d);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---- backward of the localisation net: one workgroup per sample (data path + that sample's convolution weight-gradient terms), then one
// launch that sums the samples' terms in a fixed order and forms the dense layers' weight gradients over the batch --------------------------------
//   per sample:  dfc1 = (dtheta W2^T) [fc1 > 0];  dflat = dfc1 W1^T;  dpool2 = conv_dgrad(dflat, k2);  dc1 = MaxPool backward (first maximum);
//                terms [dk2 10000 | db2 20 | dk1 500 | db1 20]: dk2[tap][c][o] = sum_pix pool2[pix + tap][c] dflat[pix][o], dk1[tap][o] = sum pool1[.] dc1[.]
// Nothing but dfc1 and the terms goes to HBM (the stand-alone kernels wrote dflat, dpool2 and the mostly-zero dc1 and read them back).
#define LOC_TERMS (LOC_K * LOC_K * LOC_CO * LOC_CO + LOC_CO + LOC_K * LOC_K * LOC_CO + LOC_CO)
static size_t loc_bwd_lds(const LocDims& d) {
  const size_t np2 = (size_t)d.Hs2 * d.Ws2 * LOC_CO;
  return sizeof(float) * (LOC_K * LOC_K * LOC_CO * LOC_CO + 2 * np2 + d.F + 64 + (size_t)d.Hs1 * d.Ws1 + 16) + ((np2 + 15) & ~(size_t)15) +
         sizeof(int) * ((size_t)d.Ho2 * d.Wo2 + 4);   // k2 | pool2 | dpool2 | dflat | dfc1 | pool1 | arg bytes | pixel offsets
}
#ifdef CRNN_LOC_TRACE   // timing build: s_memrealtime stamps of workgroup 0 after every phase, behind the samples' terms
#define LOC_TRC(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(terms + (long)gridDim.x * LOC_TERMS)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LOC_TRC(i) do {} while (0)
#endif
__global__ __launch_bounds__(LOC_NTB) void loc_net_bwd_kernel(const float* __restrict__ dtheta, const float* __restrict__ fc1, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, const float* __restrict__ k2, const float* __restrict__ pool1,
                                                          const float* __restrict__ c1, const float* __restrict__ pool2, float* __restrict__ dfc1,
                                                          float* __restrict__ terms, LocDims d) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  constexpr int NK2 = LOC_K * LOC_K * LOC_CO * LOC_CO;
  const int np2 = d.Hs2 * d.Ws2 * LOC_CO, F = d.F;
  float* const k2s = lsm;                       // [5][5][20][20]
  float* const p2 = k2s + NK2;                  // pool2 of the sample [Hs2][Ws2][20]
  float* const dp2 = p2 + np2;                  // dpool2
  float* const df = dp2 + np2;                  // dflat [Ho2][Wo2][20]
  float* const ds = df + F;                     // dfc1 [50] (64 reserved)
  float* const p1 = ds + 64;                    // pool1 of the sample [Hs1][Ws1]
  unsigned char* const arg = reinterpret_cast<unsigned char*>(p1 + d.Hs1 * d.Ws1 + 16);   // first-maximum position (0..3) of every pooling window x channel
  int* const xo = reinterpret_cast<int*>(arg + ((np2 + 15) & ~15));   // offset of output pixel (ho, wo) of the second convolution inside pool2: (ho Ws2 + wo) 20
  const int img = blockIdx.x, tid = threadIdx.x;
  float* const tm = terms + (long)img * LOC_TERMS;
  LOC_TRC(0);
  for (int i = tid; i < d.Ho2 * d.Wo2 + 4; i += LOC_NTB) xo[i] = (i < d.Ho2 * d.Wo2) ? ((i / d.Wo2) * d.Ws2 + i % d.Wo2) * LOC_CO : 0;
  for (int i = tid; i < NK2 / 4; i += LOC_NTB) reinterpret_cast<float4*>(k2s)[i] = reinterpret_cast<const float4*>(k2)[i];
  for (int i = tid; i < np2 / 4; i += LOC_NTB) reinterpret_cast<float4*>(p2)[i] = reinterpret_cast<const float4*>(pool2 + (long)img * np2)[i];
  for (int i = tid; i < d.Hs1 * d.Ws1; i += LOC_NTB) p1[i] = pool1[(long)img * d.Hs1 * d.Ws1 + i];
  // first maximum of every 2x2 window of c1 (maxpool_bwd_kernel's scan: strict '>' keeps the first)
  for (int i = tid; i < np2; i += LOC_NTB) {
    const int c = i % LOC_CO, pix = i / LOC_CO, pw = pix % d.Ws2, ph = pix / d.Ws2;
    const float* cb = c1 + (((long)img * d.Ho1 + 2 * ph) * d.Wo1 + 2 * pw) * LOC_CO + c;
    float best = cb[0]; int a = 0;
    for (int ii = 0; ii < 2; ++ii)
      for (int j = 0; j < 2; ++j) {
        const float v = cb[(ii * d.Wo1 + j) * LOC_CO];
        if (v > best) { best = v; a = ii * 2 + j; }
      }
    arg[i] = (unsigned char)a;
  }
  // ---- dense layers backwards (loc_fc_bwd_data_kernel)
  if (tid < LOC_H1) {
    float a = 0.f;
    for (int o = 0; o < LOC_H2; ++o) a = fmaf(dtheta[(long)img * LOC_H2 + o], w2[tid * LOC_H2 + o], a);
    a = (fc1[(long)img * LOC_H1 + tid] > 0.f) ? a : 0.f;
    ds[tid] = a;
    dfc1[(long)img * LOC_H1 + tid] = a;
  }
  __syncthreads();
  LOC_TRC(1);
  for (int k = tid; k < F; k += LOC_NTB) {
    const float2* wr = reinterpret_cast<const float2*>(w1 + (long)k * LOC_H1);    // the row's 25 loads issued together (same sums, same order)
    float2 wv[LOC_H1 / 2];
#pragma unroll
    for (int q = 0; q < LOC_H1 / 2; ++q) wv[q] = wr[q];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < LOC_H1 / 2; ++q) { a0 = fmaf(ds[2 * q], wv[q].x, a0); a1 = fmaf(ds[2 * q + 1], wv[q].y, a1); }
    df[k] = a0 + a1;
  }
  __syncthreads();
  LOC_TRC(2);
  // ---- second convolution: data gradient (loc_conv_dgrad_kernel's chain) into LDS ...
  // (threads 0..511: the data gradient; threads 512..1011: the weight-gradient term below -- independent work side by side)
  for (int t = tid; t < d.Hs2 * d.Ws2 * (LOC_CO / 4) && tid < 512; t += 512) {
    const int g = t % (LOC_CO / 4), pix = t / (LOC_CO / 4);
    const int w = pix % d.Ws2, h = pix / d.Ws2;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < LOC_K; ++i) {
      const int ho = h - i; if (ho < 0 || ho >= d.Ho2) continue;
      for (int j = 0; j < LOC_K; ++j) {
        const int wo = w - j; if (wo < 0 || wo >= d.Wo2) continue;
        const float* gp = df + (ho * d.Wo2 + wo) * LOC_CO;
        const float* kp = k2s + ((i * LOC_K + j) * LOC_CO + 4 * g) * LOC_CO;
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
    *reinterpret_cast<float4*>(dp2 + pix * LOC_CO + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  // ... and this sample's term of its weight gradient: thread = (half of the taps, input channel c, 4 filters)
  if (tid >= 512 && tid < 512 + LOC_K * LOC_CO * (LOC_CO / 4)) {  // 500 threads: (tap row, input channel, 4 filters)
    const int t5 = tid - 512;
    const int og = t5 % (LOC_CO / 4), c = (t5 / (LOC_CO / 4)) % LOC_CO, tg = t5 / (LOC_CO * (LOC_CO / 4));
    // the five taps of kernel row tg share a pixel's gradient: one read of it feeds 20 fmas (a tap at a time read it once per 4)
    const int npx = d.Ho2 * d.Wo2;
    const float* xb = p2 + (tg * d.Ws2) * LOC_CO + c;
    float4 a[LOC_K];
#pragma unroll
    for (int j = 0; j < LOC_K; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pix = 0; pix < npx; pix += 2) {              // two pixels per trip (a pixel past the end: gradient 0 -> the sums unchanged)
      float xv[2][LOC_K]; float4 gv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool in = pix + u < npx;
        const float* xp = xb + xo[pix + u];
        gv[u] = in ? *reinterpret_cast<const float4*>(df + (pix + u) * LOC_CO + 4 * og) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) xv[u][j] = xp[j * LOC_CO];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) {
          a[j].x = fmaf(xv[u][j], gv[u].x, a[j].x); a[j].y = fmaf(xv[u][j], gv[u].y, a[j].y);
          a[j].z = fmaf(xv[u][j], gv[u].z, a[j].z); a[j].w = fmaf(xv[u][j], gv[u].w, a[j].w);
        }
    }
#pragma unroll
    for (int j = 0; j < LOC_K; ++j) *reinterpret_cast<float4*>(tm + ((tg * LOC_K + j) * LOC_CO + c) * LOC_CO + 4 * og) = a[j];
  }
  __syncthreads();
  LOC_TRC(3);
  // ---- first convolution: dc1 is dpool2 at the first maximum of each window, zero elsewhere -- its weight-gradient term straight from the windows
  float* const t1p = tm + NK2 + LOC_CO;
  float* const qs = k2s;                                           // (the kernel of the second convolution is not needed any more) [4][125][4] partial sums
  constexpr int NT1 = LOC_K * LOC_K * (LOC_CO / 4);                // 125 (tap, 4 filters) pairs, each over four quarters of the windows
  const int nwin = d.Hs2 * d.Ws2, wq = (nwin + 3) / 4;
  if (tid < 4 * NT1) {
    const int qd = tid / NT1, t = tid % NT1;
    const int og = t % (LOC_CO / 4), tap = t / (LOC_CO / 4), i = tap / LOC_K, j = tap % LOC_K;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int pend = min(nwin, (qd + 1) * wq);
    for (int pix = qd * wq; pix < pend; pix += 2) {       // two windows per trip: both rounds of LDS reads (positions, then the pooled image) in flight together
      float g4[2][4]; float xv[2][4];
      unsigned am[2]; int pb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int px = min(pix + u, nwin - 1);
        const float4 gv = *reinterpret_cast<const float4*>(dp2 + px * LOC_CO + 4 * og);
        const bool in = pix + u < pend;
        g4[u][0] = in ? gv.x : 0.f; g4[u][1] = in ? gv.y : 0.f; g4[u][2] = in ? gv.z : 0.f; g4[u][3] = in ? gv.w : 0.f;
        am[u] = *reinterpret_cast<const unsigned*>(arg + px * LOC_CO + 4 * og);
        pb[u] = (2 * (px / d.Ws2) + i) * d.Ws1 + 2 * (px % d.Ws2) + j;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ar = (am[u] >> (8 * q)) & 3;
          xv[u][q] = p1[pb[u] + (ar >> 1) * d.Ws1 + (ar & 1)];
        }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = fmaf(xv[u][q], g4[u][q], a[q]);
    }
    *reinterpret_cast<float4*>(qs + (qd * NT1 + t) * 4) = make_float4(a[0], a[1], a[2], a[3]);
  }
  __syncthreads();
  LOC_TRC(4);
  if (tid < NT1) {                                                 // the four quarters in a fixed order
    const int og = tid % (LOC_CO / 4), tap = tid / (LOC_CO / 4);
    float4 v = *reinterpret_cast<const float4*>(qs + tid * 4);
#pragma unroll
    for (int qd = 1; qd < 4; ++qd) {
      const float4 u = *reinterpret_cast<const float4*>(qs + (qd * NT1 + tid) * 4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(t1p + tap * LOC_CO + 4 * og) = v;
  } else if (tid >= 128 && tid < 128 + LOC_CO) {                  // its bias: sum of dc1 = sum of dpool2
    const int o = tid - 128;
    float a = 0.f;
    for (int pix = 0; pix < d.Hs2 * d.Ws2; ++pix) a += dp2[pix * LOC_CO + o];
    t1p[LOC_K * LOC_K * LOC_CO + o] = a;
  } else if (tid >= 192 && tid < 192 + LOC_CO) {                  // bias of the second convolution: sum of dflat over the pixels
    const int o = tid - 192;
    float a = 0.f;
    for (int pix = 0; pix < d.Ho2 * d.Wo2; ++pix) a += df[pix * LOC_CO + o];
    tm[NK2 + o] = a;
  }
  LOC_TRC(5);
}
// Second launch: blocks [0, nred): the samples' terms summed in sample order (4 interleaved chains) -> dk2 | db2c | dk1 | db1c;
// blocks [nred, nred + nw1): rows of dW1 (+ db1) = flat^T dfc1 over the batch; last block: dW2, db2 = fc1^T dtheta.
#define LOC_WROWS 8
#define LOC_WIMG 128
__global__ __launch_bounds__(256) void loc_net_bwd_reduce_kernel(const float* __restrict__ terms, const float* __restrict__ flat, const float* __restrict__ fc1,
                                                                 const float* __restrict__ dfc1, const float* __restrict__ dtheta, float* __restrict__ dk2,
                                                                 float* __restrict__ dbc2, float* __restrict__ dk1, float* __restrict__ dbc1,
                                                                 float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                                 float* __restrict__ db2, int B, int F, int nred, int nw1) {
  constexpr int NK2 = LOC_K * LOC_K * LOC_CO * LOC_CO, NK1 = LOC_K * LOC_K * LOC_CO;
  __shared__ __attribute__((aligned(16))) float ls[LOC_WIMG][LOC_H1];   // left operand of a chunk: 8 columns of flat (dW1 blocks) | fc1 (the dW2 block)
  __shared__ __attribute__((aligned(16))) float rs[LOC_WIMG][LOC_H1];   // right operand: dfc1 | dtheta
  const int tid = threadIdx.x, bid = blockIdx.x;
  if (bid < nred) {
    const int o = bid * 256 + tid;
    if (o >= LOC_TERMS) return;
    float a[16];                                   // 16 interleaved chains: 16 loads in flight per thread, combined in a fixed tree
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.f;
    int i = 0;
    for (; i + 15 < B; i += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += terms[(long)(i + u) * LOC_TERMS + o];
    }
    for (; i < B; ++i) a[0] += terms[(long)i * LOC_TERMS + o];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int u = 0; u < w; ++u) a[u] += a[u + w];
    const float v = a[0];
    if (o < NK2) dk2[o] = v;
    else if (o < NK2 + LOC_CO) dbc2[o - NK2] = v;
    else if (o < NK2 + LOC_CO + NK1) dk1[o - NK2 - LOC_CO] = v;
    else dbc1[o - NK2 - LOC_CO - NK1] = v;
    return;
  }
  // dense layers' weight gradients over the batch, images staged through LDS in chunks of LOC_WIMG (contiguous float4 copies where the source is contiguous):
  //   blocks nred .. nred + nw1 - 1: rows r0 .. r0 + 7 of dW1 = flat^T dfc1 (the first of them also db1 = column sums of dfc1);
  //   last block: dW2 = fc1^T dtheta and db2 = column sums of dtheta
  const bool second = bid == nred + nw1;
  const int r0 = second ? 0 : (bid - nred) * LOC_WROWS;
  const int nlr = second ? LOC_H1 : LOC_WROWS;                       // staged left columns per image
  const int nc = second ? LOC_H2 : LOC_H1;
  const float* Rm = second ? dtheta : dfc1;
  float* const lsf = &ls[0][0]; float* const rsf = &rs[0][0];        // flat views: left [LOC_WIMG][nlr], right [LOC_WIMG][nc]
  const int o0 = tid, o1 = tid + 256, nout = nlr * nc;              // this thread's outputs (8 * 50 = 400 | 50 * 6 = 300)
  const bool v0 = o0 < nout, v1 = o1 < nout;
  const int ra = v0 ? o0 / nc : 0, ca = v0 ? o0 % nc : 0, rb = v1 ? o1 / nc : 0, cb = v1 ? o1 % nc : 0;
  float acc0 = 0.f, acc1 = 0.f, accb = 0.f;
  for (int i0 = 0; i0 < B; i0 += LOC_WIMG) {
    const int ni = min(LOC_WIMG, B - i0);
    __syncthreads();
    // contiguous source -> float4 copy, zero past the batch (the sources are 16-byte aligned: chunks start at multiples of 128 images)
    auto stage = [&](float* dst, const float* src, int nvalid, int ntotal) {
#pragma unroll 4
      for (int t4 = tid; t4 < ntotal / 4; t4 += 256) {
        const int e = 4 * t4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e + 3 < nvalid) v = *reinterpret_cast<const float4*>(src + e);
        else { if (e < nvalid) v.x = src[e]; if (e + 1 < nvalid) v.y = src[e + 1]; if (e + 2 < nvalid) v.z = src[e + 2]; }
        reinterpret_cast<float4*>(dst)[t4] = v;
      }
    };
    if (second) {
      stage(lsf, fc1 + (long)i0 * LOC_H1, ni * LOC_H1, LOC_WIMG * LOC_H1);
    } else {
      for (int t = tid; t < LOC_WIMG * LOC_WROWS; t += 256) {
        const int i = t / LOC_WROWS, r = t % LOC_WROWS;
        lsf[t] = (i < ni && r0 + r < F) ? flat[(long)(i0 + i) * F + r0 + r] : 0.f;
      }
    }
    stage(rsf, Rm + (long)i0 * nc, ni * nc, LOC_WIMG * nc);
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll 4
    for (int i = 0; i < LOC_WIMG; i += 2) {          // both outputs side by side, eight images per trip: the LDS reads of a trip are in flight together
      a0 = fmaf(lsf[i * nlr + ra], rsf[i * nc + ca], a0); a1 = fmaf(lsf[(i + 1) * nlr + ra], rsf[(i + 1) * nc + ca], a1);
      b0 = fmaf(lsf[i * nlr + rb], rsf[i * nc + cb], b0); b1 = fmaf(lsf[(i + 1) * nlr + rb], rsf[(i + 1) * nc + cb], b1);
    }
    acc0 += a0 + a1; acc1 += b0 + b1;
    if (tid < nc) {                                  // column sums of the right operand (rows past ni are zero)
      float s0 = 0.f, s1 = 0.f;
      for (int i = 0; i < LOC_WIMG; i += 2) { s0 += rsf[i * nc + tid]; s1 += rsf[(i + 1) * nc + tid]; }
      accb += s0 + s1;
    }
  }
  if (second) {
    if (v0) dw2[o0] = acc0;
    if (v1) dw2[o1] = acc1;
    if (tid < LOC_H2) db2[tid] = accb;
  } else {
    if (v0 && r0 + ra < F) dw1[(long)(r0 + ra) * LOC_H1 + ca] = acc0;
    if (v1 && r0 + rb < F) dw1[(long)(r0 + rb) * LOC_H1 + cb] = acc1;
    if (bid == nred && tid < LOC_H1) db1[tid] = accb;
  }
}
// floats of scratch crnn_loc_net_bwd needs (the samples' convolution weight-gradient terms)
extern "C" long crnn_loc_net_bwd_scratch(int B) { return (long)B * LOC_TERMS; }
// Backward of crnn_loc_net_fwd from dtheta [B][6]: every gradient of the eight localisation-net parameters (dk1 [5][5][1][20], dbc1 [20], dk2 [5][5][20][20],
// dbc2 [20], dw1 [F][50], db1 [50], dw2 [50][6], db2 [6]); dfc1 [B][50] and `scratch` (crnn_loc_net_bwd_scratch floats) are work buffers.  The sums of
// crnn_loc_fc_bwd + crnn_loc_conv_wgrad + crnn_loc_conv_dgrad + crnn_maxpool_bwd + crnn_loc_conv_wgrad in another order (fixed: deterministic).
extern "C" int crnn_loc_net_bwd(const float* dtheta, const float* flat, const float* fc1, const float* pool1, const float* c1, const float* pool2,
                                const float* w1, const float* w2, const float* k2, float* dfc1, float* scratch, float* dk1, float* dbc1, float* dk2,
                                float* dbc2, float* dw1, float* db1, float* dw2, float* db2, int B, int H0, int W0, hipStream_t s) {
  if (!dtheta || !flat || !fc1 || !pool1 || !c1 || !pool2 || !w1 || !w2 || !k2 || !dfc1 || !scratch || !dk1 || !dbc1 || !dk2 || !dbc2 || !dw1 || !db1 ||
      !dw2 || !db2 || B <= 0) return CRNN_ERR_ARG;
  CRNN_TRY(crnn_loc_net_fused_supported(H0, W0));
  if ((((uintptr_t)k2 | (uintptr_t)pool2 | (uintptr_t)scratch | (uintptr_t)dfc1 | (uintptr_t)fc1 | (uintptr_t)dtheta) & 15) || ((uintptr_t)w1 & 7)) return CRNN_ERR_UNSUPPORTED;
  const LocDims d = loc_dims(H0, W0);
  CRNN_LDS_ATTR(loc_net_bwd_kernel, 150 * 1024);
  hipLaunchKernelGGL(loc_net_bwd_kernel, dim3(B), dim3(LOC_NTB), loc_bwd_lds(d), s, dtheta, fc1, w1, w2, k2, pool1, c1, pool2, dfc1, scratch, d);
  CRNN_LAUNCH_CHECK();
  const int nred = cdiv(LOC_TERMS, 256), nw1 = cdiv(d.F, LOC_WROWS);
  hipLaunchKernelGGL(loc_net_bwd_reduce_kernel, dim3(nred + nw1 + 1), dim3(256), 0, s, scratch, flat, fc1, dfc1, dtheta, dk2, dbc2, dk1, dbc1, dw1, db1, dw2,
                     db2, B, d.F, nred, nw1);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
