// Depthwise-separable conv stack kernels (utils.py:43-56, 64-70), NHWC; tensors stored as fp32 or bf16 (CRNN_F32 /
// CRNN_BF16, chosen per call), arithmetic and statistics always fp32.
//   depthwise 3x3 'same' (fwd, data-grad = same kernel with flipped taps, weight-grad): the (TH+2) x (TW+2) halo tile of
//   one 128-byte channel slab is staged in LDS with 16-byte loads, each thread produces 3 adjacent pixels from one 3x5
//   window, and the BatchNorm batch statistics (or the 9 tap gradients) leave as a deterministic per-tile partial;
//   BatchNorm finalize / apply(+ReLU6 +MaxPool +Dropout) / backward (two-pass reduce + apply), 16-byte vectors;
//   generic column reductions and the second-stage (double) reductions.
// All kernels are HBM-bandwidth bound; the pointwise 1x1 convs go through gemm.hip.
#include "common.h"
#include <stdlib.h>

#define BN_EPS 1e-3f
#ifndef DW_NT
#define DW_NT 256   // threads per depthwise workgroup
#endif

// VEC consecutive channels of one pixel, widened to fp32 (VEC = 1, 4 or 8; 8 = one 16-byte access of bf16 storage)
template <int VEC>
struct VecF { float v[VEC]; };

// Cache policy of the streaming passes (round 5).  The 256 MB last-level cache holds about two of the conv stack's 123 MB tensors; a pass that reads a
// tensor for the LAST time in the step (or for the last time before the other half of the step) loads it nontemporally so that it does not push out what the
// next kernel is about to read (its own output, the tensors still to be re-read).  Same values, same order: results bit-identical.  Measured on the bf16s step
// (profiles/r05_nt_loads_ab.txt): BatchNorm-backward apply pass -0.03 ms, depthwise-stage backward -0.07 ms, together with the forward passes -0.2 ms.
#ifndef CRNN_NT_BN_BWD2
#define CRNN_NT_BN_BWD2 1   // BatchNorm backward, apply pass: q and g are read for the last time
#endif
#ifndef CRNN_NT_BN_ACT
#define CRNN_NT_BN_ACT 1    // BatchNorm-2 + ReLU6 + pool + dropout (forward): q is not read again before the backward pass
#endif
template <int VEC, typename T>
__device__ __forceinline__ VecF<VEC> vload_nt(const T* p);
template <int VEC, bool NT, typename T>
__device__ __forceinline__ VecF<VEC> vload_s(const T* p);
template <int VEC, typename T>
__device__ __forceinline__ VecF<VEC> vload(const T* p) {
  VecF<VEC> r;
  if (VEC == 8) {
    float8 q = ld8(p);
    r.v[0] = q.lo.x; r.v[1 % VEC] = q.lo.y; r.v[2 % VEC] = q.lo.z; r.v[3 % VEC] = q.lo.w;
    r.v[4 % VEC] = q.hi.x; r.v[5 % VEC] = q.hi.y; r.v[6 % VEC] = q.hi.z; r.v[7 % VEC] = q.hi.w;
  } else if (VEC == 4) {
    float4 q = ld4(p); r.v[0] = q.x; r.v[1 % VEC] = q.y; r.v[2 % VEC] = q.z; r.v[3 % VEC] = q.w;
  } else {
    r.v[0] = ld1(p);
  }
  return r;
}
template <int VEC, typename T>
__device__ __forceinline__ void vstore(T* p, const VecF<VEC>& r) {
  if (VEC == 8) {
    float8 q;
    q.lo = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
    q.hi = make_float4(r.v[4 % VEC], r.v[5 % VEC], r.v[6 % VEC], r.v[7 % VEC]);
    st8(p, q);
  } else if (VEC == 4) {
    st4(p, make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]));
  } else {
    st1(p, r.v[0]);
  }
}
// widest vector the storage type moves in one 16-byte access
template <typename T> struct VecMax { static const int value = 4; };
template <> struct VecMax<bf16_t> { static const int value = 8; };

// ---------------------------------------------------------------------------------------------
// depthwise 3x3, C % 32 == 0 : LDS halo tile
// grid.x = C/32, grid.y = B * ceil(H/TH); block 256 = 8 channel-quads x 32 pixel threads
// mode 0: out = conv(x, k) (flip=1 -> taps flipped = data gradient), optional stats partials
// mode 1: weight gradient partials: dk[tap][c] += x[shifted] * g[center]
// ---------------------------------------------------------------------------------------------
// raw channel vector of the storage type (what sits in HBM and in the LDS tile): 4 fp32 (16 B) or DW_BF16_VN bf16
#ifndef DW_BF16_VN
#define DW_BF16_VN 4
#endif
template <typename T> struct RawV;
template <> struct RawV<float> { typedef float4 type; static const int N = 4; };
#if DW_BF16_VN == 8
template <> struct RawV<bf16_t> { typedef uint4 type; static const int N = 8; };
#else
template <> struct RawV<bf16_t> { typedef uint2 type; static const int N = 4; };
#endif
template <typename V, typename T> __device__ __forceinline__ V ldraw(const T* p) { return *reinterpret_cast<const V*>(p); }
__device__ __forceinline__ void widen(const float4& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void widen(const uint2& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void widen(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void zerov(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zerov(uint2& v) { v = make_uint2(0u, 0u); }
__device__ __forceinline__ void zerov(uint4& v) { v = make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ void narrow_store(float* p, const float (&f)[4]) { st4(p, make_float4(f[0], f[1], f[2], f[3])); }
__device__ __forceinline__ void narrow_store(bf16_t* p, const float (&f)[4]) { st4(p, make_float4(f[0], f[1], f[2], f[3])); }
__device__ __forceinline__ void narrow_store(bf16_t* p, const float (&f)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
}
// sum over the pixel-threads of a wave that share a channel lane (lanes l, l^CL, l^2CL, ...), fixed order
template <int CL>
__device__ __forceinline__ float pixlane_sum(float v) {
#pragma unroll
  for (int o = CL; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// A workgroup handles TH rows x full width x one 128-BYTE channel slab (32 fp32 or 64 bf16 channels): every
// pixel's slab is one full 128-B HBM segment in either storage type (CL lanes x sizeof(V) bytes).
// Each thread produces PXB horizontally adjacent pixels per step: the 3 x (PXB+2) window is read from LDS and widened
// once and feeds all PXB outputs (5 LDS vectors per output at PXB = 3 instead of 9).
#define DW_PXB 3   // fp32 storage; bf16 storage (8 channels per lane, 72 weight registers) uses 2
template <int MODE, int NT, typename T>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void dwconv_tile_kernel(const T* __restrict__ x, const float* __restrict__ k,
                                                          const T* __restrict__ g, T* __restrict__ out,
                                                          float* __restrict__ partials, int B, int H, int W, int C,
                                                          int TH, int TW, int flip, const float* __restrict__ bnstate) {
  // MODE 0: out = conv (+ statistics partials)   MODE 1: weight-gradient partials
  // MODE 2: out = ReLU6(conv * scale + shift) with bnstate = [mean|var|scale|shift] (inference: BatchNorm folded in)
  constexpr bool FWD = (MODE != 1);
  typedef typename RawV<T>::type V;
  constexpr int VN = RawV<T>::N;              // channels per lane
  constexpr int CL = 128 / sizeof(V);         // lanes per pixel (the 128-B slab)
  constexpr int PT = NT / CL;                 // pixel threads
  constexpr int MAXLD = 49152 / (NT * sizeof(V));   // loads per thread for a full 48 KiB tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  V* tile = reinterpret_cast<V*>(smem);
  const int tid = threadIdx.x, c4 = tid & (CL - 1), pt = tid / CL;
  const int cc0 = blockIdx.x * (VN * CL);
  const int nHb = (H + TH - 1) / TH, nWb = (W + TW - 1) / TW;
  const int wb = blockIdx.y % nWb, hb = (blockIdx.y / nWb) % nHb, b = blockIdx.y / (nWb * nHb);
  const int h0 = hb * TH, w0 = wb * TW;
  const int Wt = TW + 2;
  // halo-tile fill with 16-byte loads (8 lanes per pixel slab, whatever the compute vector is); all of this thread's
  // loads are issued before the first LDS write (one HBM latency per workgroup).  32-bit offsets inside the image,
  // (fy, fx) stepped without division or branches.
  {
    constexpr int FL = 8, FPT = NT / FL;                  // fill lanes per pixel, fill pixel-threads
    constexpr int FMAX = 49152 / (NT * 16);               // loads per thread for a full 48 KiB tile
    constexpr int FE = 16 / sizeof(T);                    // elements per 16-byte load
    uint4* tile16 = reinterpret_cast<uint4*>(smem);
    const T* xb = x + (long)b * H * W * C + cc0 + FE * (tid & (FL - 1));
    const int n16 = (TH + 2) * Wt * FL;
    const int dfy = FPT / Wt, dfx = FPT - dfy * Wt;
    for (int base = tid; base < n16; base += FMAX * NT) {
      uint4 v[FMAX];
      int pix0 = base / FL;
      int fy = pix0 / Wt, fx = pix0 - fy * Wt;
#pragma unroll
      for (int uu = 0; uu < FMAX; ++uu) {
        int i = base + uu * NT;
        int gh = h0 + fy - 1, gw = w0 + fx - 1;
        v[uu] = make_uint4(0u, 0u, 0u, 0u);
        if (i < n16 && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
          v[uu] = *reinterpret_cast<const uint4*>(xb + (gh * W + gw) * C);
        fx += dfx; fy += dfy;
        if (fx >= Wt) { fx -= Wt; ++fy; }
      }
#pragma unroll
      for (int uu = 0; uu < FMAX; ++uu) {
        int i = base + uu * NT;
        if (i < n16) tile16[i] = v[uu];
      }
    }
  }
  float kw[9][VN];
  if (FWD) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int ts = flip ? 8 - t : t;
      VecF<VN> w = vload<VN>(&k[ts * C + cc0 + VN * c4]);
#pragma unroll
      for (int e = 0; e < VN; ++e) kw[t][e] = w.v[e];
    }
  }
  float bsc[VN], bsh[VN];
  if (MODE == 2) {
    VecF<VN> v1 = vload<VN>(&bnstate[2 * C + cc0 + VN * c4]), v2 = vload<VN>(&bnstate[3 * C + cc0 + VN * c4]);
#pragma unroll
    for (int e = 0; e < VN; ++e) { bsc[e] = v1.v[e]; bsh[e] = v2.v[e]; }
  }
  __syncthreads();
  float s[VN], ss[VN], dk[9][VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  if (MODE == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < VN; ++e) dk[t][e] = 0.f;
  }
  constexpr int PXB = DW_PXB;
  const int gpr = (TW + PXB - 1) / PXB;        // pixel groups per tile row
  const int npg = TH * gpr;
  const int dly = PT / gpr, dlg = PT - dly * gpr;
  int ly = pt / gpr, lg = pt - ly * gpr;       // stepped incrementally below
  const T* gb = g + (long)b * H * W * C + cc0 + VN * c4;
  T* ob = out + (long)b * H * W * C + cc0 + VN * c4;
  for (int pg = pt; pg < npg; pg += PT) {
    const int lx = lg * PXB;
    const int gh = h0 + ly;
    if (gh >= H) break;
    const int o = (gh * W + w0 + lx) * C;
    float gv[PXB][VN];
    bool pv[PXB];                                // pixel e of this group exists (the last group of a row may be ragged)
#pragma unroll
    for (int e = 0; e < PXB; ++e) pv[e] = (lx + e < TW) && (w0 + lx + e < W);
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < PXB; ++e) {
        V gr; zerov(gr);
        if (pv[e]) gr = ldraw<V>(gb + o + e * C);
        widen(gr, gv[e]);
      }
    }
    float a[PXB][VN];
#pragma unroll
    for (int e = 0; e < PXB; ++e)
#pragma unroll
      for (int c = 0; c < VN; ++c) a[e][c] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float r[PXB + 2][VN];   // (the last group of a ragged row reads past the row end: those outputs are discarded)
#pragma unroll
      for (int j = 0; j < PXB + 2; ++j) widen(tile[((ly + i) * Wt + lx + j) * CL + c4], r[j]);
#pragma unroll
      for (int e = 0; e < PXB; ++e)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int c = 0; c < VN; ++c) {
            if (FWD) a[e][c] = fmaf(r[e + j][c], kw[i * 3 + j][c], a[e][c]);
            else if (pv[e]) dk[i * 3 + j][c] = fmaf(r[e + j][c], gv[e][c], dk[i * 3 + j][c]);   // (a ragged group's window may hold LDS garbage: 0 * NaN)
          }
      __builtin_amdgcn_sched_barrier(0);   // one window row at a time: keeps the live set under the 3-waves/SIMD budget
    }
    if (FWD) {
#pragma unroll
      for (int e = 0; e < PXB; ++e)
        if (pv[e]) {
          if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < VN; ++c) a[e][c] = relu6f(fmaf(a[e][c], bsc[c], bsh[c]));
          }
          narrow_store(ob + o + e * C, a[e]);
          if (partials != nullptr) {
#pragma unroll
            for (int c = 0; c < VN; ++c) { s[c] += a[e][c]; ss[c] = fmaf(a[e][c], a[e][c], ss[c]); }
          }
        }
    }
    lg += dlg; ly += dly;
    if (lg >= gpr) { lg -= gpr; ++ly; }
  }
  if (partials == nullptr || MODE == 2) return;
  // per-tile partials: the 8 pixel-threads of each wave are combined with lane shuffles, the NT/64 waves through LDS
  __syncthreads();  // tile no longer needed
  constexpr int NW = NT / 64, NV = FWD ? 2 : 9;
  float* red = smem;   // [NW][NV][CL][VN]
  const int wave = tid >> 6, lane = tid & 63;
  if (FWD) {
#pragma unroll
    for (int e = 0; e < VN; ++e) { s[e] = pixlane_sum<CL>(s[e]); ss[e] = pixlane_sum<CL>(ss[e]); }
    if (lane < CL) {
#pragma unroll
      for (int e = 0; e < VN; ++e) { red[((wave * NV + 0) * CL + c4) * VN + e] = s[e]; red[((wave * NV + 1) * CL + c4) * VN + e] = ss[e]; }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < VN; ++e) dk[t][e] = pixlane_sum<CL>(dk[t][e]);
    if (lane < CL) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < VN; ++e) red[((wave * NV + t) * CL + c4) * VN + e] = dk[t][e];
    }
  }
  __syncthreads();
  for (int i = tid; i < NV * CL * VN; i += NT) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += red[w * NV * CL * VN + i];
    int v = i / (CL * VN), cch = i % (CL * VN);
    partials[((long)blockIdx.y * NV + v) * C + cc0 + cch] = a;
  }
}

// generic fallback (any C, e.g. the C=1 first block): one thread per output element
template <int MODE>
__global__ void dwconv_naive_kernel(const float* __restrict__ x, const float* __restrict__ k, const float* __restrict__ g,
                                    float* __restrict__ out, int B, int H, int W, int C, int flip) {
  if (MODE == 0) {
    long total = (long)B * H * W * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      int c = (int)(i % C); long pix = i / C;
      int w = (int)(pix % W); long r = pix / W; int h = (int)(r % H); long b = r / H;
      float a = 0.f;
      for (int ii = 0; ii < 3; ++ii)
        for (int j = 0; j < 3; ++j) {
          int gh = h + ii - 1, gw = w + j - 1;
          if (gh >= 0 && gh < H && gw >= 0 && gw < W) {
            int t = ii * 3 + j; if (flip) t = 8 - t;
            a = fmaf(x[((b * H + gh) * W + gw) * C + c], k[t * C + c], a);
          }
        }
      out[i] = a;
    }
  } else {
    // weight gradient, tiny tensors only: one block per (tap, channel), block-wide reduction
    int t = blockIdx.x / C, c = blockIdx.x % C;
    int ii = t / 3, j = t % 3;
    float a = 0.f;
    long npix = (long)B * H * W;
    for (long p = threadIdx.x; p < npix; p += blockDim.x) {
      int w = (int)(p % W); long r = p / W; int h = (int)(r % H); long b = r / H;
      int gh = h + ii - 1, gw = w + j - 1;
      if (gh >= 0 && gh < H && gw >= 0 && gw < W) a = fmaf(x[((b * H + gh) * W + gw) * C + c], g[p * C + c], a);
    }
    __shared__ float red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[t * C + c] = red[0];
  }
}

// single-channel weight gradient: partials[blk][9]
__global__ __launch_bounds__(256) void dwconv_wgrad_c1_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              float* __restrict__ partials, int B, int H, int W) {
  __shared__ float red[9][256];
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const long npix = (long)B * H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < npix; p += (long)gridDim.x * 256L) {
    int w = (int)(p % W); long r = p / W; int h = (int)(r % H);
    float gv = g[p];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int gh = h + i - 1, gw = w + j - 1;
        if (gh >= 0 && gh < H && gw >= 0 && gw < W) acc[i * 3 + j] = fmaf(x[p + (long)(i - 1) * W + (j - 1)], gv, acc[i * 3 + j]);
      }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = acc[t];
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2)
#pragma unroll
      for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x < 9) partials[(long)blockIdx.x * 9 + threadIdx.x] = red[threadIdx.x][0];
}

// Tile = TH rows x TW columns of one 128-byte channel slab.  TW splits W into ceil(W/64) equal column tiles (the CRNN
// maps are at most 36 wide: full-width tiles), TH is as tall as a 48 KiB halo tile allows (3 workgroups per CU) and
// then balanced over the bands of H.
struct DwTile { int TH, TW, nHb, nWb; size_t lds; };
static DwTile dw_pick_tile(int H, int W) {
  const int th_env = crnn_knob("CRNN_DW_TH", 0), tw_env = crnn_knob("CRNN_DW_TW", 0);   // tile overrides (experiment builds only)
  DwTile t;
  t.nWb = cdiv(W, 64);
  t.TW = cdiv(W, t.nWb);
  if (tw_env > 0) t.TW = tw_env;
  t.nWb = cdiv(W, t.TW);
  const int lds_env = crnn_knob("CRNN_DW_LDS", 49152);                                   // halo-tile budget in bytes
  int thmax = (int)((size_t)lds_env / ((size_t)(t.TW + 2) * 128)) - 2;
  if (thmax < 1) thmax = 1;
  if (thmax > H) thmax = H;
  int nb = cdiv(H, thmax);
  t.TH = cdiv(H, nb);
  if (th_env > 0) t.TH = th_env;
  t.nHb = cdiv(H, t.TH);
  const size_t red = (size_t)(DW_NT / 64) * 9 * 64 * 4;  // weight-grad cross-wave reduction scratch (<= 64 channels per slab)
  size_t tile = (size_t)(t.TH + 2) * (t.TW + 2) * 128 + 128 * DW_PXB;   // + slack for the ragged last pixel group
  t.lds = tile > red ? tile : red;
  return t;
}

// number of tiles (= partial rows) the tiled kernels produce for a (B,H,W) map
extern "C" int crnn_dwconv_num_tiles(int B, int H, int W) {
  DwTile t = dw_pick_tile(H, W);
  return B * t.nHb * t.nWb;
}

// out = dwconv3x3(x, k[9][C]); flip=1 gives the data gradient.  If `stat_partials` != null (C%32==0
// only) it receives [num_tiles][2][C] (sum, sumsq) partial BatchNorm statistics of `out`.
// dtype: storage of x/out (CRNN_F32 | CRNN_BF16); arithmetic and statistics are fp32 either way.
template <typename T>
static int dwconv_fwd_launch(const T* x, const float* k, T* out, float* stat_partials, int B, int H, int W, int C, int flip,
                             hipStream_t stream, const float* bnstate = nullptr) {
  DwTile t = dw_pick_tile(H, W);
  if (t.lds > 160 * 1024) return CRNN_ERR_UNSUPPORTED;
  if (t.lds > 48 * 1024) {   // more dynamic LDS than the default launch limit: raise it to exactly what this tile needs
    hipError_t e = hipFuncSetAttribute((const void*)dwconv_tile_kernel<0, DW_NT, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds);
    if (e != hipSuccess) return (int)e;
    if (bnstate) { e = hipFuncSetAttribute((const void*)dwconv_tile_kernel<2, DW_NT, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds); if (e != hipSuccess) return (int)e; }
  }
  const int slab = 128 / (int)sizeof(T);   // channels per workgroup: 32 (fp32) or 64 (bf16)
  if (C % slab) return CRNN_ERR_UNSUPPORTED;
  dim3 grid(C / slab, B * t.nHb * t.nWb);
  if (bnstate) hipLaunchKernelGGL((dwconv_tile_kernel<2, DW_NT, T>), grid, dim3(DW_NT), t.lds, stream, x, k, (const T*)nullptr, out, (float*)nullptr, B, H, W, C, t.TH, t.TW, flip, bnstate);
  else hipLaunchKernelGGL((dwconv_tile_kernel<0, DW_NT, T>), grid, dim3(DW_NT), t.lds, stream, x, k, (const T*)nullptr, out, stat_partials, B, H, W, C, t.TH, t.TW, flip, (const float*)nullptr);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

extern "C" int crnn_dwconv3x3_fwd_ex(const void* x, const float* k, void* out, float* stat_partials, int B, int H, int W,
                                     int C, int flip, int dtype, hipStream_t stream) {
  if (C % 32 == 0) {
    if (dtype == CRNN_BF16) return dwconv_fwd_launch<bf16_t>((const bf16_t*)x, k, (bf16_t*)out, stat_partials, B, H, W, C, flip, stream);
    return dwconv_fwd_launch<float>((const float*)x, k, (float*)out, stat_partials, B, H, W, C, flip, stream);
  }
  if (stat_partials || dtype != CRNN_F32) return CRNN_ERR_UNSUPPORTED;
  long total = (long)B * H * W * C;
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dwconv_naive_kernel<0>, dim3(blocks), dim3(256), 0, stream, (const float*)x, k, nullptr, (float*)out, B, H, W, C, flip);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// Inference form: out = ReLU6(BatchNorm(dwconv3x3(x, k))) with the normalisation folded into the conv's epilogue
// (bnstate = [mean|var|scale|shift] from crnn_bn_infer_state); C % (128 / sizeof(storage)) == 0 only.
extern "C" int crnn_dwconv3x3_bn_relu6_fwd(const void* x, const float* k, const float* bnstate, void* out, int B, int H, int W, int C,
                                           int dtype, hipStream_t stream) {
  if (!bnstate) return CRNN_ERR_ARG;
  if (dtype == CRNN_BF16) return dwconv_fwd_launch<bf16_t>((const bf16_t*)x, k, (bf16_t*)out, nullptr, B, H, W, C, 0, stream, bnstate);
  return dwconv_fwd_launch<float>((const float*)x, k, (float*)out, nullptr, B, H, W, C, 0, stream, bnstate);
}
extern "C" int crnn_dwconv3x3_fwd(const float* x, const float* k, float* out, float* stat_partials, int B, int H, int W,
                                  int C, int flip, hipStream_t stream) {
  return crnn_dwconv3x3_fwd_ex(x, k, out, stat_partials, B, H, W, C, flip, CRNN_F32, stream);
}

// ---------------------------------------------------------------------------------------------
// Column reductions over a row-major [M][C] matrix -> partials [nchunk][NV][C]
// NV=1: sum; NV=2: sum and sum of squares.  Deterministic (fixed chunking, fixed order).
// ---------------------------------------------------------------------------------------------
template <int VEC, int NV, typename T>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ x, float* __restrict__ partials, long M,
                                                        int C, int ld, int CW, int rows_per_chunk) {
  // CW = power of two >= min(C/VEC, 256); thread -> (cl = tid % CW, rt = tid / CW)
  __shared__ float red[2][256 * VEC];
  const int tid = threadIdx.x, cl = tid % CW, rt = tid / CW, RT = 256 / CW;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  const int CL = C / VEC;
  for (int cb = 0; cb < CL; cb += CW) {
    int c = cb + cl;
    float s[VEC], q[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (c < CL) {
      long r = r0 + rt;
      if (VEC > 1) {  // 4 rows in flight per thread (independent 16-byte loads)
        for (; r + 3L * RT < r1; r += 4L * RT) {
          VecF<VEC> v0 = vload<VEC>(&x[r * ld + VEC * c]);
          VecF<VEC> v1 = vload<VEC>(&x[(r + RT) * ld + VEC * c]);
          VecF<VEC> v2 = vload<VEC>(&x[(r + 2L * RT) * ld + VEC * c]);
          VecF<VEC> v3 = vload<VEC>(&x[(r + 3L * RT) * ld + VEC * c]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            s[e] += (v0.v[e] + v1.v[e]) + (v2.v[e] + v3.v[e]);
            if (NV == 2) q[e] += (v0.v[e] * v0.v[e] + v1.v[e] * v1.v[e]) + (v2.v[e] * v2.v[e] + v3.v[e] * v3.v[e]);
          }
        }
      }
      for (; r < r1; r += RT) {
        VecF<VEC> v = vload<VEC>(&x[r * ld + VEC * c]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] += v.v[e]; if (NV == 2) q[e] = fmaf(v.v[e], v.v[e], q[e]); }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][tid * VEC + e] = s[e]; if (NV == 2) red[1][tid * VEC + e] = q[e]; }
    __syncthreads();
    if (rt == 0 && c < CL) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float a = 0.f, b2 = 0.f;
        for (int r = 0; r < RT; ++r) { a += red[0][(r * CW + cl) * VEC + e]; if (NV == 2) b2 += red[1][(r * CW + cl) * VEC + e]; }
        partials[((long)blockIdx.x * NV + 0) * C + c * VEC + e] = a;
        if (NV == 2) partials[((long)blockIdx.x * NV + 1) * C + c * VEC + e] = b2;
      }
    }
  }
}

static inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// rows per chunk: aim for >= ~512 chunks (fills 256 CUs), between 16 and 1024 rows each
static inline int colreduce_rpc(long M) { long r = 1024; while (r > 16 && M / r < 512) r >>= 1; return (int)r; }
extern "C" int crnn_colreduce_chunks(long M) { return cdiv(M, colreduce_rpc(M)); }

// partials [crnn_colreduce_chunks(M)][nv][C]; dtype = storage of x
template <int VEC, typename T>
static void colreduce_go(const T* x, float* partials, long M, int C, int ld, int nv, int chunks, int rpc, hipStream_t stream) {
  const int CL = C / VEC, CW = pow2_ge(CL < 256 ? CL : 256);
  if (nv == 1) hipLaunchKernelGGL((colreduce_kernel<VEC, 1, T>), dim3(chunks), dim3(256), 0, stream, x, partials, M, C, ld, CW, rpc);
  else hipLaunchKernelGGL((colreduce_kernel<VEC, 2, T>), dim3(chunks), dim3(256), 0, stream, x, partials, M, C, ld, CW, rpc);
}
template <typename T>
static int colreduce_launch(const T* x, float* partials, long M, int C, int ld, int nv, hipStream_t stream) {
  int rpc = colreduce_rpc(M);
  int chunks = cdiv(M, rpc);
  const int VM = VecMax<T>::value;
  const bool al = ((((uintptr_t)x) & 15) == 0);
  if (VM == 8 && al && (C % 8 == 0) && (ld % 8 == 0)) colreduce_go<VecMax<T>::value, T>(x, partials, M, C, ld, nv, chunks, rpc, stream);
  else if (al && (C % 4 == 0) && (ld % 4 == 0)) colreduce_go<4, T>(x, partials, M, C, ld, nv, chunks, rpc, stream);
  else colreduce_go<1, T>(x, partials, M, C, ld, nv, chunks, rpc, stream);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_colreduce_ex(const void* x, float* partials, long M, int C, int ld, int nv, int dtype, hipStream_t stream) {
  if (nv != 1 && nv != 2) return CRNN_ERR_ARG;
  if (dtype == CRNN_BF16) return colreduce_launch<bf16_t>((const bf16_t*)x, partials, M, C, ld, nv, stream);
  return colreduce_launch<float>((const float*)x, partials, M, C, ld, nv, stream);
}
extern "C" int crnn_colreduce(const float* x, float* partials, long M, int C, int ld, int nv, hipStream_t stream) {
  return crnn_colreduce_ex(x, partials, M, C, ld, nv, CRNN_F32, stream);
}

// Second-stage reductions run as (RED_CH outputs) x (RED_PL part-lanes) blocks: each lane adds every RED_PL-th
// partial in double, four independent loads in flight; lanes are then combined in a fixed order (deterministic).
#define RED_CH 16
#define RED_PL 64
__device__ __forceinline__ double part_lane_sum(const float* __restrict__ base, long stride, int nparts, int lane) {
  double a = 0.0;
  int p = lane;
  for (; p + 3 * RED_PL < nparts; p += 4 * RED_PL) {
    float v0 = base[(long)p * stride], v1 = base[(long)(p + RED_PL) * stride];
    float v2 = base[(long)(p + 2 * RED_PL) * stride], v3 = base[(long)(p + 3 * RED_PL) * stride];
    a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
  }
  for (; p < nparts; p += RED_PL) a += (double)base[(long)p * stride];
  return a;
}

// two interleaved sums (sum / sum-of-products rows of one partials array): eight independent loads in flight
__device__ __forceinline__ void part_lane_sum2(const float* __restrict__ b0, const float* __restrict__ b1, long stride, int nparts, int lane,
                                               double& s0, double& s1) {
  double a = 0.0, b = 0.0;
  int p = lane;
  // sixteen independent loads in flight: a lane's share of ~1900 partial rows is ~30 rows, i.e. four round trips to a remote L2 / HBM instead of eight
  for (; p + 7 * RED_PL < nparts; p += 8 * RED_PL) {
    float v[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long o = (long)(p + u * RED_PL) * stride; v[u] = b0[o]; w[u] = b1[o]; }
    a += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) + (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
    b += (((double)w[0] + (double)w[1]) + ((double)w[2] + (double)w[3])) + (((double)w[4] + (double)w[5]) + ((double)w[6] + (double)w[7]));
  }
  for (; p + 3 * RED_PL < nparts; p += 4 * RED_PL) {
    const long o0 = (long)p * stride, o1 = (long)(p + RED_PL) * stride, o2 = (long)(p + 2 * RED_PL) * stride, o3 = (long)(p + 3 * RED_PL) * stride;
    float v0 = b0[o0], v1 = b0[o1], v2 = b0[o2], v3 = b0[o3];
    float w0 = b1[o0], w1 = b1[o1], w2 = b1[o2], w3 = b1[o3];
    a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    b += ((double)w0 + (double)w1) + ((double)w2 + (double)w3);
  }
  for (; p < nparts; p += RED_PL) { a += (double)b0[(long)p * stride]; b += (double)b1[(long)p * stride]; }
  s0 = a; s1 = b;
}
// lanes -> one value per output, fixed two-level order (8 groups of 8 lanes): red[2][RED_PL][RED_CH], result valid for y == 0
__device__ __forceinline__ void lanes_sum2(double (&red)[2][RED_PL][RED_CH], double& s, double& q) {
  static_assert(RED_PL == 64, "two-level lane reduction is written for 64 part-lanes");
  __syncthreads();
  const int x = threadIdx.x, y = threadIdx.y;
  if (y < 8) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a += red[0][y * 8 + r][x]; b += red[1][y * 8 + r][x]; }
    red[0][y * 8][x] = a; red[1][y * 8][x] = b;      // (only this thread reads rows y*8 .. y*8+7)
  }
  __syncthreads();
  s = 0.0; q = 0.0;
  if (y == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) { s += red[0][r * 8][x]; q += red[1][r * 8][x]; }
  }
}

// out[i] = scale * sum_p partials[p][i], i < n  (double accumulation)
__global__ __launch_bounds__(RED_CH * RED_PL) void partials_sum_kernel(const float* __restrict__ partials, int nparts, int n, float* __restrict__ out, float scale) {
  // gridDim.y > 1: blockIdx.y reduces its own contiguous chunk of the partial rows into out[blockIdx.y][n]
  __shared__ double red[RED_PL][RED_CH];
  const int chunk = (nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * chunk;
  const int np = min(chunk, nparts - p0);
  int i = blockIdx.x * RED_CH + threadIdx.x;
  red[threadIdx.y][threadIdx.x] = (i < n && np > 0) ? part_lane_sum(partials + (long)p0 * n + i, n, np, threadIdx.y) : 0.0;
  __syncthreads();
  if (threadIdx.y == 0 && i < n) {
    double s = 0.0;
    for (int r = 0; r < RED_PL; ++r) s += red[r][threadIdx.x];
    out[(long)blockIdx.y * n + i] = (float)(s * scale);
  }
}

extern "C" int crnn_partials_sum(const float* partials, int nparts, int n, float* out, float scale, hipStream_t stream) {
  hipLaunchKernelGGL(partials_sum_kernel, dim3(cdiv(n, RED_CH)), dim3(RED_CH, RED_PL), 0, stream, partials, nparts, n, out, scale);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// weight gradient of the depthwise conv: dk[9][C] = sum x[shifted] * g.  scratch: [num_tiles][9][C]
template <typename T>
static int dwconv_wgrad_launch(const T* x, const T* g, float* dk, float* scratch, int B, int H, int W, int C, hipStream_t stream) {
  DwTile t = dw_pick_tile(H, W);
  if (t.lds > 160 * 1024) return CRNN_ERR_UNSUPPORTED;
  if (t.lds > 48 * 1024) {   // more dynamic LDS than the default launch limit: raise it to exactly what this tile needs
    hipError_t e = hipFuncSetAttribute((const void*)dwconv_tile_kernel<1, DW_NT, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds);
    if (e != hipSuccess) return (int)e;
  }
  int ntiles = B * t.nHb * t.nWb;
  const int slab = 128 / (int)sizeof(T);
  if (C % slab) return CRNN_ERR_UNSUPPORTED;
  dim3 grid(C / slab, ntiles);
  hipLaunchKernelGGL((dwconv_tile_kernel<1, DW_NT, T>), grid, dim3(DW_NT), t.lds, stream, x, (const float*)nullptr, g, (T*)nullptr, scratch, B, H, W, C, t.TH, t.TW, 0, (const float*)nullptr);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, ntiles, 9 * C, dk, 1.f, stream);
}

extern "C" int crnn_dwconv3x3_wgrad_ex(const void* x, const void* g, float* dk, float* scratch, int B, int H, int W, int C,
                                       int dtype, hipStream_t stream) {
  if (C % 32 == 0) {
    if (dtype == CRNN_BF16) return dwconv_wgrad_launch<bf16_t>((const bf16_t*)x, (const bf16_t*)g, dk, scratch, B, H, W, C, stream);
    return dwconv_wgrad_launch<float>((const float*)x, (const float*)g, dk, scratch, B, H, W, C, stream);
  }
  if (dtype != CRNN_F32) return CRNN_ERR_UNSUPPORTED;
  if (C == 1) {  // first block: one channel, B*H*W pixels -> per-block partials [nblk][9], then the 2nd stage
    long npix = (long)B * H * W;
    int nblk = cdiv(npix, 2048); if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(dwconv_wgrad_c1_kernel, dim3(nblk), dim3(256), 0, stream, (const float*)x, (const float*)g, scratch, B, H, W);
    CRNN_LAUNCH_CHECK();
    return crnn_partials_sum(scratch, nblk, 9, dk, 1.f, stream);
  }
  hipLaunchKernelGGL(dwconv_naive_kernel<1>, dim3(9 * C), dim3(256), 0, stream, (const float*)x, nullptr, (const float*)g, dk, B, H, W, C, 0);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_dwconv3x3_wgrad(const float* x, const float* g, float* dk, float* scratch, int B, int H, int W, int C,
                                    hipStream_t stream) {
  return crnn_dwconv3x3_wgrad_ex(x, g, dk, scratch, B, H, W, C, CRNN_F32, stream);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm (axis=-1, eps=1e-3): statistics finalize.  bnstate = [mean | var | scale | shift] (4*C)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RED_CH * RED_PL) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, int C, double inv_n,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ bnstate) {
  __shared__ double red[2][RED_PL][RED_CH];
  int c = blockIdx.x * RED_CH + threadIdx.x;
  double s = 0.0, q = 0.0;
  if (c < C) part_lane_sum2(partials + c, partials + C + c, 2L * C, nparts, threadIdx.y, s, q);
  red[0][threadIdx.y][threadIdx.x] = s; red[1][threadIdx.y][threadIdx.x] = q;
  lanes_sum2(red, s, q);
  if (threadIdx.y == 0 && c < C) {
    double mean = s * inv_n;
    double var = q * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    float inv = (float)(1.0 / sqrt(var + (double)BN_EPS));
    float sc = gamma[c] * inv;
    bnstate[c] = (float)mean; bnstate[C + c] = (float)var;
    bnstate[2 * C + c] = sc; bnstate[3 * C + c] = beta[c] - (float)mean * sc;
  }
}

// inference mode: scale/shift from the moving statistics
__global__ void bn_infer_state_kernel(const float* __restrict__ mmean, const float* __restrict__ mvar,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                      float* __restrict__ bnstate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float sc = gamma[c] / sqrtf(mvar[c] + BN_EPS);
    bnstate[c] = mmean[c]; bnstate[C + c] = mvar[c];
    bnstate[2 * C + c] = sc; bnstate[3 * C + c] = beta[c] - mmean[c] * sc;
  }
}

extern "C" int crnn_bn_finalize(const float* partials, int nparts, int C, long n, const float* gamma, const float* beta,
                                float* bnstate, hipStream_t stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, RED_CH)), dim3(RED_CH, RED_PL), 0, stream, partials, nparts, C, 1.0 / (double)n, gamma, beta, bnstate);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// Same, for long partial lists (one row per GEMM tile): a first launch folds the rows into CRNN_BN_FOLD_ROWS chunk
// sums (scratch: CRNN_BN_FOLD_ROWS * 2 * C floats) with hundreds of workgroups, the finalize then reads those.
#define CRNN_BN_FOLD_ROWS 32
extern "C" int crnn_bn_finalize_folded(const float* partials, int nparts, int C, long n, const float* gamma, const float* beta,
                                       float* bnstate, float* scratch, hipStream_t stream) {
  if (nparts <= 1024 || scratch == nullptr) return crnn_bn_finalize(partials, nparts, C, n, gamma, beta, bnstate, stream);
  hipLaunchKernelGGL(partials_sum_kernel, dim3(cdiv(2 * C, RED_CH), CRNN_BN_FOLD_ROWS), dim3(RED_CH, RED_PL), 0, stream, partials, nparts, 2 * C, scratch, 1.f);
  CRNN_LAUNCH_CHECK();
  return crnn_bn_finalize(scratch, CRNN_BN_FOLD_ROWS, C, n, gamma, beta, bnstate, stream);
}

extern "C" int crnn_bn_infer_state(const float* mmean, const float* mvar, const float* gamma, const float* beta, int C,
                                   float* bnstate, hipStream_t stream) {
  hipLaunchKernelGGL(bn_infer_state_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, mmean, mvar, gamma, beta, C, bnstate);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// The same for up to CRNN_BN_INFER_BATCH_MAX BatchNorm layers in one launch (the predict path knows all of them before its first conv: 14
// launches of ~5 us each otherwise): arrays of n device pointers / channel counts on the host; grid.y = layer.
struct BnInferBatch { const float* mmean[CRNN_BN_INFER_BATCH_MAX]; const float* mvar[CRNN_BN_INFER_BATCH_MAX]; const float* gamma[CRNN_BN_INFER_BATCH_MAX];
                      const float* beta[CRNN_BN_INFER_BATCH_MAX]; float* bnstate[CRNN_BN_INFER_BATCH_MAX]; int C[CRNN_BN_INFER_BATCH_MAX]; };
__global__ void bn_infer_state_batch_kernel(BnInferBatch b) {
  const int j = blockIdx.y, C = b.C[j], c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float mm = b.mmean[j][c], mv = b.mvar[j][c];
    const float sc = b.gamma[j][c] / sqrtf(mv + BN_EPS);
    float* st = b.bnstate[j];
    st[c] = mm; st[C + c] = mv; st[2 * C + c] = sc; st[3 * C + c] = b.beta[j][c] - mm * sc;
  }
}
extern "C" int crnn_bn_infer_state_batch(int n, const float* const* mmean, const float* const* mvar, const float* const* gamma, const float* const* beta,
                                         const int* C, float* const* bnstate, hipStream_t stream) {
  if (n < 0 || n > CRNN_BN_INFER_BATCH_MAX || (n && (!mmean || !mvar || !gamma || !beta || !C || !bnstate))) return CRNN_ERR_ARG;
  if (n == 0) return CRNN_OK;
  BnInferBatch b; int cmax = 0;
  for (int j = 0; j < n; ++j) {
    if (!mmean[j] || !mvar[j] || !gamma[j] || !beta[j] || !bnstate[j] || C[j] <= 0) return CRNN_ERR_ARG;
    b.mmean[j] = mmean[j]; b.mvar[j] = mvar[j]; b.gamma[j] = gamma[j]; b.beta[j] = beta[j]; b.bnstate[j] = bnstate[j]; b.C[j] = C[j];
    if (C[j] > cmax) cmax = C[j];
  }
  hipLaunchKernelGGL(bn_infer_state_batch_kernel, dim3(cdiv(cmax, 256), n), dim3(256), 0, stream, b);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// ---------------------------------------------------------------------------------------------
// y = Dropout(MaxPool(ReLU6(x*scale+shift)))   (utils.py:45-56).  ph=pw=1: no pooling; rate=0: no dropout.
// x [B,H,W,C] -> y [B,H/ph,W/pw,C]
// ---------------------------------------------------------------------------------------------
// IDX: index type of the element counter -- unsigned (32-bit divisions: ~25 VALU instructions each) whenever the tensor has fewer than 2^31
// vectors, long otherwise (64-bit divisions cost ~100 each, and a vector needs two to five of them against ~60 instructions of payload).
// PHT x PWT: the pool window as compile-time constants (0 x 0: run-time ph x pw).
template <int VEC, typename TI, typename TO, typename IDX, int PHT, int PWT>
__global__ void bn_act_pool_drop_kernel(const TI* __restrict__ x, const float* __restrict__ bnstate,
                                        TO* __restrict__ y, int B, int H, int W, int C, int ph_, int pw_, float rate,
                                        uint64_t seed, uint32_t layer, TI* __restrict__ qmax = nullptr) {
  const int ph = PHT ? PHT : ph_, pw = PWT ? PWT : pw_;
  const int Ho = H / ph, Wo = W / pw, CL = C / VEC;
  const IDX total = (IDX)((long)B * Ho * Wo * CL);
  const float inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  const float* sc = bnstate + 2 * C; const float* sh = bnstate + 3 * C;
  for (IDX i = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * (IDX)blockDim.x) {
    const int cl = (int)(i % (IDX)CL); const IDX pix = i / (IDX)CL;
    VecF<VEC> m, s = vload<VEC>(sc + cl * VEC), t = vload<VEC>(sh + cl * VEC);
    if (ph * pw == 1) {
      VecF<VEC> v = vload_s<VEC, (CRNN_NT_BN_ACT != 0)>(&x[(long)pix * C + cl * VEC]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) m.v[e] = relu6f(fmaf(v.v[e], s.v[e], t.v[e]));
    } else {
      const int wo = (int)(pix % (IDX)Wo); const IDX r = pix / (IDX)Wo; const int ho = (int)(r % (IDX)Ho); const long b = (long)(r / (IDX)Ho);
#pragma unroll
      for (int e = 0; e < VEC; ++e) m.v[e] = -INFINITY;
      const long base = ((b * H + (long)ho * ph) * W + (long)wo * pw) * C + cl * VEC;
      if (PHT && qmax) {   // training: also x at the window's FIRST maximum of the unclamped BatchNorm value (the backward's statistics pass reads it instead of the window)
        VecF<VEC> bt, bq;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { bt.v[e] = -INFINITY; bq.v[e] = 0.f; }
#pragma unroll
        for (int ii = 0; ii < (PHT ? PHT : 1); ++ii)
#pragma unroll
          for (int j = 0; j < (PWT ? PWT : 1); ++j) {
            VecF<VEC> v = vload_s<VEC, (CRNN_NT_BN_ACT != 0)>(&x[base + ((long)ii * W + j) * C]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const float tt = fmaf(v.v[e], s.v[e], t.v[e]);
              bq.v[e] = tt > bt.v[e] ? v.v[e] : bq.v[e];        // strict '>': the first maximum in scan order (bn_bwd_pool_kernel's rule)
              bt.v[e] = fmaxf(bt.v[e], tt);
            }
          }
#pragma unroll
        for (int e = 0; e < VEC; ++e) m.v[e] = relu6f(bt.v[e]);  // ReLU6 is monotone: the maximum of the clamped values is the clamped maximum, exactly
        vstore<VEC>(&qmax[(long)pix * C + cl * VEC], bq);
      } else if (PHT) {
#pragma unroll
        for (int ii = 0; ii < (PHT ? PHT : 1); ++ii)
#pragma unroll
          for (int j = 0; j < (PWT ? PWT : 1); ++j) {
            VecF<VEC> v = vload_s<VEC, (CRNN_NT_BN_ACT != 0)>(&x[base + ((long)ii * W + j) * C]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) m.v[e] = fmaxf(m.v[e], relu6f(fmaf(v.v[e], s.v[e], t.v[e])));
          }
      } else {
        for (int ii = 0; ii < ph; ++ii)
          for (int j = 0; j < pw; ++j) {
            VecF<VEC> v = vload_s<VEC, (CRNN_NT_BN_ACT != 0)>(&x[base + ((long)ii * W + j) * C]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) m.v[e] = fmaxf(m.v[e], relu6f(fmaf(v.v[e], s.v[e], t.v[e])));
          }
      }
    }
    if (!y) continue;                                          // (qmax only: the next depthwise kernel forms the block output from it)
    const long obase = (long)pix * C + cl * VEC;
    float dm[VEC];
    drop_scale_vec<VEC>(seed, layer, (uint64_t)obase, rate, inv_keep, dm);
#pragma unroll
    for (int e = 0; e < VEC; ++e) m.v[e] *= dm[e];
    vstore<VEC>(&y[obase], m);
  }
}

template <int VEC, typename TI, typename TO>
static void bn_act_go(const TI* x, const float* bnstate, TO* y, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                      uint32_t layer, hipStream_t stream, TI* qmax = nullptr) {
  long total = (long)B * (H / ph) * (W / pw) * C;
  int blocks = cdiv(total / VEC, 256); if (blocks > 8192) blocks = 8192;
  const bool small = total / VEC + 8192L * 256 < (1L << 31);       // the grid-stride counter stays below 2^31 + one stride: fits unsigned
  const dim3 g(blocks), t(256);
#define BN_ACT_LAUNCH(IDX, PHT, PWT) hipLaunchKernelGGL((bn_act_pool_drop_kernel<VEC, TI, TO, IDX, PHT, PWT>), g, t, 0, stream, x, bnstate, y, B, H, W, C, ph, pw, rate, seed, layer, qmax)
  if (!small) BN_ACT_LAUNCH(long, 0, 0);
  else if (ph == 2 && pw == 2) BN_ACT_LAUNCH(unsigned, 2, 2);
  else if (ph == 1 && pw == 2) BN_ACT_LAUNCH(unsigned, 1, 2);
  else BN_ACT_LAUNCH(unsigned, 0, 0);
#undef BN_ACT_LAUNCH
}
template <typename TI, typename TO>
static int bn_act_launch(const TI* x, const float* bnstate, TO* y, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                         uint32_t layer, hipStream_t stream, TI* qmax = nullptr) {
  if (qmax && !((ph == 2 && pw == 2) || (ph == 1 && pw == 2))) return CRNN_ERR_UNSUPPORTED;   // (the compile-time windows)
  if (qmax && (long)B * (H / ph) * (W / pw) * C + 8192L * 256 * 8 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  const bool al = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)bnstate | (uintptr_t)qmax) & 15) == 0);
  const int VM = (VecMax<TI>::value == 8 && VecMax<TO>::value == 8) ? 8 : 4;
  if (VM == 8 && al && C % 8 == 0) bn_act_go<(VecMax<TI>::value == 8 && VecMax<TO>::value == 8) ? 8 : 4>(x, bnstate, y, B, H, W, C, ph, pw, rate, seed, layer, stream, qmax);
  else if (al && C % 4 == 0) bn_act_go<4>(x, bnstate, y, B, H, W, C, ph, pw, rate, seed, layer, stream, qmax);
  else bn_act_go<1>(x, bnstate, y, B, H, W, C, ph, pw, rate, seed, layer, stream, qmax);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// dt_in / dt_out: storage of x / y
extern "C" int crnn_bn_act_pool_drop_ex(const void* x, const float* bnstate, void* y, int B, int H, int W, int C, int ph,
                                        int pw, float rate, uint64_t seed, uint32_t layer, int dt_in, int dt_out, hipStream_t stream) {
  if (dt_in == CRNN_BF16 && dt_out == CRNN_BF16) return bn_act_launch((const bf16_t*)x, bnstate, (bf16_t*)y, B, H, W, C, ph, pw, rate, seed, layer, stream);
  if (dt_in == CRNN_BF16) return bn_act_launch((const bf16_t*)x, bnstate, (float*)y, B, H, W, C, ph, pw, rate, seed, layer, stream);
  if (dt_out == CRNN_BF16) return bn_act_launch((const float*)x, bnstate, (bf16_t*)y, B, H, W, C, ph, pw, rate, seed, layer, stream);
  return bn_act_launch((const float*)x, bnstate, (float*)y, B, H, W, C, ph, pw, rate, seed, layer, stream);
}
// ... that also writes qmax [B][H/ph][W/pw][C] (storage dt_in): x at the first maximum of x * scale + shift over each pool window (2 x 2 or 1 x 2 only) --
// what crnn_bn_bwd_qmax_ex's statistics pass reads instead of the whole window.  y as above, bit for bit; y NULL: qmax only (the block output is then
// Dropout(ReLU6(qmax * scale + shift)) element by element -- what crnn_dwconv3x3_fwd_stream_pro_ex forms from it, bit for bit).
extern "C" int crnn_bn_act_pool_drop_qmax_ex(const void* x, const float* bnstate, void* y, void* qmax, int B, int H, int W, int C, int ph, int pw, float rate,
                                             uint64_t seed, uint32_t layer, int dt_in, int dt_out, hipStream_t stream) {
  if (!x || !bnstate || !qmax) return CRNN_ERR_ARG;          // y may be NULL: qmax only
  if (dt_in == CRNN_BF16 && dt_out == CRNN_BF16) return bn_act_launch((const bf16_t*)x, bnstate, (bf16_t*)y, B, H, W, C, ph, pw, rate, seed, layer, stream, (bf16_t*)qmax);
  if (dt_in == CRNN_BF16) return bn_act_launch((const bf16_t*)x, bnstate, (float*)y, B, H, W, C, ph, pw, rate, seed, layer, stream, (bf16_t*)qmax);
  if (dt_out == CRNN_BF16) return bn_act_launch((const float*)x, bnstate, (bf16_t*)y, B, H, W, C, ph, pw, rate, seed, layer, stream, (float*)qmax);
  return bn_act_launch((const float*)x, bnstate, (float*)y, B, H, W, C, ph, pw, rate, seed, layer, stream, (float*)qmax);
}
extern "C" int crnn_bn_act_pool_drop(const float* x, const float* bnstate, float* y, int B, int H, int W, int C, int ph,
                                     int pw, float rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  return crnn_bn_act_pool_drop_ex(x, bnstate, y, B, H, W, C, ph, pw, rate, seed, layer, CRNN_F32, CRNN_F32, stream);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward (train mode).  x = pre-BN tensor [B,H,W,C]; the upstream gradient arrives as
// g [B,H/ph,W/pw,C] w.r.t. Dropout(MaxPool(ReLU6(BN(x)))) and is routed on the fly:
//   gy = [first-argmax of the pool window] * g * dropmask * [0 < relu6(bn(x)) < 6]
// pass 1 (reduce): partials [chunk][2][C] = (sum gy, sum gy*xhat)
// pass 2 (apply):  dx = scale * (gy - c1 - xhat*c2),   c1 = sum(gy)/n, c2 = sum(gy*xhat)/n
// Threads: VEC channels per thread (16-byte accesses when C % 4 == 0), CW channel-threads x 256/CW
// row-threads; rows = pixels.  POOL=false (ph=pw=1) needs no per-element index arithmetic at all.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct BnBwdArgsT {
  const T* x; const T* g; const float* bnstate; const float* gamma;
  int B, H, W, C, ph, pw; float rate; uint64_t seed; uint32_t layer;
  // round 6 (fp32 tensors): pass 2 writes dx as bf16 PLANES instead (plane pl of element i at dxp[pl * dxps + i]; the words of crnn_split3_pair) -- what the
  // pointwise GEMMs that read dx (gemm_pres.hip, gemm_wgrad3.hip) stage, split once here instead of once per tile there.  Null: dx as T
  unsigned short* dxp = nullptr; long dxps = 0; int dxpl = 0;
  // round 6: x at each pool window's first maximum, written by the forward (crnn_bn_act_pool_drop_qmax_ex): the statistics pass of the pooled kernel reads it
  // (one value per window) instead of the window -- the same sums bit for bit from a quarter (half) of the bytes
  const T* qmax = nullptr;
};
// dx (fp32 values) of VEC consecutive elements as planes (VEC = 4: two words per plane)
template <int VEC, typename T>
__device__ __forceinline__ void bn_store_dx(const BnBwdArgsT<T>& a, T* dx, long idx, const VecF<VEC>& o) {
  if constexpr (VEC == 4 && sizeof(T) == 4) {
    if (a.dxp) {
      unsigned w0[3], w1[3];
      crnn_split3_pair(o.v[0], o.v[1], w0[0], w0[1], w0[2]);
      crnn_split3_pair(o.v[2 % VEC], o.v[3 % VEC], w1[0], w1[1], w1[2]);
      unsigned short* q = a.dxp + idx;
      *reinterpret_cast<uint2*>(q) = make_uint2(w0[0], w1[0]);
      *reinterpret_cast<uint2*>(q + a.dxps) = make_uint2(w0[1], w1[1]);
      if (a.dxpl == 3) *reinterpret_cast<uint2*>(q + 2 * a.dxps) = make_uint2(w0[2], w1[2]);
      return;
    }
  }
  vstore<VEC>(&dx[idx], o);
}

template <int VEC, typename T>
__device__ __forceinline__ VecF<VEC> vload_nt(const T* p) {
  typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
  VecF<VEC> r;
  if constexpr (VEC == 8 && sizeof(T) == 2) {
    const u32x4_nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.v[2 * q] = __uint_as_float(u[q] << 16); r.v[2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u); }
    return r;
  } else if constexpr (VEC % 4 == 0 && sizeof(T) == 4) {
#pragma unroll
    for (int h = 0; h < VEC / 4; ++h) {
      const u32x4_nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p) + h);
#pragma unroll
      for (int q = 0; q < 4; ++q) r.v[4 * h + q] = __uint_as_float(u[q]);
    }
    return r;
  } else {
    return vload<VEC>(p);
  }
}
template <int VEC, bool NT, typename T>
__device__ __forceinline__ VecF<VEC> vload_s(const T* p) {
  if constexpr (NT) return vload_nt<VEC>(p); else return vload<VEC>(p);
}
// gy for VEC consecutive channels starting at c0 of pixel row r
template <int VEC, bool POOL, typename T, bool NT = false>
__device__ __forceinline__ VecF<VEC> bn_gy_vec(const BnBwdArgsT<T>& a, long r, int c0, const VecF<VEC>& xv, const VecF<VEC>& sc,
                                               const VecF<VEC>& sh, float inv_keep) {
  VecF<VEC> out, y;
  // no pooling: only 0 < y < 6 is asked of y below, which the clamp does not change (it is skipped); the window scan needs the clamped value
#pragma unroll
  for (int e = 0; e < VEC; ++e) { const float t = fmaf(xv.v[e], sc.v[e], sh.v[e]); y.v[e] = POOL ? relu6f(t) : t; }
  long oidx;
  bool arg[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) arg[e] = true;
  if (!POOL) {
    oidx = r * a.C + c0;
  } else {
    const int Ho = a.H / a.ph, Wo = a.W / a.pw;
    int w = (int)(r % a.W); long rr = r / a.W; int h = (int)(rr % a.H); long b = rr / a.H;
    int ho = h / a.ph, wo = w / a.pw;
    if (ho >= Ho || wo >= Wo) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) out.v[e] = 0.f;
      return out;
    }
    int si = h - ho * a.ph, sj = w - wo * a.pw;
    for (int ii = 0; ii < a.ph; ++ii)
      for (int j = 0; j < a.pw; ++j) {
        if (ii == si && j == sj) continue;
        VecF<VEC> o = vload_s<VEC, NT>(&a.x[(((long)b * a.H + ho * a.ph + ii) * a.W + wo * a.pw + j) * a.C + c0]);
        bool earlier = (ii < si) || (ii == si && j < sj);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float oy = relu6f(fmaf(o.v[e], sc.v[e], sh.v[e]));
          if (earlier ? (oy >= y.v[e]) : (oy > y.v[e])) arg[e] = false;   // not the first maximum
        }
      }
    oidx = (((long)b * Ho + ho) * Wo + wo) * a.C + c0;
  }
  VecF<VEC> gv = vload_s<VEC, NT>(&a.g[oidx]);
  float dm[VEC];
  drop_scale_vec<VEC>(a.seed, a.layer, (uint64_t)oidx, a.rate, inv_keep, dm);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    bool live = arg[e] && (y.v[e] > 0.f) && (y.v[e] < 6.f);
    out.v[e] = live ? gv.v[e] * dm[e] : 0.f;
  }
  return out;
}

template <int PASS, int VEC, bool POOL, typename T>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgsT<T> a, float* __restrict__ partials,
                                                     const float* __restrict__ coef, T* __restrict__ dx, int CW,
                                                     int rows_per_chunk) {
  __shared__ float red[2][256 * VEC];
  const int tid = threadIdx.x, cl = tid % CW, rt = tid / CW, RT = 256 / CW;
  const long M = (long)a.B * a.H * a.W;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  const int CL = a.C / VEC;
  const float inv_keep = a.rate > 0.f ? 1.f / (1.f - a.rate) : 1.f;
  for (int cb = 0; cb < CL; cb += CW) {
    const int c = cb + cl, c0 = c * VEC;
    VecF<VEC> s, q;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s.v[e] = 0.f; q.v[e] = 0.f; }
    if (c < CL) {
      VecF<VEC> mu = vload<VEC>(a.bnstate + c0), var = vload<VEC>(a.bnstate + a.C + c0);
      VecF<VEC> sc = vload<VEC>(a.bnstate + 2 * a.C + c0), sh = vload<VEC>(a.bnstate + 3 * a.C + c0);
      VecF<VEC> inv, c1, c2;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { inv.v[e] = 1.0f / sqrtf(var.v[e] + BN_EPS); c1.v[e] = 0.f; c2.v[e] = 0.f; }
      VecF<VEC> Pc, Qc;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { Pc.v[e] = 0.f; Qc.v[e] = 0.f; }
      if (PASS == 2) {
        c1 = vload<VEC>(coef + c0); c2 = vload<VEC>(coef + a.C + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) bn_bwd_pq(sc.v[e], c1.v[e], c2.v[e], mu.v[e], inv.v[e], Pc.v[e], Qc.v[e]);
      }
      long r = r0 + rt;
#if CRNN_BNB_ROWS4
      if (PASS == 1 && !POOL) {
        for (; r + 3L * RT < r1; r += 4L * RT) {   // four rows in flight: 8 independent 16-byte loads per thread
          VecF<VEC> xr[4], gr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) xr[u] = vload<VEC>(&a.x[(r + u * RT) * a.C + c0]);
#pragma unroll
          for (int u = 0; u < 4; ++u) gr[u] = bn_gy_vec<VEC, POOL, T>(a, r + u * RT, c0, xr[u], sc, sh, inv_keep);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { s.v[e] += gr[u].v[e]; q.v[e] = fmaf(gr[u].v[e], (xr[u].v[e] - mu.v[e]) * inv.v[e], q.v[e]); }
        }
      }
#endif
      constexpr bool NT = CRNN_NT_BN_BWD2 && PASS == 2;
      for (; r + RT < r1; r += 2L * RT) {   // two rows in flight (independent loads)
        VecF<VEC> xa = vload_s<VEC, NT>(&a.x[r * a.C + c0]);
        VecF<VEC> xb = vload_s<VEC, NT>(&a.x[(r + RT) * a.C + c0]);
        VecF<VEC> ga = bn_gy_vec<VEC, POOL, T, NT>(a, r, c0, xa, sc, sh, inv_keep);
        VecF<VEC> gb = bn_gy_vec<VEC, POOL, T, NT>(a, r + RT, c0, xb, sc, sh, inv_keep);
        VecF<VEC> oa, ob;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float ha = (xa.v[e] - mu.v[e]) * inv.v[e], hb = (xb.v[e] - mu.v[e]) * inv.v[e];
          if (PASS == 1) {
            s.v[e] += ga.v[e]; q.v[e] = fmaf(ga.v[e], ha, q.v[e]);
            s.v[e] += gb.v[e]; q.v[e] = fmaf(gb.v[e], hb, q.v[e]);
          } else {
            oa.v[e] = bn_bwd_dx_pq(xa.v[e], ga.v[e], sc.v[e], Pc.v[e], Qc.v[e]);
            ob.v[e] = bn_bwd_dx_pq(xb.v[e], gb.v[e], sc.v[e], Pc.v[e], Qc.v[e]);
          }
        }
        if (PASS == 2) { bn_store_dx<VEC, T>(a, dx, r * a.C + c0, oa); bn_store_dx<VEC, T>(a, dx, (r + RT) * a.C + c0, ob); }
      }
      for (; r < r1; r += RT) {
        VecF<VEC> xv = vload_s<VEC, NT>(&a.x[r * a.C + c0]);
        VecF<VEC> gy = bn_gy_vec<VEC, POOL, T, NT>(a, r, c0, xv, sc, sh, inv_keep);
        VecF<VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float xh = (xv.v[e] - mu.v[e]) * inv.v[e];
          if (PASS == 1) { s.v[e] += gy.v[e]; q.v[e] = fmaf(gy.v[e], xh, q.v[e]); }
          else o.v[e] = bn_bwd_dx_pq(xv.v[e], gy.v[e], sc.v[e], Pc.v[e], Qc.v[e]);
        }
        if (PASS == 2) bn_store_dx<VEC, T>(a, dx, r * a.C + c0, o);
      }
    }
    if (PASS == 1) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) { red[0][tid * VEC + e] = s.v[e]; red[1][tid * VEC + e] = q.v[e]; }
      __syncthreads();
      if (rt == 0 && c < CL) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float s2 = 0.f, q2 = 0.f;
          for (int r = 0; r < RT; ++r) { s2 += red[0][(r * CW + cl) * VEC + e]; q2 += red[1][(r * CW + cl) * VEC + e]; }
          partials[((long)blockIdx.x * 2 + 0) * a.C + c0 + e] = s2;
          partials[((long)blockIdx.x * 2 + 1) * a.C + c0 + e] = q2;
        }
      }
    }
  }
}

// Pooled variant (H % ph == 0, W % pw == 0): one thread owns a whole pool window x VEC channels, so every
// pre-BN value is read exactly once per pass and the arg-max is found once per window.
// PHT x PWT: the window as compile-time constants (0 x 0: run-time a.ph x a.pw) -- the window loops unroll without guards and the index
// arithmetic has no run-time divisions: a thread decomposes its first window index once and then steps (b, ho, wo) by RT windows.
template <int PASS, int VEC, typename T, int PHT = 0, int PWT = 0>
__global__ __launch_bounds__(256) void bn_bwd_pool_kernel(BnBwdArgsT<T> a, float* __restrict__ partials,
                                                          const float* __restrict__ coef, T* __restrict__ dx, int CW,
                                                          int rows_per_chunk) {
  __shared__ float red[2][256 * VEC];
  const int tid = threadIdx.x, cl = tid % CW, rt = tid / CW, RT = 256 / CW;
  const int ph = PHT ? PHT : a.ph, pw = PWT ? PWT : a.pw;
  const int Ho = a.H / ph, Wo = a.W / pw;
  const long Mo = (long)a.B * Ho * Wo;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > Mo) r1 = Mo;
  const int CL = a.C / VEC;
  const float inv_keep = a.rate > 0.f ? 1.f / (1.f - a.rate) : 1.f;
  const int nwin = ph * pw;       // <= 4
  // first window of this thread: (b, ho, wo), stepped by RT windows per iteration
  struct Pos { int b, ho, wo; };
  auto pos_of = [&](long r) { Pos p; p.wo = (int)(r % Wo); const long rr = r / Wo; p.ho = (int)(rr % Ho); p.b = (int)(rr / Ho); return p; };
  auto advance = [&](Pos& p, int n) {
    p.wo += n;
    while (p.wo >= Wo) { p.wo -= Wo; if (++p.ho == Ho) { p.ho = 0; ++p.b; } }
  };
  for (int cb = 0; cb < CL; cb += CW) {
    const int c = cb + cl, c0i = c * VEC;
    VecF<VEC> s, q;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s.v[e] = 0.f; q.v[e] = 0.f; }
    if (c < CL) {
      VecF<VEC> mu = vload<VEC>(a.bnstate + c0i), var = vload<VEC>(a.bnstate + a.C + c0i);
      VecF<VEC> sc = vload<VEC>(a.bnstate + 2 * a.C + c0i), sh = vload<VEC>(a.bnstate + 3 * a.C + c0i);
      VecF<VEC> inv, c1, c2;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { inv.v[e] = 1.0f / sqrtf(var.v[e] + BN_EPS); c1.v[e] = 0.f; c2.v[e] = 0.f; }
      if (PASS == 2) { c1 = vload<VEC>(coef + c0i); c2 = vload<VEC>(coef + a.C + c0i); }
      // one pool window per step; pass 1 (reduce only) keeps two windows' loads in flight
      auto load_window = [&](long r, const Pos& p, VecF<VEC>& gv, VecF<VEC> (&xw)[4], long& xbase) {
        constexpr bool NT = CRNN_NT_BN_BWD2 && PASS == 2;
        gv = vload_s<VEC, NT>(&a.g[r * a.C + c0i]);
        xbase = (((long)p.b * a.H + p.ho * ph) * a.W + p.wo * pw) * a.C + c0i;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < nwin) { int ii = k / pw, j = k - ii * pw; xw[k] = vload_s<VEC, NT>(&a.x[xbase + ((long)ii * a.W + j) * a.C]); }
      };
      // These two kernels are VALU-bound (SQ counters: ~3 resident waves/SIMD each 27-34 % VALU-active), so the window is
      // processed with as few operations as the arithmetic allows: only the arg-max position carries gradient, hence
      //   pass 1: sum gy = gsel, sum gy*xhat = gsel * xhat(x at the arg-max)            (x tracked alongside the maximum)
      //   pass 2: dx_k = scale*(gy_k - c1 - xhat_k*c2) = fma(cx, x_k, c0) + [k == arg] * scale*gsel
      //           with the per-channel constants cx = -scale*c2*inv, c0 = -scale*c1 - cx*mu.
      VecF<VEC> cx, c0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { cx.v[e] = -sc.v[e] * c2.v[e] * inv.v[e]; c0.v[e] = -sc.v[e] * c1.v[e] - cx.v[e] * mu.v[e]; }
      auto do_window = [&](long r, const VecF<VEC>& gv, const VecF<VEC> (&xw)[4], long xbase) {
        const long oidx = r * a.C + c0i;
        // The scan runs on the UNCLAMPED BatchNorm value t: a window carries gradient only if its maximum lies in (0, 6), and then the
        // first maximum of ReLU6(t) is the first maximum of t (values below 0 clamp to 0 < max, none reaches 6); for every other
        // window gsel = 0 and neither xs nor arg is used.  Same bits, two VALU operations per element fewer.
        float best[VEC], xs[VEC]; int arg[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; xs[e] = 0.f; arg[e] = 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k < nwin) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              float y = fmaf(xw[k].v[e], sc.v[e], sh.v[e]);
              const bool up = y > best[e];                      // strict '>' keeps the FIRST maximum (scan order)
              best[e] = up ? y : best[e];
              if (PASS == 1) xs[e] = up ? xw[k].v[e] : xs[e]; else arg[e] = up ? k : arg[e];
            }
          }
        }
        float gsel[VEC], dm[VEC];
        drop_scale_vec<VEC>(a.seed, a.layer, (uint64_t)oidx, a.rate, inv_keep, dm);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          bool live = (best[e] > 0.f) && (best[e] < 6.f);
          gsel[e] = live ? gv.v[e] * dm[e] : 0.f;
        }
        if (PASS == 1) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            s.v[e] += gsel[e];
            q.v[e] = fmaf(gsel[e], (xs[e] - mu.v[e]) * inv.v[e], q.v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) gsel[e] *= sc.v[e];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < nwin) {
              VecF<VEC> o;
#pragma unroll
              for (int e = 0; e < VEC; ++e) o.v[e] = fmaf(cx.v[e], xw[k].v[e], c0.v[e]) + ((arg[e] == k) ? gsel[e] : 0.f);
              int ii = k / pw, j = k - ii * pw;
              bn_store_dx<VEC, T>(a, dx, xbase + ((long)ii * a.W + j) * a.C, o);
            }
          }
        }
      };
      long r = r0 + rt;
      Pos pa = pos_of(r < r1 ? r : r0);
      if (PASS == 1 && a.qmax) {   // the window's maximum is x * scale + shift of the saved x; four windows in flight
        auto one = [&](long rr, const VecF<VEC>& gv, const VecF<VEC>& xs) {
          float dm[VEC];
          drop_scale_vec<VEC>(a.seed, a.layer, (uint64_t)(rr * a.C + c0i), a.rate, inv_keep, dm);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float best = fmaf(xs.v[e], sc.v[e], sh.v[e]);
            const bool live = (best > 0.f) && (best < 6.f);
            const float gsel = live ? gv.v[e] * dm[e] : 0.f;
            s.v[e] += gsel;
            q.v[e] = fmaf(gsel, (xs.v[e] - mu.v[e]) * inv.v[e], q.v[e]);
          }
        };
        for (; r + 3L * RT < r1; r += 4L * RT) {
          VecF<VEC> gv[4], xs[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { gv[u] = vload<VEC>(&a.g[(r + u * RT) * a.C + c0i]); xs[u] = vload<VEC>(&a.qmax[(r + u * RT) * a.C + c0i]); }
#pragma unroll
          for (int u = 0; u < 4; ++u) one(r + u * RT, gv[u], xs[u]);
        }
        for (; r < r1; r += RT) {
          const VecF<VEC> gv = vload<VEC>(&a.g[r * a.C + c0i]), xs = vload<VEC>(&a.qmax[r * a.C + c0i]);
          one(r, gv, xs);
        }
      } else
      if (PASS == 1) {
        for (; r + RT < r1; r += 2L * RT) {
          VecF<VEC> ga, gb, xa[4], xb[4]; long ba, bb;
          Pos pb = pa; advance(pb, RT);
          load_window(r, pa, ga, xa, ba);
          load_window(r + RT, pb, gb, xb, bb);
          do_window(r, ga, xa, ba);
          do_window(r + RT, gb, xb, bb);
          pa = pb; advance(pa, RT);
        }
      }
      for (; r < r1; r += RT) {
        VecF<VEC> gv, xw[4]; long xbase;
        load_window(r, pa, gv, xw, xbase);
        do_window(r, gv, xw, xbase);
        advance(pa, RT);
      }
    }
    if (PASS == 1) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) { red[0][tid * VEC + e] = s.v[e]; red[1][tid * VEC + e] = q.v[e]; }
      __syncthreads();
      if (rt == 0 && c < CL) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float s2 = 0.f, q2 = 0.f;
          for (int r = 0; r < RT; ++r) { s2 += red[0][(r * CW + cl) * VEC + e]; q2 += red[1][(r * CW + cl) * VEC + e]; }
          partials[((long)blockIdx.x * 2 + 0) * a.C + c0i + e] = s2;
          partials[((long)blockIdx.x * 2 + 1) * a.C + c0i + e] = q2;
        }
      }
    }
  }
}

// dgamma = sum gy*xhat, dbeta = sum gy; coef = [c1 | c2]
__global__ __launch_bounds__(RED_CH * RED_PL) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts, int C, double inv_n,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ double red[2][RED_PL][RED_CH];
  int c = blockIdx.x * RED_CH + threadIdx.x;
  double s = 0.0, q = 0.0;
  if (c < C) part_lane_sum2(partials + c, partials + C + c, 2L * C, nparts, threadIdx.y, s, q);
  red[0][threadIdx.y][threadIdx.x] = s; red[1][threadIdx.y][threadIdx.x] = q;
  lanes_sum2(red, s, q);
  if (threadIdx.y == 0 && c < C) {
    dbeta[c] = (float)s; dgamma[c] = (float)q;
    coef[c] = (float)(s * inv_n); coef[C + c] = (float)(q * inv_n);
  }
}

// Chunking of the two BatchNorm-backward passes: rows per chunk = the largest power of two <= 512 that still yields CRNN_BNB_MINCHUNKS
// chunks.  256 (468 partial rows for the 52-row maps at batch 256, 1872 for the 104-row maps) measured best over 256 / 512 / 768 / 1024 /
// 2048 (scripts/bnp1_bench.py): the statistics pass itself barely cares, the finalize that reads the partial rows halves; chunks of 1024
// rows (936 workgroups) cost the 104-row maps 6 % in the step.
#ifndef CRNN_BNB_MINCHUNKS
#define CRNN_BNB_MINCHUNKS 256
#endif
#ifndef CRNN_BNB_ROWS4
#define CRNN_BNB_ROWS4 1       // statistics pass without pooling: four rows (8 independent 16-byte loads) in flight per thread; 0 = two
#endif
static inline int bn_bwd_rows_per_chunk(long rows) { long r = 512; while (r > 16 && rows / r < CRNN_BNB_MINCHUNKS) r >>= 1; return (int)r; }
// upper bound on the number of partial rows for a [M][C] BN backward (pooled variants iterate over M/2 or M/4 windows)
extern "C" int crnn_bn_bwd_chunks(long M) {
  int best = 0;
  for (int div = 1; div <= 4; div *= 2) { int c = cdiv(M / div, bn_bwd_rows_per_chunk(M / div)); if (c > best) best = c; }
  return best;
}

template <int VEC, bool POOL, typename T>
static int bn_bwd_launch(const BnBwdArgsT<T>& a, T* dx, float* dgamma, float* dbeta, float* parts, float* coef, hipStream_t stream, bool apply_only = false) {
  const long M = (long)a.B * a.H * a.W;
  const int CL = a.C / VEC;
  const int CW = pow2_ge(CL < 256 ? CL : 256);
  const bool window = POOL && (a.H % a.ph == 0) && (a.W % a.pw == 0) && (a.ph * a.pw <= 4);
  const long rows = window ? M / (a.ph * a.pw) : M;
  const int rpc = bn_bwd_rows_per_chunk(rows), chunks = cdiv(rows, rpc);
  const int pk = (a.ph == 2 && a.pw == 2) ? 1 : ((a.ph == 1 && a.pw == 2) ? 2 : 0);      // the CRNN's two pool shapes as compile-time windows
  if (!apply_only) {   // (apply_only: coef comes from elsewhere -- the statistics were taken by the kernel that produced g)
  if (window && pk == 1) hipLaunchKernelGGL((bn_bwd_pool_kernel<1, VEC, T, 2, 2>), dim3(chunks), dim3(256), 0, stream, a, parts, (const float*)nullptr, (T*)nullptr, CW, rpc);
  else if (window && pk == 2) hipLaunchKernelGGL((bn_bwd_pool_kernel<1, VEC, T, 1, 2>), dim3(chunks), dim3(256), 0, stream, a, parts, (const float*)nullptr, (T*)nullptr, CW, rpc);
  else if (window) hipLaunchKernelGGL((bn_bwd_pool_kernel<1, VEC, T>), dim3(chunks), dim3(256), 0, stream, a, parts, (const float*)nullptr, (T*)nullptr, CW, rpc);
  else hipLaunchKernelGGL((bn_bwd_kernel<1, VEC, POOL, T>), dim3(chunks), dim3(256), 0, stream, a, parts, (const float*)nullptr, (T*)nullptr, CW, rpc);
  CRNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(a.C, RED_CH)), dim3(RED_CH, RED_PL), 0, stream, parts, chunks, a.C, 1.0 / (double)M, dgamma, dbeta, coef);
  CRNN_LAUNCH_CHECK();
  }
  if (dx == nullptr && a.dxp == nullptr) return CRNN_OK;   // statistics only: dgamma, dbeta and coef = [mean(gy) | mean(gy * xhat)]; the caller applies pass 2 itself
  if (window && pk == 1) hipLaunchKernelGGL((bn_bwd_pool_kernel<2, VEC, T, 2, 2>), dim3(chunks), dim3(256), 0, stream, a, (float*)nullptr, coef, dx, CW, rpc);
  else if (window && pk == 2) hipLaunchKernelGGL((bn_bwd_pool_kernel<2, VEC, T, 1, 2>), dim3(chunks), dim3(256), 0, stream, a, (float*)nullptr, coef, dx, CW, rpc);
  else if (window) hipLaunchKernelGGL((bn_bwd_pool_kernel<2, VEC, T>), dim3(chunks), dim3(256), 0, stream, a, (float*)nullptr, coef, dx, CW, rpc);
  else hipLaunchKernelGGL((bn_bwd_kernel<2, VEC, POOL, T>), dim3(chunks), dim3(256), 0, stream, a, (float*)nullptr, coef, dx, CW, rpc);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

template <typename T>
static int bn_bwd_typed(const T* x, const T* g, const float* bnstate, const float* gamma, T* dx, float* dgamma, float* dbeta,
                        float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed,
                        uint32_t layer, hipStream_t stream, bool apply_only = false, void* dx_planes = nullptr, long plane_stride = 0, int planes = 0, const T* qmax = nullptr) {
  BnBwdArgsT<T> a{x, g, bnstate, gamma, B, H, W, C, ph, pw, rate, seed, layer};
  if (qmax) {   // (the pooled kernel's compile-time windows only)
    if (!((ph == 2 && pw == 2) || (ph == 1 && pw == 2)) || H % ph || W % pw || ((uintptr_t)qmax & 15)) return CRNN_ERR_ARG;
    a.qmax = qmax;
  }
  if (dx_planes) {   // fp32 tensors, four channels per thread, the planes 8-byte aligned
    if (sizeof(T) != 4 || dx || (planes != 2 && planes != 3) || (C & 3) || ((uintptr_t)dx_planes & 7) || (plane_stride & 3) || plane_stride < (long)B * H * W * C ||
        (((uintptr_t)x | (uintptr_t)g | (uintptr_t)bnstate | (uintptr_t)coef) & 15)) return CRNN_ERR_ARG;
    a.dxp = reinterpret_cast<unsigned short*>(dx_planes); a.dxps = plane_stride; a.dxpl = planes;
  }
  const bool pool = (ph * pw) > 1;
  const bool al = ((((uintptr_t)x | (uintptr_t)g | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef) & 15) == 0);   // (dx may be null: statistics only)
  const bool vec = (C % 4 == 0) && al;
  if (VecMax<T>::value == 8 && al && C % 8 == 0)
    return pool ? bn_bwd_launch<VecMax<T>::value, true, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only)
                : bn_bwd_launch<VecMax<T>::value, false, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only);
  if (vec) return pool ? bn_bwd_launch<4, true, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only)
                       : bn_bwd_launch<4, false, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only);
  return pool ? bn_bwd_launch<1, true, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only)
              : bn_bwd_launch<1, false, T>(a, dx, dgamma, dbeta, scratch_partials, coef, stream, apply_only);
}

// Full BN backward through Dropout/MaxPool/ReLU6: writes dx [B,H,W,C], dgamma[C], dbeta[C].
// scratch_partials: [crnn_bn_bwd_chunks(B*H*W)][2][C]; coef: [2*C].  dtype = storage of x, g and dx.
extern "C" int crnn_bn_bwd_ex(const void* x, const void* g, const float* bnstate, const float* gamma, void* dx,
                              float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C,
                              int ph, int pw, float rate, uint64_t seed, uint32_t layer, int dtype, hipStream_t stream) {
  if (dtype == CRNN_BF16)
    return bn_bwd_typed<bf16_t>((const bf16_t*)x, (const bf16_t*)g, bnstate, gamma, (bf16_t*)dx, dgamma, dbeta, scratch_partials, coef, B, H, W,
                                C, ph, pw, rate, seed, layer, stream);
  return bn_bwd_typed<float>((const float*)x, (const float*)g, bnstate, gamma, (float*)dx, dgamma, dbeta, scratch_partials, coef, B, H, W, C,
                             ph, pw, rate, seed, layer, stream);
}
// Pass 2 of crnn_bn_bwd_ex alone: dx = scale * (gy - coef[0..C) - xhat * coef[C..2C)) with coef = [mean(gy) | mean(gy * xhat)] from crnn_bn_bwd_finalize
// (the statistics taken by the kernel that produced g: crnn_gemm_wres_bf16_bnstats, crnn_gemm_f32x3_bnstats).  Same arguments otherwise.
extern "C" int crnn_bn_bwd_apply_ex(const void* x, const void* g, const float* bnstate, const float* coef, void* dx, int B, int H, int W, int C,
                                    int ph, int pw, float rate, uint64_t seed, uint32_t layer, int dtype, hipStream_t stream) {
  if (!x || !g || !bnstate || !coef || !dx) return CRNN_ERR_ARG;
  if (dtype == CRNN_BF16)
    return bn_bwd_typed<bf16_t>((const bf16_t*)x, (const bf16_t*)g, bnstate, nullptr, (bf16_t*)dx, nullptr, nullptr, nullptr, const_cast<float*>(coef), B, H, W,
                                C, ph, pw, rate, seed, layer, stream, true);
  return bn_bwd_typed<float>((const float*)x, (const float*)g, bnstate, nullptr, (float*)dx, nullptr, nullptr, nullptr, const_cast<float*>(coef), B, H, W, C,
                             ph, pw, rate, seed, layer, stream, true);
}
// crnn_bn_bwd_ex / crnn_bn_bwd_apply_ex for fp32 tensors with dx written as bf16 planes (planes = 2 | 3; plane pl of dx[i] at dx_planes[pl * plane_stride + i], the
// words of crnn_split3_planes of the fp32 dx the entry points above write) -- the operand format of crnn_gemm_pres_bnstats and crnn_pwconv_bnrelu6_wgrad_planes_stream_gp.
extern "C" int crnn_bn_bwd_planes_ex(const float* x, const float* g, const float* bnstate, const float* gamma, void* dx_planes, long plane_stride, int planes,
                                     float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate,
                                     uint64_t seed, uint32_t layer, hipStream_t stream) {
  if (!x || !g || !bnstate || !dx_planes || !coef) return CRNN_ERR_ARG;
  return bn_bwd_typed<float>(x, g, bnstate, gamma, nullptr, dgamma, dbeta, scratch_partials, coef, B, H, W, C, ph, pw, rate, seed, layer, stream, false, dx_planes,
                             plane_stride, planes);
}
extern "C" int crnn_bn_bwd_apply_planes_ex(const float* x, const float* g, const float* bnstate, const float* coef, void* dx_planes, long plane_stride, int planes,
                                           int B, int H, int W, int C, int ph, int pw, float rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  if (!x || !g || !bnstate || !coef || !dx_planes) return CRNN_ERR_ARG;
  return bn_bwd_typed<float>(x, g, bnstate, nullptr, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(coef), B, H, W, C, ph, pw, rate, seed, layer, stream, true,
                             dx_planes, plane_stride, planes);
}
// The general form: crnn_bn_bwd_ex with qmax (may be NULL; storage `dtype` like x: the forward's crnn_bn_act_pool_drop_qmax_ex output -- the statistics pass of a
// pooled BatchNorm then reads one value per window, same sums bit for bit) and dx either as a tensor of `dtype` or (fp32 only, dx NULL) as `planes` bf16 planes.
extern "C" int crnn_bn_bwd_qmax_ex(const void* x, const void* qmax, const void* g, const float* bnstate, const float* gamma, void* dx, void* dx_planes, long plane_stride,
                                   int planes, float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C, int ph, int pw, float rate,
                                   uint64_t seed, uint32_t layer, int dtype, hipStream_t stream) {
  if (!x || !g || !bnstate || !coef || (dx && dx_planes)) return CRNN_ERR_ARG;
  if (dtype == CRNN_BF16) {
    if (dx_planes) return CRNN_ERR_ARG;
    return bn_bwd_typed<bf16_t>((const bf16_t*)x, (const bf16_t*)g, bnstate, gamma, (bf16_t*)dx, dgamma, dbeta, scratch_partials, coef, B, H, W, C, ph, pw, rate, seed,
                                layer, stream, false, nullptr, 0, 0, (const bf16_t*)qmax);
  }
  return bn_bwd_typed<float>((const float*)x, (const float*)g, bnstate, gamma, (float*)dx, dgamma, dbeta, scratch_partials, coef, B, H, W, C, ph, pw, rate, seed, layer,
                             stream, false, dx_planes, plane_stride, planes, (const float*)qmax);
}
// Second stage of a BatchNorm backward whose statistics pass ran elsewhere (crnn_gemm_wres_bf16_bnstats): partials [nparts][2][C] = partial
// sums of gy and gy * xhat over `count` elements per channel -> dgamma, dbeta and coef = [mean(gy) | mean(gy * xhat)] (what
// crnn_dwconv3x3_bwd_stream / crnn_dwconv3x3_bwd_fused apply).  The finalize launch of crnn_bn_bwd_ex, on its own.
extern "C" int crnn_bn_bwd_finalize(const float* partials, int nparts, int C, long count, float* dgamma, float* dbeta, float* coef, hipStream_t stream) {
  if (!partials || nparts < 1 || C < 1 || count < 1 || !dgamma || !dbeta || !coef) return CRNN_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, RED_CH)), dim3(RED_CH, RED_PL), 0, stream, partials, nparts, C, 1.0 / (double)count, dgamma, dbeta, coef);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
// Same, for long partial lists (one row per GEMM tile row: crnn_gemm_f32x3_bnstats): rows folded into CRNN_BN_FOLD_ROWS chunk sums first
// (scratch: CRNN_BN_FOLD_ROWS * 2 * C floats), as crnn_bn_finalize_folded does for the forward statistics.
extern "C" int crnn_bn_bwd_finalize_folded(const float* partials, int nparts, int C, long count, float* dgamma, float* dbeta, float* coef, float* scratch,
                                           hipStream_t stream) {
  if (nparts <= 1024 || scratch == nullptr) return crnn_bn_bwd_finalize(partials, nparts, C, count, dgamma, dbeta, coef, stream);
  hipLaunchKernelGGL(partials_sum_kernel, dim3(cdiv(2 * C, RED_CH), CRNN_BN_FOLD_ROWS), dim3(RED_CH, RED_PL), 0, stream, partials, nparts, 2 * C, scratch, 1.f);
  CRNN_LAUNCH_CHECK();
  return crnn_bn_bwd_finalize(scratch, CRNN_BN_FOLD_ROWS, C, count, dgamma, dbeta, coef, stream);
}
extern "C" int crnn_bn_bwd(const float* x, const float* g, const float* bnstate, const float* gamma, float* dx,
                           float* dgamma, float* dbeta, float* scratch_partials, float* coef, int B, int H, int W, int C,
                           int ph, int pw, float rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  return crnn_bn_bwd_ex(x, g, bnstate, gamma, dx, dgamma, dbeta, scratch_partials, coef, B, H, W, C, ph, pw, rate, seed, layer, CRNN_F32, stream);
}

// ---------------------------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------------------------
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}
extern "C" int crnn_add(const float* a, const float* b, float* o, long n, hipStream_t stream) {
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, stream, a, b, o, n);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// y[r*ldy + c] = x[r*ldx + c] * dropmask(idx = r*C + c)   (rows x C view, strided)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C, int ldx, int ldy,
                               float rate, uint64_t seed, uint32_t layer) {
  long n = rows * C;
  float inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i / C; int c = (int)(i % C);
    y[r * ldy + c] = x[r * ldx + c] * drop_scale(seed, layer, (uint64_t)i, rate, inv_keep);
  }
}
// four consecutive elements per thread: one 16-byte load / store and ONE hash (drop_scale_vec: a hash covers an aligned group of four
// element indices) instead of four of each; 32-bit index arithmetic.  C, ldx, ldy multiples of 4, 16-byte aligned, < 2^31 vectors.
__global__ void dropout_vec4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned nvec, int C4, int ldx, int ldy,
                                    float rate, uint64_t seed, uint32_t layer) {
  const float inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
    const unsigned r = i / (unsigned)C4, c4 = i - r * (unsigned)C4;
    const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ldx + 4 * c4);
    float dm[4];
    drop_scale_vec<4>(seed, layer, (uint64_t)i * 4, rate, inv_keep, dm);
    *reinterpret_cast<float4*>(y + (long)r * ldy + 4 * c4) = make_float4(v.x * dm[0], v.y * dm[1], v.z * dm[2], v.w * dm[3]);
  }
}
extern "C" int crnn_dropout(const float* x, float* y, long rows, int C, int ldx, int ldy, float rate, uint64_t seed,
                            uint32_t layer, hipStream_t stream) {
  long n = rows * C;
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && n / 4 + 4096L * 256 < (1L << 31) && n > 0) {
    int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_vec4_kernel, dim3(blocks), dim3(256), 0, stream, x, y, (unsigned)(n / 4), C / 4, ldx, ldy, rate, seed, layer);
    CRNN_LAUNCH_CHECK();
    return CRNN_OK;
  }
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, stream, x, y, rows, C, ldx, ldy, rate, seed, layer);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// eight consecutive elements per thread: two 16-byte loads / stores, ONE hash, and the group's keep byte (bit e: element 8 g + e is kept -- the table
// crnn_dropout_keep_bytes writes) next to the dropped activations: dense2's one-pass backward (dense.hip) reads the decisions instead of hashing again
__global__ void dropout_vec8_keep_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ keep, unsigned ngroups, int C8,
                                         int ldx, int ldy, float rate, uint64_t seed, uint32_t layer) {
  const float inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < ngroups; i += gridDim.x * blockDim.x) {
    const unsigned r = i / (unsigned)C8, c8 = i - r * (unsigned)C8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + (long)r * ldx + 8 * c8), v1 = *reinterpret_cast<const float4*>(x + (long)r * ldx + 8 * c8 + 4);
    float dm[8];
    drop_scale_vec<8>(seed, layer, (uint64_t)i * 8, rate, inv_keep, dm);
    *reinterpret_cast<float4*>(y + (long)r * ldy + 8 * c8) = make_float4(v0.x * dm[0], v0.y * dm[1], v0.z * dm[2], v0.w * dm[3]);
    *reinterpret_cast<float4*>(y + (long)r * ldy + 8 * c8 + 4) = make_float4(v1.x * dm[4], v1.y * dm[5], v1.z * dm[6], v1.w * dm[7]);
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) m |= (dm[e] != 0.f ? 1u : 0u) << e;
    keep[i] = (unsigned char)m;
  }
}
extern "C" int crnn_dropout_keep(const float* x, float* y, void* keep, long rows, int C, int ldx, int ldy, float rate, uint64_t seed, uint32_t layer,
                                 hipStream_t stream) {
  const long n = rows * C;
  if (!x || !y || !keep || rate >= 1.f) return CRNN_ERR_ARG;
  if (C % 8 || ldx % 4 || ldy % 4 || (((uintptr_t)x | (uintptr_t)y) & 15) || n <= 0 || n / 8 + 4096L * 256 >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  int blocks = cdiv(n / 8, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_vec8_keep_kernel, dim3(blocks), dim3(256), 0, stream, x, y, static_cast<unsigned char*>(keep), (unsigned)(n / 8), C / 8, ldx, ldy, rate, seed, layer);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// materialise the multiplier (0 or 1/(1-rate)) that dropout site `layer` applies -- test hook
__global__ void dropout_mask_kernel(float* __restrict__ m, long n, float rate, uint64_t seed, uint32_t layer) {
  float inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    m[i] = drop_scale(seed, layer, (uint64_t)i, rate, inv_keep);
}
extern "C" int crnn_dropout_mask(float* m, long n, float rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, stream, m, n, rate, seed, layer);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}

// Keep bits of a dropout site, one BYTE per group of 8 consecutive elements (bit e: element 8 g + e is kept) -- what drop_scale /
// drop_scale_vec decide, in the form the prologue row-stream depthwise kernels read (dwconv_stream.hip, dwconv_bwd_stream.hip): the mask
// depends on (seed, site, index) only, so it is evaluated once per step here -- four groups per thread, their dropout words side by side --
// instead of once per consumer pass inside kernels whose transform waves have no spare issue slots (36 of ~100 operations per group).
struct KeepBatch { unsigned* out[CRNN_KEEP_BATCH_MAX]; long nwords[CRNN_KEEP_BATCH_MAX]; long ngroups[CRNN_KEEP_BATCH_MAX]; uint32_t layer[CRNN_KEEP_BATCH_MAX]; };
__global__ __launch_bounds__(256) void dropout_keep_bytes_kernel(KeepBatch b, uint32_t thr, uint64_t seed) {
  const int site = blockIdx.y;
  const crnn_rng_key key = crnn_rng_make_key(seed, b.layer[site]);
  unsigned* __restrict__ out = b.out[site];
  const long nwords = b.nwords[site], ngroups = b.ngroups[site];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x) {
    uint32_t w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) crnn_rng8(key, (uint64_t)(4 * i + j), w[j]);
    unsigned v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned m = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        m |= ((w[j][q] & 0xffffu) >= thr ? 1u : 0u) << (2 * q);
        m |= ((w[j][q] >> 16) >= thr ? 1u : 0u) << (2 * q + 1);
      }
      if (4 * i + j >= ngroups) m = 0;
      v |= m << (8 * j);
    }
    out[i] = v;
  }
}
// out: 4-byte aligned, (ngroups + 3) / 4 * 4 bytes are written (the tail bytes of the last word are zero); rate <= 0: every element kept (0xFF)
extern "C" int crnn_dropout_keep_bytes_batch(int n, void* const* out, const long* ngroups, const uint32_t* layer, float rate, uint64_t seed, hipStream_t stream) {
  if (n < 0 || n > CRNN_KEEP_BATCH_MAX || rate >= 1.f || (n && (!out || !ngroups || !layer))) return CRNN_ERR_ARG;
  KeepBatch b; long maxw = 0; int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!out[i] || ngroups[i] < 0 || ((uintptr_t)out[i] & 3)) return CRNN_ERR_ARG;
    if (ngroups[i] == 0) continue;
    const long nw = (ngroups[i] + 3) / 4;
    if (rate <= 0.f) {
      hipError_t e = hipMemsetAsync(out[i], 0xFF, (size_t)nw * 4, stream);
      if (e != hipSuccess) return (int)e;
      continue;
    }
    b.out[m] = (unsigned*)out[i]; b.nwords[m] = nw; b.ngroups[m] = ngroups[i]; b.layer[m] = layer[i]; ++m;
    if (nw > maxw) maxw = nw;
  }
  if (m == 0) return CRNN_OK;
  long blocks = (maxw + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_keep_bytes_kernel, dim3((unsigned)blocks, m), dim3(256), 0, stream, b, crnn_drop_threshold(rate), seed);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_dropout_keep_bytes(void* out, long ngroups, float rate, uint64_t seed, uint32_t layer, hipStream_t stream) {
  return crnn_dropout_keep_bytes_batch(1, &out, &ngroups, &layer, rate, seed, stream);
}

// g_out[perm(r)][c] = g[r][c] * [y[r][c] > 0]      (backward of ReLU, and of Dropout∘ReLU when y is the
// dropped activation: the 1/(1-p) factor is passed as `scale`).  permP as in the GEMM epilogue.
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ go, bf16_t* __restrict__ go16,
                                long rows, int C, float scale, int permP) {
  long n = rows * C;
  long Q = permP ? rows / permP : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i / C; int c = (int)(i % C);
    long orow = permP ? (r % permP) * Q + r / permP : r;
    const float v = (y[i] > 0.f) ? g[i] * scale : 0.f;
    go[orow * C + c] = v;
    if (go16) st1(go16 + orow * C + c, v);      // the same values rounded to bf16 (round to nearest even): the data-gradient GEMM's operand
  }
}
extern "C" int crnn_relu_bwd_ex(const float* y, const float* g, float* go, void* go_bf16, long rows, int C, float scale, int permP,
                                hipStream_t stream) {
  long n = rows * C;
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, y, g, go, static_cast<bf16_t*>(go_bf16), rows, C, scale, permP);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_relu_bwd(const float* y, const float* g, float* go, long rows, int C, float scale, int permP,
                             hipStream_t stream) {
  return crnn_relu_bwd_ex(y, g, go, nullptr, rows, C, scale, permP, stream);
}

// a = ReLU6(x*scale+shift) (the pointwise conv's input; kept for its weight gradient)
extern "C" int crnn_bn_act(const float* x, const float* bnstate, float* y, long M, int C, hipStream_t stream) {
  return crnn_bn_act_pool_drop(x, bnstate, y, 1, 1, (int)M, C, 1, 1, 0.f, 0, 0, stream);
}

// ---------------------------------------------------------------------------------------------
// Pointwise conv with ONE input channel (block 1: 1 -> 64, utils.py:64): an outer product, not a GEMM.
//   fwd   q[m][c] = a[m] * w[c]                 (+ per-128-row-tile column statistics of q as stored)
//   dgrad da[m]   = sum_c dq[m][c] * w[c]
//   wgrad dw[c]   = sum_m a[m] * dq[m][c]        (partials per 1024-row chunk, then crnn_partials_sum)
// a / da are fp32 (one channel is never stored as bf16), q / dq fp32 or bf16, N % 8 == 0, N <= 256.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pw1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w, T* __restrict__ q,
                                                      float* __restrict__ stats, long M, int N, const float* __restrict__ bn) {
  // bn != null (inference): q = ReLU6(a * w * scale + shift), the BatchNorm after the convolution folded in (bn = [mean|var|scale|shift])
  // one workgroup per 128-row tile (= one statistics row); thread = (8-column group, row lane)
  __shared__ float red[2][256 * 8];
  const int CG = N / 8, RT = 256 / CG, tid = threadIdx.x, cg = tid % CG, rt = tid / CG;
  const long r0 = (long)blockIdx.x * 128;
  float wv[8], s[8], ss[8];
  VecF<8> wl = vload<8>(w + 8 * cg);
  float bs[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { wv[e] = wl.v[e]; s[e] = 0.f; ss[e] = 0.f; bs[e] = 1.f; bt[e] = 0.f; }
  if (bn) {
    VecF<8> v1 = vload<8>(bn + 2 * N + 8 * cg), v2 = vload<8>(bn + 3 * N + 8 * cg);
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = v1.v[e]; bt[e] = v2.v[e]; }
  }
  if (rt < RT)
    for (int r = rt; r < 128 && r0 + r < M; r += RT) {
      const float av = a[r0 + r];
      VecF<8> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = av * wv[e];
      if (bn) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = relu6f(fmaf(o.v[e], bs[e], bt[e]));
      }
      vstore<8>(&q[(r0 + r) * N + 8 * cg], o);
      if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = o.v[e];
          if (sizeof(T) == 2) v = __uint_as_float(pack2_bf16(v, 0.f) << 16);   // the value the consumer reads back
          s[e] += v; ss[e] = fmaf(v, v, ss[e]);
        }
      }
    }
  if (!stats) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][tid * 8 + e] = s[e]; red[1][tid * 8 + e] = ss[e]; }
  __syncthreads();
  if (tid < 2 * N) {
    const int v = tid / N, c = tid % N;
    float acc = 0.f;
    for (int r = 0; r < RT; ++r) acc += red[v][(r * CG + c / 8) * 8 + (c & 7)];
    stats[((long)blockIdx.x * 2 + v) * N + c] = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pw1_dgrad_kernel(const T* __restrict__ dq, const float* __restrict__ w, float* __restrict__ da, long M, int N) {
  // CG lanes per row (8 columns each), rows = 256 / CG per pass; lanes of a row are combined with shuffles
  const int CG = N / 8, tid = threadIdx.x, cg = tid % CG;
  VecF<8> wl = vload<8>(w + 8 * cg);
  const long rows_per_block = 256 / CG;
  for (long r = (long)blockIdx.x * rows_per_block + tid / CG; r < M; r += (long)gridDim.x * rows_per_block) {
    VecF<8> g = vload<8>(&dq[r * N + 8 * cg]);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(g.v[e], wl.v[e], acc);
    for (int o = 1; o < CG; o <<= 1) acc += __shfl_xor(acc, o, 64);       // CG is a power of two <= 32 (N <= 256)
    if (cg == 0) da[r] = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pw1_wgrad_kernel(const float* __restrict__ a, const T* __restrict__ dq, float* __restrict__ partials,
                                                        long M, int N, int rows_per_chunk) {
  __shared__ float red[256 * 8];
  const int CG = N / 8, RT = 256 / CG, tid = threadIdx.x, cg = tid % CG, rt = tid / CG;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (rt < RT)
    for (long r = r0 + rt; r < r1; r += RT) {
      const float av = a[r];
      VecF<8> g = vload<8>(&dq[r * N + 8 * cg]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = fmaf(av, g.v[e], s[e]);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
  __syncthreads();
  if (tid < N) {
    float acc = 0.f;
    for (int r = 0; r < RT; ++r) acc += red[(r * CG + tid / 8) * 8 + (tid & 7)];
    partials[(long)blockIdx.x * N + tid] = acc;
  }
}
// ---- block 1 with its depthwise BatchNorm folded into the neighbours (round 4) ---------------------------------------------------
// The block's single-channel stage was nine launches of 5..12 us around 3.8 MB of data.  Training forward: the depthwise kernel takes its own
// statistics (dwconv_c1_fwd_kernel), the outer product applies BatchNorm-1 + ReLU6 to d on the way in (a = relu6(fma(d, scale, shift)): the stand-alone
// pass's arithmetic -- the activated tensor is never stored).  Backward: ONE pass over dq forms the weight gradient (from a re-formed the same way), the
// data gradient and the statistics of BatchNorm-1's backward (sum gy, sum gy * xhat with gy = da [0 < BN(d) < 6]) -- the stand-alone kernels read dq twice
// and d / da once more.
template <typename T>
__global__ __launch_bounds__(256) void pw1_bn_fwd_kernel(const float* __restrict__ d, const float* __restrict__ bn1, const float* __restrict__ w,
                                                         T* __restrict__ q, float* __restrict__ stats, long M, int N) {
  __shared__ float red[2][256 * 8];
  const int CG = N / 8, RT = 256 / CG, tid = threadIdx.x, cg = tid % CG, rt = tid / CG;
  const long r0 = (long)blockIdx.x * 128;
  const float sc = bn1[2], sh = bn1[3];                      // [mean | var | scale | shift] of the one-channel BatchNorm
  float wv[8], s[8], ss[8];
  VecF<8> wl = vload<8>(w + 8 * cg);
#pragma unroll
  for (int e = 0; e < 8; ++e) { wv[e] = wl.v[e]; s[e] = 0.f; ss[e] = 0.f; }
  if (rt < RT)
    for (int r = rt; r < 128 && r0 + r < M; r += RT) {
      const float av = relu6f(fmaf(d[r0 + r], sc, sh));
      VecF<8> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = av * wv[e];
      vstore<8>(&q[(r0 + r) * N + 8 * cg], o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = o.v[e];
        if (sizeof(T) == 2) v = __uint_as_float(pack2_bf16(v, 0.f) << 16);   // the value the consumer reads back
        s[e] += v; ss[e] = fmaf(v, v, ss[e]);
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][tid * 8 + e] = s[e]; red[1][tid * 8 + e] = ss[e]; }
  __syncthreads();
  if (tid < 2 * N) {
    const int v = tid / N, c = tid % N;
    float acc = 0.f;
    for (int r = 0; r < RT; ++r) acc += red[v][(r * CG + c / 8) * 8 + (c & 7)];
    stats[((long)blockIdx.x * 2 + v) * N + c] = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pw1_bn_bwd_kernel(const float* __restrict__ d, const float* __restrict__ bn1, const float* __restrict__ w,
                                                         const T* __restrict__ dq, float* __restrict__ da, float* __restrict__ partials,
                                                         float* __restrict__ bnparts, long M, int N, int rows_per_chunk) {
  __shared__ float red[256 * 8];
  __shared__ float sred[2][256];
  const int CG = N / 8, RT = 256 / CG, tid = threadIdx.x, cg = tid % CG, rt = tid / CG;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  long r1 = r0 + rows_per_chunk; if (r1 > M) r1 = M;
  const float mu = bn1[0], inv = 1.0f / sqrtf(bn1[1] + BN_EPS), sc = bn1[2], sh = bn1[3];
  VecF<8> wl = vload<8>(w + 8 * cg);
  float s[8], sg = 0.f, sq = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (rt < RT)
    for (long r = r0 + rt; r < r1; r += RT) {
      const float dv = d[r], t = fmaf(dv, sc, sh), av = relu6f(t);
      VecF<8> g = vload<8>(&dq[r * N + 8 * cg]);
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = fmaf(av, g.v[e], s[e]); acc = fmaf(g.v[e], wl.v[e], acc); }
      for (int o = 1; o < CG; o <<= 1) acc += __shfl_xor(acc, o, 64);       // CG is a power of two <= 32: the lanes of a row (pw1_dgrad_kernel's sum)
      if (cg == 0) {
        da[r] = acc;
        const float gy = (t > 0.f && t < 6.f) ? acc : 0.f;
        sg += gy; sq = fmaf(gy, (dv - mu) * inv, sq);
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
  sred[0][tid] = sg; sred[1][tid] = sq;
  __syncthreads();
  if (tid < N) {
    float acc = 0.f;
    for (int r = 0; r < RT; ++r) acc += red[(r * CG + tid / 8) * 8 + (tid & 7)];
    partials[(long)blockIdx.x * N + tid] = acc;
  }
  if (tid < 2) {                                                          // the chunk's two BatchNorm-backward sums, row lanes in order
    float acc = 0.f;
    for (int r = 0; r < RT; ++r) acc += sred[tid][r * CG];
    bnparts[(long)blockIdx.x * 2 + tid] = acc;
  }
}
// one-channel depthwise 3x3 forward with the statistics of its outputs: partial sums / sums of squares per block of 1024 pixels, [blocks][2]
__global__ __launch_bounds__(256) void dwconv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ out,
                                                            float* __restrict__ stats, int B, int H, int W) {
  __shared__ float red[2][256];
  const long total = (long)B * H * W;
  float kk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) kk[t] = k[t];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = (long)blockIdx.x * 1024 + u * 256 + threadIdx.x;
    if (i < total) {
      const int w = (int)(i % W); const long r = i / W; const int h = (int)(r % H); const long b = r / H;
      float a = 0.f;
#pragma unroll
      for (int ii = 0; ii < 3; ++ii)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int gh = h + ii - 1, gw = w + j - 1;
          if (gh >= 0 && gh < H && gw >= 0 && gw < W) a = fmaf(x[(b * H + gh) * W + gw], kk[ii * 3 + j], a);   // dwconv_naive_kernel's chain
        }
      out[i] = a;
      s += a; ss = fmaf(a, a, ss);
    }
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = ss;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x < 2) stats[(long)blockIdx.x * 2 + threadIdx.x] = red[threadIdx.x][0];
}
// one-channel depthwise stage backwards in one kernel: dd = BatchNorm-1 backward pass 2 of (d, da) (bn_bwd_kernel<2>'s arithmetic) formed ONCE per pixel of the
// block's 1024 pixels and a halo of W + 1 on either side into LDS, the input x beside it; dx = correlation of dd with the mirrored taps
// (dwconv_naive_kernel<0>, flip: the same chain), dk partials per block
#define C1B_PIX 1024
#define C1B_MAXW 255
__global__ __launch_bounds__(256) void dwconv_c1_bwd_kernel(const float* __restrict__ d, const float* __restrict__ da, const float* __restrict__ bn1,
                                                            const float* __restrict__ coef, const float* __restrict__ x, const float* __restrict__ k,
                                                            float* __restrict__ dx, float* __restrict__ partials, int B, int H, int W) {
  __shared__ float ddl[C1B_PIX + 2 * (C1B_MAXW + 1)];
  __shared__ float xl[C1B_PIX + 2 * (C1B_MAXW + 1)];
  __shared__ float red[9][256];
  const long total = (long)B * H * W;
  const float mu = bn1[0], inv = 1.0f / sqrtf(bn1[1] + BN_EPS), sc = bn1[2], sh = bn1[3];
  float P, Q;
  bn_bwd_pq(sc, coef[0], coef[1], mu, inv, P, Q);
  float kk[9], acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { kk[t] = k[t]; acc[t] = 0.f; }
  const long p0 = (long)blockIdx.x * C1B_PIX - (W + 1);            // first pixel of the staged range
  const int nst = C1B_PIX + 2 * (W + 1);
  for (int i = threadIdx.x; i < nst; i += 256) {
    const long p = p0 + i;
    float g = 0.f, xv = 0.f;
    if (p >= 0 && p < total) {
      const float dv = d[p], t = fmaf(dv, sc, sh);
      const float gy = (t > 0.f && t < 6.f) ? da[p] : 0.f;
      g = bn_bwd_dx_pq(dv, gy, sc, P, Q);
      xv = x[p];
    }
    ddl[i] = g; xl[i] = xv;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int li = u * 256 + threadIdx.x;
    const long p = (long)blockIdx.x * C1B_PIX + li;
    if (p < total) {
      const int w = (int)(p % W); const long r = p / W; const int h = (int)(r % H);
      const int c = li + W + 1;                                      // this pixel inside the staged range
      const float g0 = ddl[c];
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int gh = h + i - 1, gw = w + j - 1;
          if (gh >= 0 && gh < H && gw >= 0 && gw < W) {
            const int cn = c + (i - 1) * W + (j - 1);
            acc[i * 3 + j] = fmaf(xl[cn], g0, acc[i * 3 + j]);        // dwconv_wgrad_c1_kernel's term
            a = fmaf(ddl[cn], kk[8 - (i * 3 + j)], a);
          }
        }
      if (dx) dx[p] = a;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = acc[t];
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2)
#pragma unroll
      for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x < 9) partials[(long)blockIdx.x * 9 + threadIdx.x] = red[threadIdx.x][0];
}
static bool pw1_ok(int N, const void* q) { return N % 8 == 0 && N <= 256 && (N & (N - 1)) == 0 && ((uintptr_t)q & 15) == 0; }
// q [M][N] = a[M] (x) w[N]; stat_partials (may be NULL): [ceil(M/128)][2][N] like crnn_pwconv_fwd
static int pw1_fwd_launch(const float* a, const float* w, void* q, long M, int N, float* stat_partials, const float* bn, int dt_q, hipStream_t stream) {
  if (!pw1_ok(N, q) || M <= 0) return CRNN_ERR_UNSUPPORTED;
  const int blocks = cdiv(M, 128);
  if (dt_q == CRNN_BF16) hipLaunchKernelGGL(pw1_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, a, w, (bf16_t*)q, stat_partials, M, N, bn);
  else hipLaunchKernelGGL(pw1_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, a, w, (float*)q, stat_partials, M, N, bn);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_pw1_fwd(const float* a, const float* w, void* q, long M, int N, float* stat_partials, int dt_q, hipStream_t stream) {
  return pw1_fwd_launch(a, w, q, M, N, stat_partials, nullptr, dt_q, stream);
}
// inference: y = ReLU6(BN(a (x) w)) with out_bnstate = [mean|var|scale|shift] of the BatchNorm after the convolution (crnn_bn_infer_state)
extern "C" int crnn_pw1_fwd_folded(const float* a, const float* w, void* y, long M, int N, const float* out_bnstate, int dt_y, hipStream_t stream) {
  if (!out_bnstate) return CRNN_ERR_ARG;
  return pw1_fwd_launch(a, w, y, M, N, nullptr, out_bnstate, dt_y, stream);
}
// da[M] = dq[M][N] . w[N];   dw[N] = sum_m a[m] * dq[m][N]  (scratch: crnn_colreduce_chunks(M) * N floats)
extern "C" int crnn_pw1_bwd(const float* a, const float* w, const void* dq, float* da, float* dw, float* scratch, long M, int N,
                            int dt_q, hipStream_t stream) {
  if (!pw1_ok(N, dq) || M <= 0) return CRNN_ERR_UNSUPPORTED;
  const int rpc = colreduce_rpc(M), chunks = cdiv(M, rpc);
  if (dt_q == CRNN_BF16) hipLaunchKernelGGL(pw1_wgrad_kernel<bf16_t>, dim3(chunks), dim3(256), 0, stream, a, (const bf16_t*)dq, scratch, M, N, rpc);
  else hipLaunchKernelGGL(pw1_wgrad_kernel<float>, dim3(chunks), dim3(256), 0, stream, a, (const float*)dq, scratch, M, N, rpc);
  CRNN_LAUNCH_CHECK();
  CRNN_TRY(crnn_partials_sum(scratch, chunks, N, dw, 1.f, stream));
  if (da) {
    int blocks = cdiv(M, 256 / (N / 8)); if (blocks > 8192) blocks = 8192;
    if (dt_q == CRNN_BF16) hipLaunchKernelGGL(pw1_dgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dq, w, da, M, N);
    else hipLaunchKernelGGL(pw1_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dq, w, da, M, N);
    CRNN_LAUNCH_CHECK();
  }
  return CRNN_OK;
}
// Block 1's single-channel stage with BatchNorm-1 folded in (round 4; d, da fp32 [M]; in_bnstate = [mean|var|scale|shift] of the one-channel BatchNorm):
//   crnn_dwconv3x3_c1_fwd: out = dwconv3x3(x, k[9]) on [B][H][W] and [crnn_dwconv_c1_stat_rows][2] partial sums / sums of squares of out;
//   crnn_pw1_bn_fwd: q [M][N] = relu6(fma(d, scale, shift)) (x) w -- crnn_bn_act_pool_drop_ex + crnn_pw1_fwd without the activated tensor, same bits;
//   crnn_pw1_bn_bwd: dw [N], da [M] as crnn_pw1_bwd on the re-formed a (same bits), and bn_stat_partials [crnn_pw1_bn_bwd_rows][2] = partial sums of gy and
//                    gy * xhat for crnn_bn_bwd_finalize (C = 1), gy = da where 0 < BN(d) < 6; scratch: crnn_pw1_bn_bwd_rows(M) * N floats.
extern "C" int crnn_dwconv_c1_stat_rows(int B, int H, int W) { return cdiv((long)B * H * W, 1024); }
extern "C" int crnn_dwconv3x3_c1_fwd(const float* x, const float* k, float* out, float* stat_partials, int B, int H, int W, hipStream_t stream) {
  if (!x || !k || !out || !stat_partials || B <= 0 || H <= 0 || W <= 0) return CRNN_ERR_ARG;
  if ((long)B * H * W >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dwconv_c1_fwd_kernel, dim3(crnn_dwconv_c1_stat_rows(B, H, W)), dim3(256), 0, stream, x, k, out, stat_partials, B, H, W);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
extern "C" int crnn_pw1_bn_fwd(const float* d, const float* in_bnstate, const float* w, void* q, long M, int N, float* stat_partials, int dt_q,
                               hipStream_t stream) {
  if (!d || !in_bnstate || !w || !q || !stat_partials) return CRNN_ERR_ARG;
  if (!pw1_ok(N, q) || M <= 0) return CRNN_ERR_UNSUPPORTED;
  const int blocks = cdiv(M, 128);
  if (dt_q == CRNN_BF16) hipLaunchKernelGGL(pw1_bn_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, d, in_bnstate, w, (bf16_t*)q, stat_partials, M, N);
  else hipLaunchKernelGGL(pw1_bn_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, d, in_bnstate, w, (float*)q, stat_partials, M, N);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
//   crnn_dwconv3x3_c1_bwd: the one-channel depthwise stage backwards in one kernel -- crnn_bn_bwd_apply_ex (coef from crnn_bn_bwd_finalize) + crnn_dwconv3x3_wgrad_ex +
//                    crnn_dwconv3x3_fwd_ex(flip = 1): dx [B][H][W] (NULL: not wanted) bit-identical, dk [9] to the order of its partial sums; scratch: rows * 9 floats.
extern "C" int crnn_dwconv_c1_bwd_rows(int B, int H, int W) { return cdiv((long)B * H * W, 1024); }
extern "C" int crnn_dwconv3x3_c1_bwd(const float* d, const float* da, const float* bnstate, const float* coef, const float* x, const float* k, float* dx, float* dk,
                                     float* scratch, int B, int H, int W, hipStream_t stream) {
  if (!d || !da || !bnstate || !coef || !x || !k || !dk || !scratch || B <= 0 || H <= 0 || W <= 0) return CRNN_ERR_ARG;
  if ((long)B * H * W >= (1L << 31) || W > C1B_MAXW) return CRNN_ERR_UNSUPPORTED;
  const int rows = crnn_dwconv_c1_bwd_rows(B, H, W);
  hipLaunchKernelGGL(dwconv_c1_bwd_kernel, dim3(rows), dim3(256), 0, stream, d, da, bnstate, coef, x, k, dx, scratch, B, H, W);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, rows, 9, dk, 1.f, stream);
}
extern "C" int crnn_pw1_bn_bwd_rows(long M) { return cdiv(M, colreduce_rpc(M)); }
extern "C" int crnn_pw1_bn_bwd(const float* d, const float* in_bnstate, const float* w, const void* dq, float* da, float* dw, float* scratch,
                               float* bn_stat_partials, long M, int N, int dt_q, hipStream_t stream) {
  if (!d || !in_bnstate || !w || !dq || !da || !dw || !scratch || !bn_stat_partials) return CRNN_ERR_ARG;
  if (!pw1_ok(N, dq) || M <= 0 || N < 8) return CRNN_ERR_UNSUPPORTED;
  const int rpc = colreduce_rpc(M), chunks = cdiv(M, rpc);
  if (dt_q == CRNN_BF16) hipLaunchKernelGGL(pw1_bn_bwd_kernel<bf16_t>, dim3(chunks), dim3(256), 0, stream, d, in_bnstate, w, (const bf16_t*)dq, da, scratch, bn_stat_partials, M, N, rpc);
  else hipLaunchKernelGGL(pw1_bn_bwd_kernel<float>, dim3(chunks), dim3(256), 0, stream, d, in_bnstate, w, (const float*)dq, da, scratch, bn_stat_partials, M, N, rpc);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, chunks, N, dw, 1.f, stream);
}
