// EXPERIMENT (round 2, not built into libcrnn_mi355x.so): streaming depthwise kernel.  Bit-identical to the halo-tile kernel but slower
// (0.99 vs 0.79 ms over the step shapes): an ablation showed the depthwise kernels are issue-bound, not latency-bound -- with loads AND
// stores removed the 104x36x128 launch still takes 86 of 112 us -- so overlapping load and compute buys nothing.  See DESIGN.md section 4.
// Streaming depthwise 3x3 (DepthwiseConv2D(3x3, same, no bias), utils.py:44): forward (+ BatchNorm statistics of the
// output) and data gradient (flipped taps) of the conv stack, NHWC, fp32 or bf16 storage.
//
// The halo-tile kernel (conv.hip) alternates per workgroup between "fill the tile from HBM" and "compute from LDS"; with
// three workgroups per CU the HBM pipe idles whenever the residents happen to compute (0.49-0.55 of the 8 TB/s roof, issue
// 36 %).  Here the two phases of ONE workgroup overlap:
//   * a workgroup owns one 128-byte channel slab (32 fp32 / 64 bf16 channels) of a segment of image rows and streams the
//     rows top to bottom through an LDS ring of 2*BH + 2 row slots ((W + 2) pixels x 128 B each, ~48 KiB);
//   * a LOADER wave brings the BH rows of band j+1 in with global_load_lds (16 B per lane straight into the ring, no VGPRs,
//     out-of-image rows are written as zeros, the two halo columns are zeroed once) while the four COMPUTE waves work on
//     band j out of the slots that are already there: every input row is fetched once per workgroup (halo rows of a
//     segment boundary excepted) and one raw s_barrier per band hands a landed band over;
//   * only the loader waits on the vector-memory counter (s_waitcnt vmcnt(0) per band, its own LDS-DMA only), so the
//     compute waves' result stores retire in the background;
//   * the arithmetic is the halo-tile kernel's: each thread produces 3 adjacent pixels of 4 channels from one 3 x 5 window,
//     fp32 fma in tap order, bf16 results rounded once (RNE), statistics of the values as stored.
// Bit-identical outputs to conv.hip's dwconv_tile_kernel<0>; the statistics partials are per workgroup (segment) instead of
// per tile (crnn_dwconv_stat_rows).
#include "common.h"
#include <stdio.h>

namespace {

constexpr int kPXB = 3;
#ifdef CRNN_DW_TRACE
// trace build (scripts/dw_trace.py): `partials` is a u64 buffer; workgroup (0, 7) records 100-MHz timestamps per band
#define DW_TRACE(slot_) do { if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 7) \
    reinterpret_cast<unsigned long long*>(partials)[(DW_TRACE_BAND) * 8 + (slot_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DW_TRACE(slot_) do {} while (0)
#endif

template <typename T> struct SV;
template <> struct SV<float> { typedef float4 type; };
template <> struct SV<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ void widen4(const float4& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void widen4(const uint2& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct StreamGeom { int BH, nseg, SH, R; size_t lds; };

template <typename T>
__global__ __launch_bounds__(320) void dwconv_stream_kernel(const T* __restrict__ x, const float* __restrict__ k, T* __restrict__ out,
                                                            float* __restrict__ partials, int B, int H, int W, int C, int flip, int BH,
                                                            int nseg, int SH) {
  typedef typename SV<T>::type V;
  constexpr int VN = 4;
  constexpr int CL = 128 / sizeof(V);          // lanes per pixel (the 128-B slab): 8 fp32 / 16 bf16
  constexpr int PT = 256 / CL;                 // pixel threads among the 256 compute threads
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Wt = W + 2, RB = Wt * 128, R = 2 * BH + 2;
  const int cc0 = blockIdx.x * (128 / (int)sizeof(T));
  const int b = blockIdx.y / nseg, seg = blockIdx.y % nseg;
  const int y0 = seg * SH, y1 = min(H, y0 + SH);
  const int nb = (y1 - y0 + BH - 1) / BH;
  // the two halo columns of every ring slot stay zero for the whole kernel (LDS-DMA only writes pixels 1..W)
  for (int i = tid; i < R * 16; i += 320) {
    const int slot = i >> 4, c = i & 15;
    *reinterpret_cast<uint4*>(ring + slot * RB + (c < 8 ? 0 : (Wt - 1) * 128) + (c & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();

  if (wave == 4) {
    // ------------------------------------------------------------------ loader wave
    // the youngest wave of the workgroup loses the age-based issue arbitration against the VALU-dense compute waves sharing its
    // SIMD (measured: ~100 ns per global_load_lds at default priority, the whole band issue as long as the band's arithmetic)
    __builtin_amdgcn_s_setprio(3);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(x + ((long)b * H * W) * C + cc0);
    const long rowstride = (long)W * C * sizeof(T), pixstride = (long)C * sizeof(T);
    const int nch = W * 8;                                   // 16-byte chunks of one row's slab
    auto load_rows = [&](int r_lo, int r_hi) {               // input rows [r_lo, r_hi], relative slot = (r - (y0 - 1)) % R
      for (int r = r_lo; r <= r_hi; ++r) {
        unsigned char* dst = ring + ((r - (y0 - 1)) % R) * RB + 128;   // pixel 1 of the slot
        if (r < 0 || r >= H) {
          for (int ch = lane; ch < nch; ch += 64) *reinterpret_cast<uint4*>(dst + ch * 16) = make_uint4(0u, 0u, 0u, 0u);
        } else {
          const unsigned char* src = xb + r * rowstride;
          for (int c0 = 0; c0 < nch; c0 += 64) {
            const int ch = c0 + lane;
#ifdef CRNN_DW_TRACE
            if (flip & 4) continue;       // ablation: no loads
#endif
            if (ch < nch) glds16(src + (ch >> 3) * pixstride + (ch & 7) * 16, dst + c0 * 16);
          }
        }
      }
    };
    load_rows(y0 - 1, min(y1, y0 + BH));                     // band 0: its rows and both halo rows
    for (int j = 0; j < nb; ++j) {
#define DW_TRACE_BAND j
      DW_TRACE(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      DW_TRACE(1);
      __builtin_amdgcn_s_barrier();                          // band j is in LDS; the slots band j-1 alone used are free
      DW_TRACE(2);
      if (j + 1 < nb) load_rows(y0 + (j + 1) * BH + 1, min(y1, y0 + (j + 2) * BH));
      DW_TRACE(3);
#undef DW_TRACE_BAND
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef CRNN_DW_TRACE
    if (partials != nullptr) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
#endif
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int c4 = tid & (CL - 1), pt = tid / CL;
  float kw[9][VN];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ts = (flip & 1) ? 8 - t : t;
    const float4 w = *reinterpret_cast<const float4*>(&k[ts * C + cc0 + VN * c4]);
    kw[t][0] = w.x; kw[t][1] = w.y; kw[t][2] = w.z; kw[t][3] = w.w;
  }
  float s[VN], ss[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) { s[e] = 0.f; ss[e] = 0.f; }
  const int gpr = (W + kPXB - 1) / kPXB;
  T* ob = out + ((long)b * H * W) * C + cc0 + VN * c4;
  for (int j = 0; j < nb; ++j) {
#define DW_TRACE_BAND j
    if (wave == 0) DW_TRACE(4);
    __builtin_amdgcn_s_barrier();
    if (wave == 0) DW_TRACE(5);
    const int yb = y0 + j * BH;
    const int s0 = (j * BH) % R;                              // ring slot of this band's first window row (wave-uniform)
    const int rows = min(BH, y1 - yb);
    const int npg = rows * gpr;
    const int dly = PT / gpr, dlg = PT - dly * gpr;
    int ly = pt / gpr, lg = pt - ly * gpr;
    for (int pg = pt; pg < npg; pg += PT) {
      const int lx = lg * kPXB;
      const int gy = yb + ly;
      bool pv[kPXB];
#pragma unroll
      for (int e = 0; e < kPXB; ++e) pv[e] = lx + e < W;
      float a[kPXB][VN];
#pragma unroll
      for (int e = 0; e < kPXB; ++e)
#pragma unroll
        for (int c = 0; c < VN; ++c) a[e][c] = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int slot = s0 + ly + i;                                // window rows ly, ly+1, ly+2 of the band; ly + 2 <= BH + 1 < R
        slot = slot >= R ? slot - R : slot;
        const V* rowp = reinterpret_cast<const V*>(ring + slot * RB) + c4;
        float r[kPXB + 2][VN];
#pragma unroll
        for (int jj = 0; jj < kPXB + 2; ++jj) widen4(rowp[(lx + jj) * CL], r[jj]);
#pragma unroll
        for (int e = 0; e < kPXB; ++e)
#pragma unroll
          for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int c = 0; c < VN; ++c) a[e][c] = fmaf(r[e + jj][c], kw[i * 3 + jj][c], a[e][c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      const long o = ((long)gy * W + lx) * C;
#pragma unroll
      for (int e = 0; e < kPXB; ++e)
        if (pv[e]) {
#ifdef CRNN_DW_TRACE
          if (!(flip & 2) || a[e][0] == 1234.5f)   // ablation: no stores
#endif
          st4(ob + o + e * C, make_float4(a[e][0], a[e][1], a[e][2], a[e][3]));
#ifndef CRNN_DW_TRACE
          if (partials != nullptr) {
#pragma unroll
            for (int c = 0; c < VN; ++c) { s[c] += a[e][c]; ss[c] = fmaf(a[e][c], a[e][c], ss[c]); }
          }
#endif
        }
      lg += dlg; ly += dly;
      if (lg >= gpr) { lg -= gpr; ++ly; }
    }
    if (wave == 0) DW_TRACE(6);
#undef DW_TRACE_BAND
  }
#ifdef CRNN_DW_TRACE
  return;
#endif
  if (partials == nullptr) return;
  // per-workgroup statistics: pixel-threads of a wave by lane shuffles, the 4 compute waves through LDS
  __builtin_amdgcn_s_barrier();        // the ring is no longer read
  float* red = reinterpret_cast<float*>(ring);   // [4 waves][2][CL][VN]
#pragma unroll
  for (int e = 0; e < VN; ++e) {
#pragma unroll
    for (int o = CL; o < 64; o <<= 1) { s[e] += __shfl_xor(s[e], o, 64); ss[e] += __shfl_xor(ss[e], o, 64); }
  }
  if (lane < CL) {
#pragma unroll
    for (int e = 0; e < VN; ++e) { red[((wave * 2 + 0) * CL + c4) * VN + e] = s[e]; red[((wave * 2 + 1) * CL + c4) * VN + e] = ss[e]; }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < 2 * CL * VN; i += 256) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += red[w * 2 * CL * VN + i];
    const int v = i / (CL * VN), cch = i % (CL * VN);
    partials[((long)blockIdx.y * 2 + v) * C + cc0 + cch] = acc;
  }
}

StreamGeom stream_geom(int B, int H, int W, int C, int esize) {
  StreamGeom g;
  const int gpr = (W + kPXB - 1) / kPXB;
  g.BH = 48 / gpr; if (g.BH < 1) g.BH = 1; if (g.BH > H) g.BH = H;     // ~768 (pixel group, lane) items per band and 256 threads
  const int slabs = C / (128 / esize);
  // enough workgroups for three per CU, segments of at least two bands
  int nseg = cdiv(1536, (long)slabs * B);
  const int maxseg = H / (2 * g.BH) > 0 ? H / (2 * g.BH) : 1;
  if (nseg > maxseg) nseg = maxseg;
  if (nseg < 1) nseg = 1;
  g.SH = cdiv(H, nseg);
  g.nseg = cdiv(H, g.SH);
  g.R = 2 * g.BH + 2;
  g.lds = (size_t)g.R * (W + 2) * 128 + 128 * kPXB;    // + slack: the last pixel group of a ragged row reads past its slot
  if (g.lds < 4096) g.lds = 4096;
  return g;
}

}  // namespace

// 0 if (H, W, C, dtype) runs on the streaming kernel, else CRNN_ERR_UNSUPPORTED (the halo-tile kernel of conv.hip takes over)
extern "C" int crnn_dwconv_stream_supported(int H, int W, int C, int dtype) {
  const int esize = dtype == CRNN_BF16 ? 2 : 4;
  if (W < 3 || W > 62 || H < 2 || C % (128 / esize)) return CRNN_ERR_UNSUPPORTED;
  return CRNN_OK;
}
// rows of [2][C] statistics partials the depthwise forward writes for a (B, H, W, C) map in storage `dtype`
extern "C" int crnn_dwconv_stat_rows(int B, int H, int W, int C, int dtype) {
  if (crnn_dwconv_stream_supported(H, W, C, dtype) != CRNN_OK) return crnn_dwconv_num_tiles(B, H, W);
  return B * stream_geom(B, H, W, C, dtype == CRNN_BF16 ? 2 : 4).nseg;
}

extern "C" int crnn_dwconv3x3_stream(const void* x, const float* k, void* out, float* stat_partials, int B, int H, int W, int C, int flip,
                                     int dtype, hipStream_t stream) {
  CRNN_TRY(crnn_dwconv_stream_supported(H, W, C, dtype));
  if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)k) & 15)) return CRNN_ERR_UNSUPPORTED;
  const int esize = dtype == CRNN_BF16 ? 2 : 4;
  const StreamGeom g = stream_geom(B, H, W, C, esize);
  if (g.lds > 64 * 1024) return CRNN_ERR_UNSUPPORTED;
  dim3 grid(C / (128 / esize), B * g.nseg);
#ifdef CRNN_DW_TRACE
  { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)dwconv_stream_kernel<bf16_t>, 320, g.lds);
    printf("stream geom: BH %d nseg %d SH %d R %d lds %zu grid %d x %d  occupancy %d blocks/CU\n", g.BH, g.nseg, g.SH, g.R, g.lds, grid.x, grid.y, nb); }
#endif
  if (dtype == CRNN_BF16) {
    if (g.lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)dwconv_stream_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(dwconv_stream_kernel<bf16_t>, grid, dim3(320), g.lds, stream, (const bf16_t*)x, k, (bf16_t*)out, stat_partials, B, H, W, C, flip,
                       g.BH, g.nseg, g.SH);
  } else {
    if (g.lds > 48 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)dwconv_stream_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); if (e != hipSuccess) return (int)e; }
    hipLaunchKernelGGL(dwconv_stream_kernel<float>, grid, dim3(320), g.lds, stream, (const float*)x, k, (float*)out, stat_partials, B, H, W, C, flip,
                       g.BH, g.nseg, g.SH);
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
