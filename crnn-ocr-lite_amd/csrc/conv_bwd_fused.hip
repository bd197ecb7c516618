// Backward of one depthwise stage of a conv block in ONE kernel (bf16 storage, training) on halo tiles -- since the end of round 2 the fallback of
// the row-stream kernel of the same stage (dwconv_bwd_stream.hip: crnn_dwconv3x3_bwd_stream), taken for map widths that kernel refuses and under
// CRNN_FLAG_DW_TILE_KERNEL:
//     a = ReLU6(BatchNorm_1(d)),  d = DepthwiseConv2D(3x3)(x)          (utils.py:44-46)
// given da = dL/da (from the pointwise data-gradient GEMM), the BatchNorm statistics / coefficients (crnn_bn_bwd_ex with
// dx = null: first pass + finalize) it produces   dx = dL/dx   and   dk = dL/d(depthwise kernel).
//
// The unfused schedule moves the BatchNorm-input gradient dd three times through HBM: BatchNorm backward pass 2 (read d, da,
// write dd), depthwise weight gradient (read dd, x), depthwise data gradient (read dd, write dx) -- 7 tensor passes of the
// block's depthwise map.  Here the halo-tile fill forms dd = scale * (gy - c1 - xhat * c2), gy = da * [0 < BN(d) < 6], on the
// fly from d and da (rounded to bf16 exactly as the stored dd would be), keeps it only in LDS next to the x tile, and one
// pass over the pixel groups produces both gradients: 4 tensor passes (d, da, x in; dx out), two launches fewer per block.
// Arithmetic per output is that of the separate kernels (same tap order, same fp32 fma chains, same bf16 roundings): dx is
// bit-identical, dk differs only through the grouping of its partial sums.
#include "common.h"

namespace {

constexpr int kNT = 256, kPXB = 3, kVN = 4, kCL = 16, kPT = kNT / kCL;   // 16 lanes x 4 bf16 channels = one 128-byte slab per pixel
#define BN_EPS_F 1e-3f
#ifndef FUSED_WPE
#define FUSED_WPE 2
#endif
#ifndef FUSED_ABL
#define FUSED_ABL 0
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: a uint4 struct copy becomes a memcpy through scratch
__device__ __forceinline__ void widen(const uint2& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ void widen8(const u32x4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

constexpr int kFL = 8, kFPT = kNT / kFL, kNCH = 4;   // fill: 8 lanes x 16 bytes per pixel; at most kNCH chunks per thread and band

struct FillGeom { int gfirst, n16, slot0, Wt, R, H, W, C, w0; };

// issue the global loads of up to kNCH chunks per thread of rows gfirst.. (d, da, x); `inside` bit uu: chunk uu lies in the image
__device__ __forceinline__ void fill_load(const FillGeom& g, int base, const bf16_t* db, const bf16_t* gb, const bf16_t* xb,
                                          u32x4 (&vd)[kNCH], u32x4 (&vg)[kNCH], u32x4 (&vx)[kNCH], unsigned& inside) {
  const int dfy = kFPT / g.Wt, dfx = kFPT - dfy * g.Wt;
  const int pix0 = (base + (int)threadIdx.x) / kFL;
  int fy = pix0 / g.Wt, fx = pix0 - fy * g.Wt;
  inside = 0;
#pragma unroll
  for (int uu = 0; uu < kNCH; ++uu) {
    const int i = base + (int)threadIdx.x + uu * kNT;
    const int gh = g.gfirst + fy, gw = g.w0 + fx - 1;
    // every lane loads (from a clamped, valid address): a conditional load would make the number of outstanding loads
    // unknown to the wait-count insertion, which then drains all of them at the first use of any
    const int o = (min(max(gh, 0), g.H - 1) * g.W + min(max(gw, 0), g.W - 1)) * g.C;
    if (FUSED_ABL != 1) {
      vd[uu] = *reinterpret_cast<const u32x4*>(db + o);
      vg[uu] = *reinterpret_cast<const u32x4*>(gb + o);
      vx[uu] = *reinterpret_cast<const u32x4*>(xb + o);
    } else {
      vd[uu] = vg[uu] = vx[uu] = u32x4{0u, 0u, 0u, 0u};
    }
    if (i < g.n16 && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W) inside |= 1u << uu;
    fx += dfx; fy += dfy;
    if (fx >= g.Wt) { fx -= g.Wt; ++fy; }
  }
}

// BatchNorm-backward pass 2 on the loaded chunks (dd, rounded to bf16 as the stored tensor would be; 0 outside the image) and
// the stores into the two ring tiles
__device__ __forceinline__ void fill_store(const FillGeom& g, int base, const float* cst,
                                           const u32x4 (&vd)[kNCH], const u32x4 (&vg)[kNCH], const u32x4 (&vx)[kNCH], unsigned inside,
                                           float* tileD, float* tileX) {
  float sc[8], sh[8], Pc[8], Qc[8];
  {
    const float* cf = cst + 9 * 64 + 8 * (threadIdx.x & (kFL - 1));
    float* const dst[4] = {sc, sh, Pc, Qc};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const float4 v = *reinterpret_cast<const float4*>(cf + a * 64 + 4 * hq);
        dst[a][4 * hq] = v.x; dst[a][4 * hq + 1] = v.y; dst[a][4 * hq + 2] = v.z; dst[a][4 * hq + 3] = v.w;
      }
  }
  const int f8 = threadIdx.x & (kFL - 1);
  const int dfy = kFPT / g.Wt, dfx = kFPT - dfy * g.Wt;
  const int pix0 = (base + (int)threadIdx.x) / kFL;
  int fy = pix0 / g.Wt, fx = pix0 - fy * g.Wt;
#pragma unroll
  for (int uu = 0; uu < kNCH; ++uu) {
    const int i = base + (int)threadIdx.x + uu * kNT;
    if (i < g.n16) {
      int slot = g.slot0 + fy;
      slot -= slot >= g.R ? g.R : 0;
      slot -= slot >= g.R ? g.R : 0;
      const int li = ((slot * g.Wt + fx) * kFL + f8) * 8;   // float index: 64 channels x 4 B per pixel and tile
      float r[8], xw[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { r[e] = 0.f; xw[e] = 0.f; }
      if (inside & (1u << uu)) {                             // (outside the image both tiles hold zeros: the load address was clamped)
        float xv[8], gv[8];
        widen8(vd[uu], xv); widen8(vg[uu], gv); widen8(vx[uu], xw);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float tv = fmaf(xv[e], sc[e], sh[e]);                   // ReLU6 passes the gradient where 0 < BN(d) < 6 (the clamp itself is not needed)
          const float gy = (tv > 0.f && tv < 6.f) ? gv[e] : 0.f;
          r[e] = bn_bwd_dx_pq(xv[e], gy, sc[e], Pc[e], Qc[e]);
        }
        // dd as the stored bf16 tensor would hold it (round to nearest even), kept as fp32 so that the gradient loops need no widening
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const unsigned w = pack2_bf16(r[e], r[e + 1]);
          r[e] = __uint_as_float(w << 16); r[e + 1] = __uint_as_float(w & 0xffff0000u);
        }
      }
      *reinterpret_cast<float4*>(tileD + li) = make_float4(r[0], r[1], r[2], r[3]);
      *reinterpret_cast<float4*>(tileD + li + 4) = make_float4(r[4], r[5], r[6], r[7]);
      *reinterpret_cast<float4*>(tileX + li) = make_float4(xw[0], xw[1], xw[2], xw[3]);
      *reinterpret_cast<float4*>(tileX + li + 4) = make_float4(xw[4], xw[5], xw[6], xw[7]);
    }
    fx += dfx; fy += dfy;
    if (fx >= g.Wt) { fx -= g.Wt; ++fy; }
  }
}

// Work decomposition: workgroup (channel slab of 64, image, row group, column tile).  A workgroup walks its row group top to
// bottom in bands of TH output rows over a RING of R = TH + 2 tile rows in LDS (image row g lives in slot (g + 1) mod R): the
// two bottom rows of a band are the top halo of the next one, so after the first band only TH new rows are filled -- the halo
// re-read/re-transform that a fresh tile per band costs (1.4-1.8x on these maps) is paid once per row group.  The kernel is
// bound by bytes in flight, not by either pipe (loads-removed and compute-removed builds each take half the time): the global
// loads of the NEXT band's rows are issued into registers before the two gradient loops of the current band and only
// transformed and stored to the ring after them, so HBM works during the arithmetic.  The data gradient and the weight
// gradient run as two loops over the band so that the flipped taps (36 registers, reloaded per band) and the BatchNorm
// coefficients (48) are never live together; the weight-gradient accumulators (36) persist.
__global__ __launch_bounds__(kNT, FUSED_WPE) void dw_bwd_fused_kernel(const bf16_t* __restrict__ d, const bf16_t* __restrict__ da,
                                                                      const float* __restrict__ bnstate, const float* __restrict__ coef,
                                                                      const bf16_t* __restrict__ xin, const float* __restrict__ k,
                                                                      bf16_t* __restrict__ dx, float* __restrict__ partials, int B, int H, int W, int C,
                                                                      int TH, int Hg, int G, int Wc, int nCol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wt = Wc + 2, R = TH + 2;
  const int tile_bytes = R * Wt * 256;                 // fp32 tiles: every value is widened once, at fill time, not once per use
  float* tileD = reinterpret_cast<float*>(smem);
  float* tileX = reinterpret_cast<float*>(smem + tile_bytes);
  const int tid = threadIdx.x, c4 = tid & (kCL - 1), pt = tid / kCL;
  const int cc0 = blockIdx.x * 64;
  int y = blockIdx.y;
  const int col = y % nCol; y /= nCol;
  const int grp = y % G, b = y / G;
  const long img = (long)b * H * W * C;
  const int w0 = col * Wc, wc = min(Wc, W - w0);
  const int hstart = grp * Hg, hend = min(H, hstart + Hg);

  // per-slab constants in LDS (read per band: no global loads whose wait would also drain the prefetch; nothing hoisted
  // into long-lived registers): the mirrored taps [9][64], then scale | shift | P | Q [4][64] (common.h: bn_bwd_pq)
  float* cst = reinterpret_cast<float*>(smem + 2 * tile_bytes);
  for (int i = tid; i < 13 * 64; i += kNT) {
    const int a = i >> 6, ch = cc0 + (i & 63);
    float v;
    if (a < 9) v = k[(8 - a) * C + ch];
    else if (a == 9) v = bnstate[2 * C + ch];
    else if (a == 10) v = bnstate[3 * C + ch];
    else {
      float P, Q;
      bn_bwd_pq(bnstate[2 * C + ch], coef[ch], coef[C + ch], bnstate[ch], 1.0f / sqrtf(bnstate[C + ch] + BN_EPS_F), P, Q);
      v = a == 11 ? P : Q;
    }
    cst[i] = v;
  }
  float dk[9][kVN];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < kVN; ++e) dk[t][e] = 0.f;

  const int gpr = (Wc + kPXB - 1) / kPXB;
  const int dly = kPT / gpr, dlg = kPT - dly * gpr;
  const int ch0 = cc0 + 8 * (tid & (kFL - 1));
  const bf16_t* db = d + img + ch0; const bf16_t* gb = da + img + ch0; const bf16_t* xb = xin + img + ch0;
  FillGeom fg; fg.Wt = Wt; fg.R = R; fg.H = H; fg.W = W; fg.C = C; fg.w0 = w0;
  // first band: all R rows (hstart - 1 .. hstart + TH), loaded and stored in place
  fg.gfirst = hstart - 1; fg.n16 = R * Wt * kFL; fg.slot0 = hstart % R;
  __syncthreads();
  for (int base = 0; base < fg.n16; base += kNCH * kNT) {
    u32x4 vd[kNCH], vg[kNCH], vx[kNCH]; unsigned inside;
    fill_load(fg, base, db, gb, xb, vd, vg, vx, inside);
    fill_store(fg, base, cst, vd, vg, vx, inside, tileD, tileX);
  }
  __syncthreads();
  fg.n16 = TH * Wt * kFL;                              // <= kNCH * kNT by the host's choice of TH
  for (int h0 = hstart; h0 < hend; h0 += TH) {
    const bool has_next = h0 + TH < hend;
    u32x4 vd[kNCH], vg[kNCH], vx[kNCH]; unsigned inside = 0;
    fg.gfirst = h0 + TH + 1; fg.slot0 = (h0 + TH + 2) % R;
    if (has_next) fill_load(fg, 0, db, gb, xb, vd, vg, vx, inside);
    const int rows = min(TH, hend - h0);
    const int npg = (FUSED_ABL == 2) ? 0 : rows * gpr;
    const int hm = h0 % R;                             // slot of image row h0 - 1 (window row 0 of output row h0)
    // ---------------- data gradient: correlation of dd with the mirrored taps
    {
      float kw[9][kVN];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 w = *reinterpret_cast<const float4*>(cst + t * 64 + kVN * c4);
        kw[t][0] = w.x; kw[t][1] = w.y; kw[t][2] = w.z; kw[t][3] = w.w;
      }
      int ly = pt / gpr, lg = pt - ly * gpr;
      bf16_t* ob = dx + img + cc0 + kVN * c4;
      for (int pg = pt; pg < npg; pg += kPT) {
        const int lx = lg * kPXB;
        float a[kPXB][kVN];
#pragma unroll
        for (int e = 0; e < kPXB; ++e)
#pragma unroll
          for (int c = 0; c < kVN; ++c) a[e][c] = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          int slot = hm + ly + i; slot -= slot >= R ? R : 0;
          float r[kPXB + 2][kVN];
#pragma unroll
          for (int j = 0; j < kPXB + 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(tileD + ((slot * Wt + lx + j) * kCL + c4) * 4);
            r[j][0] = v.x; r[j][1] = v.y; r[j][2] = v.z; r[j][3] = v.w;
          }
#pragma unroll
          for (int e = 0; e < kPXB; ++e)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int c = 0; c < kVN; ++c) a[e][c] = fmaf(r[e + j][c], kw[i * 3 + j][c], a[e][c]);
        }
        const int o = ((h0 + ly) * W + w0 + lx) * C;
#pragma unroll
        for (int e = 0; e < kPXB; ++e)
          if (lx + e < wc && (FUSED_ABL != 4 || a[e][0] == 12345.678f)) st4(ob + o + e * C, make_float4(a[e][0], a[e][1], a[e][2], a[e][3]));
        lg += dlg; ly += dly;
        if (lg >= gpr) { lg -= gpr; ++ly; }
      }
    }
    // ---------------- weight gradient: dk[i][j] += x[p + (i-1, j-1)] * dd[p] over the band's pixels
    if (FUSED_ABL != 3) {
      int ly = pt / gpr, lg = pt - ly * gpr;
      for (int pg = pt; pg < npg; pg += kPT) {
        const int lx = lg * kPXB;
        float ddc[kPXB][kVN];                          // dd at the three output pixels (centres of their windows); 0 past the tile
        {
          int slot = hm + ly + 1; slot -= slot >= R ? R : 0;
#pragma unroll
          for (int e = 0; e < kPXB; ++e) {
            const float4 v = *reinterpret_cast<const float4*>(tileD + ((slot * Wt + lx + 1 + e) * kCL + c4) * 4);
            ddc[e][0] = v.x; ddc[e][1] = v.y; ddc[e][2] = v.z; ddc[e][3] = v.w;
            if (lx + e >= wc) { ddc[e][0] = ddc[e][1] = ddc[e][2] = ddc[e][3] = 0.f; }
          }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          int slot = hm + ly + i; slot -= slot >= R ? R : 0;
          float r[kPXB + 2][kVN];
#pragma unroll
          for (int j = 0; j < kPXB + 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(tileX + ((slot * Wt + lx + j) * kCL + c4) * 4);
            r[j][0] = v.x; r[j][1] = v.y; r[j][2] = v.z; r[j][3] = v.w;
          }
#pragma unroll
          for (int e = 0; e < kPXB; ++e)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int c = 0; c < kVN; ++c) dk[i * 3 + j][c] = fmaf(r[e + j][c], ddc[e][c], dk[i * 3 + j][c]);
        }
        lg += dlg; ly += dly;
        if (lg >= gpr) { lg -= gpr; ++ly; }
      }
    }
    if (has_next) {
      __syncthreads();                                 // the TH oldest rows are no longer read
      fill_store(fg, 0, cst, vd, vg, vx, inside, tileD, tileX);
      __syncthreads();
    }
  }
  // ---------------- weight-gradient partials of this workgroup: pixel-threads of a wave by shuffles, the 4 waves through LDS
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);      // [4 waves][9][kCL][kVN]
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < kVN; ++e) {
      float v = dk[t][e];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      dk[t][e] = v;
    }
  if (lane < kCL) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < kVN; ++e) red[((wave * 9 + t) * kCL + c4) * kVN + e] = dk[t][e];
  }
  __syncthreads();
  for (int i = tid; i < 9 * kCL * kVN; i += kNT) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += red[w * 9 * kCL * kVN + i];
    const int v = i / (kCL * kVN), cch = i % (kCL * kVN);
    partials[((long)blockIdx.y * 9 + v) * C + cc0 + cch] = acc;
  }
}

// Geometry: column tiles of <= 24 pixels (a multiple of the 3-pixel groups), band height from the LDS budget of four
// workgroups per CU and from how well TH * groups-per-row fills the 16 pixel-threads, row groups so that the launch has
// about two rounds of resident workgroups.
struct FusedGeom { int TH, Hg, G, Wc, nCol, rows; size_t lds; };
FusedGeom fused_geom(int B, int H, int W, int C) {
  FusedGeom g;
  g.nCol = cdiv(W, 24);
  g.Wc = cdiv(cdiv(W, g.nCol), kPXB) * kPXB;
  g.nCol = cdiv(W, g.Wc);
  const int gpr = g.Wc / kPXB;
  const long row_bytes = 2L * (g.Wc + 2) * 256;      // both tiles, fp32
  const long budget = (long)crnn_knob("CRNN_FUSED_LDS", 74 * 1024);   // two workgroups per CU (the registers allow no more)
  const long base = (long)B * (C / 64) * g.nCol;
  const int gt = (int)crnn_knob("CRNN_FUSED_WGS", 2048);
  int G = (int)((gt + base - 1) / base); if (G < 1) G = 1;
  int best_th = 1; double best = 1e30;
  for (int th = 1; th <= H; ++th) {
    if ((th + 2) * row_bytes > budget || th * (g.Wc + 2) * kFL > kNCH * kNT) break;
    int gg = G; if (gg > cdiv(H, 2 * th)) gg = cdiv(H, 2 * th);   // a row group is at least two bands
    if (gg < 1) gg = 1;
    const int hg = cdiv(H, gg);
    // cost of one row group: pixel-group iterations (16 pixel-threads) + fill rows + a barrier pair per band
    const int nb = cdiv(hg, th), last = hg - (nb - 1) * th;
    const double it = (nb - 1) * (double)cdiv(th * gpr, kPT) + cdiv(last * gpr, kPT);
    const double fill = (hg + 2.0) * (g.Wc + 2) / 32.0 * 0.6;    // chunks per thread, weighted against a pixel-group iteration
    const double c = (it + fill + 0.3 * nb) * cdiv(H, hg);
    if (c < best) { best = c; best_th = th; }
  }
  g.TH = (int)crnn_knob("CRNN_FUSED_TH", best_th);
  if ((g.TH + 2) * row_bytes > 78 * 1024 || g.TH * (g.Wc + 2) * kFL > kNCH * kNT || g.TH < 1) g.TH = best_th;
  if (G > cdiv(H, 2 * g.TH)) G = cdiv(H, 2 * g.TH);
  if (G < 1) G = 1;
  G = (int)crnn_knob("CRNN_FUSED_G", G);
  g.Hg = cdiv(H, G);
  g.G = cdiv(H, g.Hg);
  g.rows = B * g.G * g.nCol;
  g.lds = (size_t)((g.TH + 2) * row_bytes) + 13 * 64 * sizeof(float);
  const size_t red = 4 * 9 * kCL * kVN * sizeof(float);
  if (g.lds < red) g.lds = red;
  return g;
}

}  // namespace

// 0 if the fused depthwise backward handles (H, W, C) in bf16 storage, else -3
extern "C" int crnn_dwconv_bwd_fused_supported(int H, int W, int C) {
  return (W >= 1 && H >= 1 && C >= 64 && C % 64 == 0) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
// rows of [9][C] weight-gradient partials the fused kernel writes (scratch = rows * 9 * C floats)
extern "C" int crnn_dwconv_bwd_fused_rows(int B, int H, int W, int C) {
  if (crnn_dwconv_bwd_fused_supported(H, W, C) != CRNN_OK) return 0;
  return fused_geom(B, H, W, C).rows;
}

// d, da, xin, dx: bf16 [B,H,W,C]; bnstate = [mean|var|scale|shift] of the BatchNorm after the depthwise conv; coef = [c1|c2] from
// crnn_bn_bwd_ex(..., dx = null); k = depthwise kernel [9][C]; dk [9][C] out; scratch: crnn_dwconv_bwd_fused_rows() * 9 * C floats.
extern "C" int crnn_dwconv3x3_bwd_fused(const void* d, const void* da, const float* bnstate, const float* coef, const void* xin, const float* k,
                                        void* dx, float* dk, float* scratch, int B, int H, int W, int C, hipStream_t stream) {
  CRNN_TRY(crnn_dwconv_bwd_fused_supported(H, W, C));
  if ((((uintptr_t)d | (uintptr_t)da | (uintptr_t)xin | (uintptr_t)dx | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)k) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)H * W * C >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;
  const FusedGeom g = fused_geom(B, H, W, C);
  if (g.lds > 80 * 1024 || (long)g.rows > 65535) return CRNN_ERR_UNSUPPORTED;
  if (g.lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)dw_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(dw_bwd_fused_kernel, dim3(C / 64, g.rows), dim3(kNT), g.lds, stream, (const bf16_t*)d, (const bf16_t*)da, bnstate, coef,
                     (const bf16_t*)xin, k, (bf16_t*)dx, scratch, B, H, W, C, g.TH, g.Hg, g.G, g.Wc, g.nCol);
  CRNN_LAUNCH_CHECK();
  return crnn_partials_sum(scratch, g.rows, 9 * C, dk, 1.f, stream);
}
