// dense2's backward in one pass (round 5).  Reference: the TimeDistributed softmax Dense at the top of the recogniser (utils.py:85-86) under
// K.ctc_batch_cost (utils.py:98-103); its input is the Dropout(.2) output of the second recurrent layer (utils.py:82-84).
//
// The layer is skinny: x [M][K] (M = T * B rows, K = 2 * units = 512), dy [M][C] (C = num_classes = 38).  Its three backward products
//   dW[k][c] = sum_m x[m][k] dy[m][c]      db[c] = sum_m dy[m][c]      dx[m][k] = (sum_c dy[m][c] W[k][c]) * dropout_multiplier(m, k)
// are 2 * 2 * M * K * C flops = 1 GFLOP over 57 MB of tensors at the headline shape, and a fit for the vector ALU with the WEIGHTS AND THE
// WEIGHT-GRADIENT ACCUMULATORS IN REGISTERS: a thread owns two input features k (4 x 40 registers), walks its workgroup's rows, takes each row's
// dy as LDS broadcasts, does 4 C multiply-adds and writes two dx elements with the forward's dropout multiplier applied (the decisions come as
// keep bytes from the forward's dropout pass, crnn_dropout_keep).  Exact fp32 arithmetic in every precision mode.
// Before: two tile GEMMs with a padded 38-wide dimension, a split-K reduce, a column reduce with its second stage and a dropout pass (six
// launches, 77 us, x read twice and dx written twice).  Now 30 + 7 us (kernel + second-stage sum; profiles/r05_dense_bwd_bench.txt):
//   * the row loop is vector-ALU work on ONE wave per SIMD (4 x 40 registers of weights and accumulators per thread leave room for no second one): 80
//     packed multiply-adds (545 M in all: 7 us at the packed fp32 peak, 14 at the scalar rate) + ~35 other instructions per row and wave take ~600
//     clocks = 13 us over a workgroup's 52 rows with the LDS reads and the stores compiled out; dy's LDS broadcasts add 5 us that the half-row
//     pipelining does not hide, the dx stores 1.5: 20 us;
//   * 5-7 us until the first rows and the weights have landed (256 workgroups ask for 13 MB at once), 3 us for the 20 MB of partial gradients;
//   * what it was before each fix (the timing build's per-workgroup stamps, -DCRNN_DSB_TRACE): rows loaded into registers one step ahead -- 16 KB
//     per CU in flight, 1.6 TB/s; W read as 38 strided words per thread -- 2.5 MB through the vector cache per workgroup; W staged by predicated
//     loads -- issued one by one, 12 us; the loader wave hashing the dropout decisions -- 1.6 us per 8-row step on one wave, the critical path;
//     a bounds check between unrolled rows -- every row waited out its LDS latency, 0.44 us per row; the unrolled step left to the scheduler --
//     300-470 spilled registers.
//
// Determinism: a workgroup owns a contiguous block of rows and sums them in ascending order; the per-workgroup partial gradients are summed in
// workgroup order (fixed grid for a given M).
#include "common.h"

#ifndef CRNN_DSB_EXP
#define CRNN_DSB_EXP 0   // ablation builds for scripts/dense_bench.py: 1 no multiply-adds | 2 no LDS reads of dy | 4 no partial-gradient store | 8 no dx store | 16 no x DMA | 32 no keep hashing | 64 no W staging
#endif

namespace {

constexpr int kDsbMaxThreads = 512;

struct DsbParams {
  const float* x; const float* dy; const float* W; float* dx; float* part; const unsigned char* keep;
  long M; int K, C, ldx, lddx, rows_per_wg; long part_stride;
  float rate; uint64_t seed; uint32_t layer;
#ifdef CRNN_DSB_TRACE
  unsigned long long* trace = nullptr;   // timing build only: [workgroup][8] s_memrealtime stamps (100 MHz)
#endif
};
#ifdef CRNN_DSB_TRACE
#define DSB_STAMP(i) do { if (p.trace && tid == 0) p.trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DSB_STAMP(i) do {} while (0)
#endif

constexpr int kDsbDepth = 3, kDsbSlots = kDsbDepth + 1;   // steps in flight, ring slots
constexpr int kDsbRB = 8;                                  // rows per step
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dsb_glds16(const void* g, void* l) {   // 16 bytes per lane, nontemporal: the layer's input is not read again
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 2);
}
__device__ __forceinline__ void dsb_glds4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32: two multiply-adds per lane and issue

// LDS map: W [K][C] (the partial gradients at the end) | x ring [slots][RB][K] | dy ring [slots][RB][64] | keep ring [slots][RB][K / 8 bytes, 512 per slot] | 512 spare
__host__ __device__ inline int dsb_w_bytes(int K, int C) { return (K * C * 4 + 15) & ~15; }
__host__ __device__ inline int dsb_lds_bytes(int K, int C) { return dsb_w_bytes(K, C) + kDsbSlots * kDsbRB * (K * 4 + 256 + 64) + 512; }

// K / 128 compute waves -- thread j owns input features j and j + K/2: the dy values a row broadcasts from LDS feed four multiply-adds per class
// (one feature per thread: the LDS return path, 8 cycles per 16-byte broadcast, bounds the kernel) and pair up for packed fp32 math -- plus one
// loader wave.  A step is 8 rows: the loader brings the step's x rows (contiguous: 8 * K * 4 bytes, 1 KiB per LDS-DMA instruction) and dy rows (one
// 4-byte-per-lane instruction per row, 64 words apart in LDS: columns >= C hold a neighbour's values and meet zero weights) into ring slot s % 4 three
// steps ahead of their use and evaluates the step's dropout keep bytes.  One barrier per step; the loader waits with a counted vmcnt (it issues
// loads only: they complete in order).
template <int CP>
__global__ __launch_bounds__(kDsbMaxThreads / 2 + 64) void dense_bwd_small_kernel(DsbParams p) {
  static_assert(CP % 4 == 0 && CP <= 64, "classes are read four at a time from 64-word rows");
  constexpr int RB = kDsbRB;
  constexpr int XI = 2 * RB;                  // 1 KiB x chunks per step at K = 512 (fewer real ones below: the last is repeated)
  constexpr int NIT = XI + RB + 2;            // loader instructions per step (x, dy rows, keep bytes)
  static_assert((kDsbDepth - 1) * NIT <= 63, "vmcnt is six bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = p.K, C = p.C, KH = K >> 1, nw = KH >> 6;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, k = tid;
  const bool loader = wave == nw;
  const int nthr = KH + 64;
  DSB_STAMP(0);
  const int XSLOT = RB * K * 4;                               // bytes of x per ring slot
  const int X0 = dsb_w_bytes(K, C), DY0 = X0 + kDsbSlots * XSLOT, KP0 = DY0 + kDsbSlots * RB * 256;
  float* wl = reinterpret_cast<float*>(smem);
  const long r0 = (long)blockIdx.x * p.rows_per_wg, r1 = min(p.M, r0 + p.rows_per_wg);
  const int steps = (int)((r1 - r0 + RB - 1) / RB);
  const bool drop = p.rate > 0.f;
  const int gpr = K >> 3;   // dropout groups (8 elements) per row, <= 64
  float db = 0.f;
  f32x2 a0[CP / 2], a1[CP / 2];
#pragma unroll
  for (int c = 0; c < CP / 2; ++c) { a0[c] = f32x2{0.f, 0.f}; a1[c] = f32x2{0.f, 0.f}; }
  if (loader) {
    const unsigned char* gx = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* gdy = reinterpret_cast<const unsigned char*>(p.dy);
    const long xbytes = p.M * (long)K * 4, dybytes = p.M * (long)C * 4;
    const unsigned char* gk = p.keep ? p.keep : gdy;
    const long gkbytes = p.keep ? p.M * (long)gpr : dybytes;
    const int nxi = XSLOT >> 10;
    const uint32_t thr = crnn_drop_threshold(p.rate);
    const crnn_rng_key key = crnn_rng_make_key(p.seed, p.layer);
    auto issue = [&](int t, int slot) {
      t = t < steps ? t : steps - 1;                           // past the end: the last step again, into a slot nobody reads any more
      const long rb = r0 + (long)t * RB;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int ii = i < nxi ? i : nxi - 1;
        long off = rb * K * 4 + ii * 1024 + lane * 16;
        off = off < xbytes - 16 ? off : xbytes - 16;           // rows past the tensor (last workgroup's partial step): in-bounds bytes nobody uses
        if (!(CRNN_DSB_EXP & 16)) dsb_glds16(gx + off, smem + X0 + slot * XSLOT + ii * 1024);
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        long off = ((rb + r) * C + lane) * 4;
        off = off < dybytes - 4 ? off : dybytes - 4;
        dsb_glds4(gdy + off, smem + DY0 + (slot * RB + r) * 256);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {                            // the step's keep bytes: RB * K / 8 <= 512 contiguous bytes of the table the forward wrote
        long off = rb * gpr + j * 256 + lane * 4;
        off = off < gkbytes - 4 ? off : gkbytes - 4;
        dsb_glds4(gk + off, smem + (p.keep ? KP0 + slot * 512 : KP0 + kDsbSlots * 512) + j * 256);   // (no table: any bytes, into the spare 512)
      }
      if (drop && !p.keep && !(CRNN_DSB_EXP & 32)) {           // no table: the loader evaluates the hashes itself (1.6 us per step: it then bounds the kernel)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          if (lane < gpr) smem[KP0 + slot * 512 + r * gpr + lane] = (unsigned char)crnn_keep8(key, (uint64_t)((rb + r) * gpr + lane), thr);
      }
    };
#pragma unroll
    for (int t = 0; t < kDsbDepth; ++t) issue(t, t);
    __builtin_amdgcn_s_barrier();                              // (the compute waves have staged W)
    int slot = kDsbDepth;
    for (int t = 0; t < steps; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kDsbDepth - 1) * NIT) : "memory");   // step t has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               // ... and the keep bytes written so far
      __builtin_amdgcn_s_barrier();
      issue(t + kDsbDepth, slot);                              // the slot step t - 1 has just released
      slot = slot + 1 == kDsbSlots ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // W reaches the registers through LDS: a thread's row is C words at a stride of C words -- read straight from memory, every load of a wave touches
    // 64 cache lines for 4 bytes each (2.5 MB through the vector cache per workgroup); staged with 16-byte loads it is one coalesced pass.
    {
      const float4* gw = reinterpret_cast<const float4*>(p.W);
      const int n4 = (K * C) >> 2;
      float4 v[CP / 2];
#pragma unroll
      for (int j = 0; j < CP / 2; ++j) { const int i = k + j * KH; v[j] = gw[i < n4 ? i : n4 - 1]; }   // unconditional: all in flight at once (predicated, they went one by one: 12 us)
#pragma unroll
      for (int j = 0; j < CP / 2; ++j) { const int i = k + j * KH; if (i < n4 && !(CRNN_DSB_EXP & 64)) reinterpret_cast<float4*>(wl)[i] = v[j]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DSB_STAMP(1);
    f32x2 w0[CP / 2], w1[CP / 2];
#pragma unroll
    for (int c = 0; c < CP / 2; ++c) {
      w0[c] = f32x2{2 * c < C ? wl[k * C + 2 * c] : 0.f, 2 * c + 1 < C ? wl[k * C + 2 * c + 1] : 0.f};
      w1[c] = f32x2{2 * c < C ? wl[(k + KH) * C + 2 * c] : 0.f, 2 * c + 1 < C ? wl[(k + KH) * C + 2 * c + 1] : 0.f};
    }
    const float inv_keep = drop ? 1.f / (1.f - p.rate) : 1.f;
    const unsigned nodrop = drop ? 0u : 1u;
    int slot = 0;
    for (int t = 0; t < steps; ++t) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t == 0) DSB_STAMP(2);
      const long rb = r0 + (long)t * RB;
      const float* xs = reinterpret_cast<const float*>(smem + X0 + slot * XSLOT);
      const float* dls = reinterpret_cast<const float*>(smem + DY0 + slot * RB * 256);
      const unsigned char* kp = smem + KP0 + slot * 512;
      // A row is one iteration of a ROLLED loop (one basic block), software-pipelined by hand over half rows (20 classes = five 16-byte LDS
      // broadcasts): the reads of the next half are issued before the 40 packed multiply-adds of the current one and land under them -- 2 x 20
      // registers of dy next to the 160 of weights and accumulators.  (Unrolled, the step's reads were hoisted / its second halves sunk wholesale and
      // 300-470 registers spilled; with a bounds check between unrolled rows each row waited out its LDS latency: 0.44 us per row.)
      const int nr = rb + RB <= r1 ? RB : (int)(r1 - rb);
      float4 d[2][CP / 8];
      auto ld = [&](int h, int r, int half) {
#pragma unroll
        for (int j = 0; j < CP / 8; ++j)
          d[h][j] = (CRNN_DSB_EXP & 2) ? make_float4(1.f, 2.f, 3.f, (float)r) : *reinterpret_cast<const float4*>(dls + r * 64 + half * (CP / 2) + 4 * j);
      };
      // (the row sums run as four independent chains -- two per feature -- interleaved with the accumulator updates: with one wave per SIMD a chain's
      //  next link issued right behind the previous one waits out the multiply-add pipeline)
      auto mac = [&](int h, int half, f32x2 x0, f32x2 x1, f32x2& s0, f32x2& s1, f32x2& t0, f32x2& t1) {
#pragma unroll
        for (int j = 0; j < CP / 8; ++j) {
          const int c2 = half * (CP / 4) + 2 * j;          // index of the class pair
          const f32x2 dlo = f32x2{d[h][j].x, d[h][j].y}, dhi = f32x2{d[h][j].z, d[h][j].w};
          if (CRNN_DSB_EXP & 1) { s0 += dlo; a0[c2] += x0; continue; }
          s0 = pk_fma(dlo, w0[c2], s0); a0[c2] = pk_fma(x0, dlo, a0[c2]);
          s1 = pk_fma(dlo, w1[c2], s1); a1[c2] = pk_fma(x1, dlo, a1[c2]);
          t0 = pk_fma(dhi, w0[c2 + 1], t0); a0[c2 + 1] = pk_fma(x0, dhi, a0[c2 + 1]);
          t1 = pk_fma(dhi, w1[c2 + 1], t1); a1[c2 + 1] = pk_fma(x1, dhi, a1[c2 + 1]);
        }
      };
      float xn0 = xs[k], xn1 = xs[k + KH];
      unsigned kn0 = kp[k >> 3], kn1 = kp[(k + KH) >> 3];   // (KH is a multiple of 64: the same bit of another byte; rows are K / 8 bytes apart)
      float dbn = dls[lane];                                // (read by every thread: no branch in the row; only threads k < C keep the sum)
      ld(0, 0, 0);
      float* dxr = p.dx + rb * p.lddx + k;
#pragma unroll 1
      for (int r = 0; r < nr; ++r) {
        const float xv0 = xn0, xv1 = xn1;
        const unsigned kb0 = kn0, kb1 = kn1;
        db += dbn;
        const f32x2 x0 = f32x2{xv0, xv0}, x1 = f32x2{xv1, xv1};
        f32x2 s0 = f32x2{0.f, 0.f}, s1 = f32x2{0.f, 0.f}, t0 = f32x2{0.f, 0.f}, t1 = f32x2{0.f, 0.f};
        ld(1, r, 1);
        __builtin_amdgcn_sched_barrier(0);
        mac(0, 0, x0, x1, s0, s1, t0, t1);
        __builtin_amdgcn_sched_barrier(0);
        const int rn = r + 1 < RB ? r + 1 : RB - 1;         // the row after the step's last: the last again (unused)
        ld(0, rn, 0);
        xn0 = xs[rn * K + k]; xn1 = xs[rn * K + k + KH];
        kn0 = kp[rn * gpr + (k >> 3)]; kn1 = kp[rn * gpr + ((k + KH) >> 3)];
        dbn = dls[rn * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        mac(1, 1, x0, x1, s0, s1, t0, t1);
        s0 += t0; s1 += t1;
        const float m0 = (((kb0 >> (k & 7)) & 1) | nodrop) ? inv_keep : 0.f, m1 = (((kb1 >> (k & 7)) & 1) | nodrop) ? inv_keep : 0.f;
        if (!(CRNN_DSB_EXP & 8) || s0.x == 1234.5f) {
          __builtin_nontemporal_store((s0.x + s0.y) * m0, dxr);
          __builtin_nontemporal_store((s1.x + s1.y) * m1, dxr + KH);
        }
        dxr += p.lddx;
        __builtin_amdgcn_sched_barrier(0);
      }
      slot = slot + 1 == kDsbSlots ? 0 : slot + 1;
    }
    DSB_STAMP(3);
    // every thread read its weights before its first step barrier: their place takes the partial gradients
#pragma unroll
    for (int c = 0; c < CP / 2; ++c) {
      if (2 * c < C) { wl[k * C + 2 * c] = a0[c].x; wl[(k + KH) * C + 2 * c] = a1[c].x; }
      if (2 * c + 1 < C) { wl[k * C + 2 * c + 1] = a0[c].y; wl[(k + KH) * C + 2 * c + 1] = a1[c].y; }
    }
  }
  __syncthreads();
  DSB_STAMP(4);
  float* part = p.part + (long)blockIdx.x * p.part_stride;
  if (!(CRNN_DSB_EXP & 4)) for (int i = tid; i < (K * C) >> 2; i += nthr) reinterpret_cast<float4*>(part)[i] = reinterpret_cast<const float4*>(wl)[i];
  if (!loader && k < C) part[(long)K * C + k] = db;
#ifdef CRNN_DSB_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DSB_STAMP(5);
#endif
}

// second stage: out[i] = sum over the workgroups' partial rows, in row order (8 interleaved chains, combined in order), double accumulation as
// crnn_partials_sum; 16-byte loads over 512-byte row segments (the generic kernel's 64-byte segments ran at 1.3 TB/s over these 20 MB)
__global__ __launch_bounds__(256) void dense_bwd_small_sum_kernel(const float* __restrict__ part, int G, long stride, int n, float* __restrict__ out) {
  __shared__ double red[8][32][4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long col = (blockIdx.x * 32L + tx) * 4;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  if (col < stride)
    for (int g0 = ty; g0 < G; g0 += 64) {          // eight loads in flight per thread (a rolled loop waits out one memory latency per partial row)
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int g = g0 + 8 * j;
        v[j] = g < G ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + g * stride + col)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[0] += v[j].x; a[1] += v[j].y; a[2] += v[j].z; a[3] += v[j].w; }
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ty][tx][e] = a[e];
  __syncthreads();
  if (ty == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double s = 0.0;
      for (int r = 0; r < 8; ++r) s += red[r][tx][e];
      if (col + e < n) out[col + e] = (float)s;
    }
  }
}

struct DsbGeom { int G, rows_per_wg; long part_stride; };
constexpr int kDsbClasses = 40;   // padded class count: 4 x 40 registers of weights and accumulators per thread (two input features)
DsbGeom dsb_geom(long M, int K, int C) {
  DsbGeom g;
  g.G = (int)std::min<long>(256, (M + kDsbRB - 1) / kDsbRB);
  const long rpw = (M + g.G - 1) / g.G;     // the last step of a workgroup may be partial
  g.rows_per_wg = (int)rpw;
  g.G = (int)((M + rpw - 1) / rpw);
  g.part_stride = ((long)K * C + C + 3) & ~3L;   // [K][C] weight gradient + [C] bias gradient, 16-byte rows
  return g;
}

}  // namespace

#ifdef CRNN_DSB_TRACE
static unsigned long long* g_dsb_trace = nullptr;
extern "C" void crnn_dense_bwd_small_set_trace(unsigned long long* t) { g_dsb_trace = t; }
#endif
extern "C" int crnn_dense_bwd_small_supported(long M, int K, int C) {
  return (M > 0 && K >= 128 && K % 128 == 0 && K <= kDsbMaxThreads && C >= 1 && C <= kDsbClasses) ? CRNN_OK : CRNN_ERR_UNSUPPORTED;
}
extern "C" size_t crnn_dense_bwd_small_scratch_bytes(long M, int K, int C) {
  if (crnn_dense_bwd_small_supported(M, K, C) != CRNN_OK) return 0;
  const DsbGeom g = dsb_geom(M, K, C);
  return (size_t)g.G * (size_t)g.part_stride * sizeof(float);
}
extern "C" int crnn_dense_bwd_small(const float* x, const float* dy, const float* W, float* dx, float* dW, float* db, float* scratch,
                                    size_t scratch_bytes, long M, int K, int C, int ldx, int lddx, const void* keep, float drop_rate, uint64_t seed,
                                    uint32_t layer, hipStream_t stream) {
  if (crnn_dense_bwd_small_supported(M, K, C) != CRNN_OK || ldx != K || lddx < K) return CRNN_ERR_UNSUPPORTED;   // x rows are contiguous (whole steps are one DMA range)
  if (!x || !dy || !W || !dx || !dW || !db || !scratch) return CRNN_ERR_ARG;
  // misaligned operands are a shape this kernel does not take, not a caller error: -3 sends crnn_backward_top to the GEMM + column-sum + dropout path
  // (as the other stream entry points do; round 6, ADVICE)
  if ((((uintptr_t)x | (uintptr_t)W | (uintptr_t)scratch) & 15) != 0 || ((uintptr_t)keep & 3) != 0) return CRNN_ERR_UNSUPPORTED;
  if (drop_rate < 0.f || drop_rate >= 1.f) return CRNN_ERR_ARG;
  if (scratch_bytes < crnn_dense_bwd_small_scratch_bytes(M, K, C)) return CRNN_ERR_ARG;
  if (db != dW + (long)K * C) return CRNN_ERR_UNSUPPORTED;   // the two gradients are one span of the gradient buffer (one second-stage sum)
  const DsbGeom g = dsb_geom(M, K, C);
  DsbParams p;
  p.x = x; p.dy = dy; p.W = W; p.dx = dx; p.part = scratch; p.keep = drop_rate > 0.f ? static_cast<const unsigned char*>(keep) : nullptr; p.M = M; p.K = K; p.C = C; p.ldx = ldx; p.lddx = lddx;
  p.rows_per_wg = g.rows_per_wg; p.part_stride = g.part_stride; p.rate = drop_rate; p.seed = seed; p.layer = layer;
#ifdef CRNN_DSB_TRACE
  p.trace = g_dsb_trace;
#endif
  const int lds = dsb_lds_bytes(K, C);      // <= 154 KiB (K = 512, C = 40)
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)dense_bwd_small_kernel<kDsbClasses>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(dense_bwd_small_kernel<kDsbClasses>, dim3(g.G), dim3(K / 2 + 64), lds, stream, p);
  CRNN_LAUNCH_CHECK();
  const int n = K * C + C;
  hipLaunchKernelGGL(dense_bwd_small_sum_kernel, dim3(cdiv(g.part_stride / 4, 32)), dim3(256), 0, stream, scratch, g.G, g.part_stride, n, dW);
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
