// Persistent, LDS-DMA-fed bf16 GEMM for the pointwise (1x1) convolutions of the conv stack (utils.py:49, Conv2D(1x1)):
//     Y[M][N] = X[M][K] . W[N][K]^T        X = pixels x channels-in (bf16, NHWC rows), W bf16, Y bf16, fp32 accumulate
// used for the data gradient (X = dq, W = the bf16 weight shadow [ci][co], Y = da) and, with W = the bf16 W^T copy, for any
// plain forward product.  M is ~10^5..10^6 rows, K and N are 64..512: the product is HBM-bound (arithmetic intensity
// 43..256 FLOP/B), so the design goal is to keep the pixel stream in flight all the time, not MFMA throughput.
//
// What the tile-per-workgroup kernel (gemm_bf16.inc) loses on these shapes -- measured by ablation (scripts/gemm_ablate.py):
// with loads, MFMAs and stores all removed it still needs 45-60 % of its time, i.e. each tile is a serial chain
// "wait for the first loads -> LDS -> barrier -> ... -> LDS-staged epilogue" and three resident workgroups per CU do not
// cover it (~25 KB of pixel rows in flight per CU against the ~50 KB that 8 TB/s x ~2.5 us of latency need).  Here:
//   * ONE persistent workgroup per CU walks over 128-row stripes; the k-chunks (64 k) of successive tiles form one
//     continuous stream through an LDS ring of R slots (R = 4 at 128 output channels per pass, 3 at 256), each holding 128
//     pixel rows (16 KiB) and the pass's weight rows; chunks i+1 .. i+R-1 are in flight while chunk i is multiplied;
//   * two LOADER waves (each half of both operands) issue nothing but global_load_lds (16 B per lane, straight to LDS, no VGPR
//     staging) and wait with counted s_waitcnt vmcnt(N); four COMPUTE waves never wait on the vector-memory counter, so
//     their result stores drain in the background (on gfx950 loads and stores share vmcnt: a wave that both prefetches
//     and stores cannot count);
//   * one raw s_barrier per k-chunk; LDS rows are 128 B unpadded with the 16-byte chunk c of row r stored at position
//     c ^ ((r >> 1) & 7) -- the permutation is applied to the per-lane GLOBAL address, the LDS image stays lane-linear
//     (LDS-DMA cannot scatter) -- which makes the ds_read_b128 fragment reads bank-conflict free;
//   * the weights are the MFMA A operand and the pixels the B operand (D = W X^T): a lane of v_mfma_f32_32x32x16_bf16 then
//     holds 4 consecutive output channels of ONE pixel per register group, v_permlane32_swap pairs two groups into 8
//     consecutive channels, and the result goes out as 16-byte stores straight from the accumulators -- no LDS
//     staging, no barrier in the epilogue.
// Numerics: the same MFMA, the same k order (64-k chunks, 16-k steps) as gemm_bf16.inc.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

struct NtParams {
  const bf16_t* X; const bf16_t* W; bf16_t* Y;
  int M, N, K;
  int stripes;   // ceil(M / 128)
  int nt;        // N / BN
  int kch;       // K / 64
};

constexpr int kStageA = 128 * 128;          // 128 pixel rows x 64 k x 2 B

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// R = ring depth (slots per operand): chunk i is consumed while chunks i+1 .. i+R-2 are in flight / landed and chunk i+R-1 is issued
template <int BN, int R>
__global__ __launch_bounds__(384) void gemm_nt_kernel(NtParams p) {
  constexpr int kStageB = BN * 128;
  constexpr int TN = BN / 64, TM = 2;                        // 32x32 tiles per compute wave: channels x pixels
  constexpr int IPS = 8 + BN / 16;                           // LDS-DMA instructions per loader wave and stage (half of each operand)
  static_assert((R - 2) * IPS <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) unsigned char smem[R * (kStageA + kStageB)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + R * kStageA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, wg = blockIdx.x;
  const int mine = (p.stripes - wg + G - 1) / G;             // stripes wg, wg + G, ...
  if (mine <= 0) return;
  const int per_stripe = p.nt * p.kch;
  const int total = mine * per_stripe;

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves: each streams half of both operands
    const int lw = wave - 4;
    const int rsub = lane >> 3, pos = lane & 7;
    const long ldk = p.K;
    auto issue = [&](int lin) {
      const int slot = lin % R;
      lin = lin < total ? lin : total - 1;              // past the end: re-read the last chunk into an already consumed slot
      const int sl = lin / per_stripe, rem = lin % per_stripe, tn = rem / p.kch, kc = rem % p.kch;
      const int r0 = (wg + sl * G) * 128;
      unsigned char* da = As + slot * kStageA;
      unsigned char* db = Bs + slot * kStageB;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = lw * 8 + jj;
        int row = r0 + 8 * j + rsub;
        row = row < p.M ? row : p.M - 1;
        const int c = pos ^ ((4 * j + (lane >> 4)) & 7);
        glds16(p.X + row * ldk + kc * 64 + c * 8, da + j * 1024);
      }
#pragma unroll
      for (int jj = 0; jj < BN / 16; ++jj) {
        const int j = lw * (BN / 16) + jj;
        const int row = tn * BN + 8 * j + rsub;
        const int c = pos ^ ((4 * j + (lane >> 4)) & 7);
        glds16(p.W + row * ldk + kc * 64 + c * 8, db + j * 1024);
      }
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue(s);
    for (int i = 0; i < total; ++i) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * IPS) : "memory");   // chunk i has landed; i+1 .. i+R-2 may still be in flight
      __builtin_amdgcn_s_barrier();
      issue(i + R - 1);                                                       // into the slot chunk i-1 has just released
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int half = lane >> 5, l31 = lane & 31, sw = (l31 >> 1) & 7;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  int sl = 0, tn = 0, kc = 0, slot = 0;
  for (int i = 0; i < total; ++i) {
    __builtin_amdgcn_s_barrier();
    const unsigned char* A = As + slot * kStageA + (wm * 64 + l31) * 128;
    const unsigned char* B = Bs + slot * kStageB + (wn * (BN / 2) + l31) * 128;
    slot = (slot + 1 == R) ? 0 : slot + 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = ((2 * ks + half) ^ sw) * 16;
      bf16x8_t fx[TM], fw[TN];
#pragma unroll
      for (int b = 0; b < TM; ++b) fx[b] = *reinterpret_cast<const bf16x8_t*>(A + b * 32 * 128 + off);
#pragma unroll
      for (int a = 0; a < TN; ++a) fw[a] = *reinterpret_cast<const bf16x8_t*>(B + a * 32 * 128 + off);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a], fx[b], acc[a][b], 0, 0, 0);
    }
    if (++kc == p.kch) {
      // ---- epilogue of (stripe, tn): lane = one pixel; register group g of a 32x32 block = channels 8g + 4half + 0..3
      kc = 0;
      const int m0 = (wg + sl * G) * 128 + wm * 64 + l31;
      const int n0 = tn * BN + wn * (BN / 2);
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + 32 * b;
        bf16_t* yrow = p.Y + (long)m * p.N + n0 + 8 * half;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const f32x16& v = acc[a][b];
            unsigned g0a = pack2_bf16(v[8 * pr + 0], v[8 * pr + 1]), g0b = pack2_bf16(v[8 * pr + 2], v[8 * pr + 3]);
            unsigned g1a = pack2_bf16(v[8 * pr + 4], v[8 * pr + 5]), g1b = pack2_bf16(v[8 * pr + 6], v[8 * pr + 7]);
            const u32x2 sa = __builtin_amdgcn_permlane32_swap(g0a, g1a, false, false);
            const u32x2 sb = __builtin_amdgcn_permlane32_swap(g0b, g1b, false, false);
            if (m < p.M) *reinterpret_cast<uint4*>(yrow + 32 * a + 16 * pr) = make_uint4(sa.x, sb.x, sa.y, sb.y);
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        }
      }
      if (++tn == p.nt) { tn = 0; ++sl; }
    }
  }
}

}  // namespace

// Y[M][N] (bf16) = X[M][K] (bf16, row stride K) . W[N][K]^T (bf16, row stride K).  Supported: K % 64 == 0, N % 128 == 0,
// 16-byte aligned pointers; anything else returns CRNN_ERR_UNSUPPORTED (use crnn_gemm_bf16_ex).  One persistent
// workgroup per CU (384 threads: 4 MFMA waves + 2 LDS-DMA loader waves).
extern "C" int crnn_gemm_nt_bf16(const void* X, const void* W, void* Y, int M, int N, int K, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return CRNN_ERR_ARG;
  if (K % 64 || N % 128 || (((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15)) return CRNN_ERR_UNSUPPORTED;
  if ((long)M * (K > N ? K : N) >= (1L << 31)) return CRNN_ERR_UNSUPPORTED;     // 32-bit row offsets
  NtParams p;
  p.X = (const bf16_t*)X; p.W = (const bf16_t*)W; p.Y = (bf16_t*)Y; p.M = M; p.N = N; p.K = K;
  p.stripes = cdiv(M, 128); p.kch = K / 64;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  const int grid = p.stripes < cus ? p.stripes : cus;
  const int variant = crnn_knob("CRNN_NT_VARIANT", 0);   // 1: 128 channels per pass + 4 slots, 2: 128 + 3 slots (experiment builds)
  if (N % 256 == 0 && variant != 1) {
    p.nt = N / 256;
    hipLaunchKernelGGL((gemm_nt_kernel<256, 3>), dim3(grid), dim3(384), 0, stream, p);
  } else {
    p.nt = N / 128;
    if (variant == 2) hipLaunchKernelGGL((gemm_nt_kernel<128, 3>), dim3(grid), dim3(384), 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<128, 4>), dim3(grid), dim3(384), 0, stream, p);
  }
  CRNN_LAUNCH_CHECK();
  return CRNN_OK;
}
