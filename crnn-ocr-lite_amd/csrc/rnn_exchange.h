// Exchange machinery shared by the persistent recurrences (rnn_persist.hip: LSTM, gru_persist.hip: GRU): the sentinel ring in
// device memory through which the workgroups of a cluster all-gather a step's values, its status words, the workgroup ->
// (cluster, member) maps and the host-side sizing of a launch.  See the header comment of rnn_persist.hip for the protocol.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kSentinel = 0xffffffffu;
// Layout of xbuf: [0,4) sticky give-up counter (never reset by a launch) | [16,20) per-launch status (all ones = clean) | [kHelloOff, +kHelloBytes)
// one word per workgroup of the launch for the XCD census | (trace build: time stamps) | the exchange ring.  Every launch fills
// [16, end) with 0xFF first.
constexpr size_t kLaunchStatusOff = 16;
constexpr size_t kHelloOff = 256, kHelloBytes = 4096;     // 1024 workgroups (a launch has at most 2 per CU)
#ifdef CRNN_RNN_TRACE
constexpr size_t kTraceOff = kHelloOff + kHelloBytes;
constexpr size_t kStatusBytes = kTraceOff + 65536;        // trace build: per-step timestamps of two workgroups
#define RNN_TRACE(slot) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3)) \
    reinterpret_cast<unsigned long long*>(xbuf + kTraceOff)[((blockIdx.x != 0) * 128 + TRACE_STEP) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr size_t kStatusBytes = kHelloOff + kHelloBytes;
#define RNN_TRACE(slot) do {} while (0)
#endif
constexpr unsigned long long kSpinTicks = 200ull * 1000 * 1000;   // a wait gives up after 2 s of the constant 100 MHz clock (s_memrealtime)
// Cache policy of the exchange (aux bits of the buffer instructions: 1 = sc0, 16 = sc1).  Loads always bypass the L1 (sc1: a CU's L1
// is never refreshed by other CUs' stores).  Stores: write-through to memory (sc1) unless the cluster has verified that all its members
// run on ONE XCD -- then plain stores (the XCD's L2 is the coherence point of its CUs): measured at B = 256, u = 256, bf16
// (profiles/r03_lstm_cache_policy.txt) LSTM forward 131 us (linear map, sc1) / 143 (XCD-local, sc1) / 98 (XCD-local, plain), BPTT
// 198 / 174-185 / 142; sc0 loads or stores read stale L1 lines (every wait gives up).
constexpr int kAuxSt = 16, kAuxLd = 16, kAuxStLocal = 0;
constexpr int kRing = 4;                      // step slots of the exchange ring

template <bool WBF> struct XE { typedef float type; };
template <> struct XE<true> { typedef bf16_t type; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned pack_e(bf16_t*, float lo, float hi) { return pack2_bf16(lo, hi); }
// 16-byte store into an exchange tile: plain when the cluster is known to share an XCD, else write-through (device scope)
__device__ __forceinline__ void xstore(const u32x4& v, __amdgpu_buffer_rsrc_t rs, int byte_off, bool local) {
  if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, kAuxStLocal);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, kAuxSt);
}

// Poll NCH 16-byte chunks of an exchange tile (chunk idx = tid + NT*i) until none carries the sentinel, then hand
// them to `sink(idx, value)`.  Groups of <= 8 chunks per thread bound the register footprint.
template <int NCH, int NT, typename Sink>
__device__ __forceinline__ void gather_tile(const void* tile, int tid, unsigned* status, bool& dead, Sink sink) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(tile, NCH * 16);
  constexpr int NLD = (NCH + NT - 1) / NT;
  constexpr int GRP = NLD < 8 ? NLD : 8;
#pragma unroll 1
  for (int i0 = 0; i0 < NLD; i0 += GRP) {
    u32x4 v[GRP];
    unsigned spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
      bool ok = true;
      asm volatile("" ::: "memory");   // the tile changes under us: every pass must re-issue its loads
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
        const int idx = tid + NT * (i0 + i);
        if (i0 + i < NLD && idx < NCH) {
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16, 0, kAuxLd);   // aux 16 = sc1: served past the L1
          ok &= (v[i].x != kSentinel) & (v[i].y != kSentinel) & (v[i].z != kSentinel) & (v[i].w != kSentinel);
        }
      }
      if (__all(ok) || dead) break;
      // bounded wait: the clock is read once per 1024 polls (a hand-off normally completes within a few), 2 s after the first reading
      // the wave stops waiting
      bool expired = false;
      if ((++spins & 1023u) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        else expired = now - t0 > kSpinTicks;
      }
      if (expired) {
        if ((tid & 63) == 0) {
          atomicAnd(status + kLaunchStatusOff / 4, ~1u);   // the per-launch word starts as all ones (one fill covers it and the ring)
          atomicAdd(status, 1u);                           // sticky: never reset by a launch
        }
        dead = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      const int idx = tid + NT * (i0 + i);
      if (i0 + i < NLD && idx < NCH) sink(idx, v[i]);
    }
  }
}

// Workgroup -> (cluster, member).  Linear: a cluster's members have consecutive block ids, which the round-robin dispatch deals over
// all 8 XCDs.  XCD-local (xmap): workgroup id % 8 is its XCD, so the members of a cluster are the ids congruent modulo 8 inside a
// group of 8*NSW ids -- the whole all-gather of a chain then stays inside one XCD's L2 domain.  Needs (#clusters % 8 == 0); the
// results do not depend on the map.
__device__ __forceinline__ int cluster_block_id(int id, int nsw, int xmap) {
  if (!xmap) return id;
  const int xcd = id & 7, loc = id >> 3;
  return ((loc / nsw) * 8 + xcd) * nsw + loc % nsw;
}

// XCD census of a cluster (XCD-local map only): HIP promises nothing about workgroup -> XCD placement ("block b runs on XCD b % 8" is
// an observation), so plain L2-resident stores are only used after the members have PROVED they share an XCD: every member writes its
// HW_REG_XCC_ID (write-through) into its word of the hello table, wave 0 polls the cluster's words (L1-bypassing, bounded like every
// wait) and the workgroup takes the verdict "all equal".  Every member sees the same words, so the cluster agrees; one hand-off
// (about 1 us) per launch.  A member that never shows up ends the wait like any other lost hand-off (give-up counter, device-scope stores).
__device__ __forceinline__ bool cluster_shares_xcd(unsigned char* xbuf, int cl, int sl, int nsw, int tid, unsigned* status, bool& dead) {
  __shared__ int verdict;
  const unsigned myx = __builtin_amdgcn_s_getreg(0x1814) & 0xfu;        // hwreg(HW_REG_XCC_ID, 0, 4)
  if (tid < 64) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(xbuf + kHelloOff + (size_t)cl * nsw * 4, nsw * 4);
    if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(myx, rs, sl * 4, 0, kAuxSt);
    unsigned v = myx, spins = 0;
    unsigned long long t0 = 0;
    bool lost = false;
    for (;;) {
      asm volatile("" ::: "memory");
      v = (tid < nsw) ? __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, 0, kAuxLd) : myx;
      if (__all(v != kSentinel)) break;
      if ((++spins & 1023u) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        else if (now - t0 > kSpinTicks) { lost = true; break; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (lost && tid == 0) { atomicAnd(status + kLaunchStatusOff / 4, ~1u); atomicAdd(status, 1u); }
    const bool same = __all(v == myx) && !lost;
    if (tid == 0) verdict = same ? 1 : (lost ? -1 : 0);
  }
  __syncthreads();
  if (verdict < 0) dead = true;
  return verdict > 0;
}

// Workgroups that are certainly co-resident on the device: per CU as many as the LDS footprint and the thread count admit, at most
// 2 -- and no more than the runtime's occupancy figure for THIS kernel (`fn`), which also knows its register allocation: two
// 512-thread workgroups share a CU only below 128 VGPRs, and several instantiations (wide fp32 tiles) need more.  A grid sized past
// what is resident would stall every cluster wait for its 2 s bound.
inline int resident_cap(size_t lds_bytes, int threads, const void* fn) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  int per_cu = (lds_bytes * 2 <= 160 * 1024) ? 2 : 1;
  if (threads * per_cu > 1024) per_cu = 1;
  if (fn && per_cu > 1) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, 0) == hipSuccess && nb >= 1 && nb < per_cu) per_cu = nb;
  }
  return cus * per_cu;
}
// rows of the batch one launch covers, and the exchange bytes that launch needs
struct Chunking { int rows_per_launch; size_t xdata_bytes; };
inline Chunking chunking(int T, int B, int u, int mt, int uw, int es, size_t lds, int per_row, const void* fn) {
  const int NSW = u / (16 * uw), BT = 16 * mt;
  int tiles = resident_cap(lds, 256 * uw, fn) / (2 * NSW);
  if (tiles < 1) tiles = 1;
  Chunking c;
  c.rows_per_launch = tiles * BT;
  const int rows = (B < c.rows_per_launch) ? cdiv(B, BT) * BT : c.rows_per_launch;
  c.xdata_bytes = 2 * (size_t)kRing * rows * per_row * es;
  return c;
}

int prep_xbuf(void* xbuf, size_t xbuf_bytes, size_t need_data, hipStream_t stream) {
  if (!xbuf || xbuf_bytes < kStatusBytes + need_data || ((uintptr_t)xbuf & 15)) return CRNN_ERR_ARG;
  // one fill per launch: the per-launch status word (all ones = no wait gave up), the hello table and the sentinel ring behind them;
  // the sticky give-up counter in front is never touched
  hipError_t e = hipMemsetAsync((unsigned char*)xbuf + kLaunchStatusOff, 0xFF, kStatusBytes - kLaunchStatusOff + need_data, stream);
  return e == hipSuccess ? CRNN_OK : (int)e;
}

}  // namespace
