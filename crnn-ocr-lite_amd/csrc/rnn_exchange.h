// Exchange machinery shared by the persistent recurrences (rnn_persist.hip: LSTM, gru_persist.hip: GRU): the sentinel ring in
// device memory through which the workgroups of a cluster all-gather a step's values, its status words, the workgroup ->
// (cluster, member) maps and the host-side sizing of a launch.  See the header comment of rnn_persist.hip for the protocol.
#pragma once
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kSentinel = 0xffffffffu;
#ifdef CRNN_RNN_TRACE
constexpr size_t kStatusBytes = 65536;        // trace build: [64..) = per-step timestamps of two workgroups
#define RNN_TRACE(slot) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3)) \
    reinterpret_cast<unsigned long long*>(xbuf + 64)[((blockIdx.x != 0) * 128 + TRACE_STEP) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr size_t kStatusBytes = 256;          // byte 0: sticky give-up counter, byte 16: per-launch status; the exchange tiles follow
#define RNN_TRACE(slot) do {} while (0)
#endif
constexpr unsigned long long kSpinTicks = 200ull * 1000 * 1000;   // a wait gives up after 2 s of the constant 100 MHz clock (s_memrealtime)
constexpr size_t kLaunchStatusOff = 16;       // the fill of every launch starts here (the counter in front of it survives)
// cache policy of the exchange (aux bits of the buffer instructions: 1 = sc0, 16 = sc1).  Default: write-through stores and
// L1-bypassing loads at device scope (works for any placement).  Other values exist for scripts/lstm_xcd_bench.py only.
#ifndef CRNN_RNN_POL
#define CRNN_RNN_POL 0
#endif
#if CRNN_RNN_POL == 0
constexpr int kAuxSt = 16, kAuxLd = 16;
#elif CRNN_RNN_POL == 1
constexpr int kAuxSt = 0, kAuxLd = 16;
#elif CRNN_RNN_POL == 2
constexpr int kAuxSt = 1, kAuxLd = 1;
#else
constexpr int kAuxSt = 17, kAuxLd = 17;
#endif
constexpr int kRing = 4;                      // step slots of the exchange ring

template <bool WBF> struct XE { typedef float type; };
template <> struct XE<true> { typedef bf16_t type; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned pack_e(bf16_t*, float lo, float hi) { return pack2_bf16(lo, hi); }

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

// Workgroups that are certainly co-resident on the device: per CU as many as the LDS footprint and the thread count
// (2048 threads, and the registers of two 256-thread workgroups / one 1024-thread workgroup) admit, at most 2.
inline int resident_cap(size_t lds_bytes, int threads) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  int per_cu = (lds_bytes * 2 <= 160 * 1024) ? 2 : 1;
  if (threads * per_cu > 1024) per_cu = 1;
  return cus * per_cu;
}
// rows of the batch one launch covers, and the exchange bytes that launch needs
struct Chunking { int rows_per_launch; size_t xdata_bytes; };
inline Chunking chunking(int T, int B, int u, int mt, int uw, int es, size_t lds, int per_row) {
  const int NSW = u / (16 * uw), BT = 16 * mt;
  int tiles = resident_cap(lds, 256 * uw) / (2 * NSW);
  if (tiles < 1) tiles = 1;
  Chunking c;
  c.rows_per_launch = tiles * BT;
  const int rows = (B < c.rows_per_launch) ? cdiv(B, BT) * BT : c.rows_per_launch;
  c.xdata_bytes = 2 * (size_t)kRing * rows * per_row * es;
  return c;
}

int prep_xbuf(void* xbuf, size_t xbuf_bytes, size_t need_data, bool reset_status, hipStream_t stream) {
  if (!xbuf || xbuf_bytes < kStatusBytes + need_data || ((uintptr_t)xbuf & 15)) return CRNN_ERR_ARG;
  // one fill: the status words (all ones = no wait gave up) and the sentinel ring behind them
  // (the sticky give-up counter in front of the per-launch word is never touched)
  unsigned char* p = (unsigned char*)xbuf + (reset_status ? kLaunchStatusOff : kStatusBytes);
  hipError_t e = hipMemsetAsync(p, 0xFF, need_data + (reset_status ? kStatusBytes - kLaunchStatusOff : 0), stream);
  return e == hipSuccess ? CRNN_OK : (int)e;
}

}  // namespace
