// Which access PATTERN does the memory system reward?  (round 5; companion of copy_probe.hip)  The row-stream depthwise kernels give every workgroup its
// own image band: 256 read streams and 256 write streams far apart, each advancing 9 KiB per step -- a plain copy with that pattern reaches 0.60-0.64 of the
// HBM peak where a copy in which the resident workgroups sweep ONE window together reaches 0.70 (nontemporal: 0.77).  This probe copies the same six tensor
// pairs (cold protocol of bench.py's depthwise_roofline) with S streams far apart, G workgroups cooperating on each stream (adjacent bursts of one window),
// `burst` bytes per workgroup and iteration, plain or nontemporal accesses, and two workgroup -> (stream, position) maps:
//   map 0: wg = stream * G + g   (the G workgroups of a stream have consecutive ids: spread over the XCDs, wg % 8)
//   map 1: wg = g * S + stream   (stream s stays on XCD s % 8 when S % 8 == 0)
// S = 1 is the sweep copy, (S = wgs, G = 1) the banded copy.
// build: hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/pattern_probe scripts/probes/pattern_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Pat { long n; int S, G, burst16, map, nt, mode; };   // n in 16-byte units; burst16 = 16-byte units per workgroup and iteration; mode 0 copy | 1 read only | 2 write only
template <int NT>
__global__ __launch_bounds__(256) void pattern_kernel(const u32x4* __restrict__ s, u32x4* __restrict__ d, Pat p) {
  const int wg = blockIdx.x;
  const int stream = p.map ? wg % p.S : wg / p.G, g = p.map ? wg / p.S : wg % p.G;
  const long per = (p.n + p.S - 1) / p.S;                       // units per stream
  const long lo = stream * per, hi = lo + per < p.n ? lo + per : p.n;
  const long step = (long)p.G * p.burst16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  if (p.mode == 3) {                                            // descending sweep (whole bursts; the tail burst first)
    const long nburst = (hi - lo + p.burst16 - 1) / p.burst16;
    for (long b = nburst - 1 - g; b >= 0; b -= p.G) {
      const long w0 = lo + b * p.burst16, w1 = w0 + p.burst16 < hi ? w0 + p.burst16 : hi;
      for (long i = w0 + threadIdx.x; i < w1; i += 1024) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < w1) v[u] = s[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * 256 < w1) d[i + u * 256] = v[u];
      }
    }
    return;
  }
  for (long w0 = lo + (long)g * p.burst16; w0 < hi; w0 += step) {
    const long w1 = w0 + p.burst16 < hi ? w0 + p.burst16 : hi;
    for (long i = w0 + threadIdx.x; i < w1; i += 1024) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p.mode == 2) { v[u] = u32x4{(unsigned)i, 1u, 2u, 3u}; continue; }
        if (i + u * 256 < w1) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p.mode == 1) { if (i + u * 256 < w1) { acc.x ^= v[u].x; acc.y += v[u].y; } continue; }
        if (i + u * 256 < w1) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
      }
    }
  }
  if (p.mode == 1 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) d[0] = acc;   // keeps the loads alive
}
int main() {
  const long B = 256;
  const long shp[6][3] = {{104, 36, 64}, {104, 36, 128}, {52, 18, 256}, {52, 18, 256}, {52, 9, 512}, {52, 9, 512}};
  void *src[6], *dst[6]; long bytes[6]; double total = 0;
  for (int i = 0; i < 6; ++i) {
    bytes[i] = B * shp[i][0] * shp[i][1] * shp[i][2] * 2;
    if (hipMalloc(&src[i], bytes[i]) != hipSuccess || hipMalloc(&dst[i], bytes[i]) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src[i], 1, bytes[i]); hipMemset(dst[i], 2, bytes[i]);
    total += 2.0 * bytes[i];
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int S, int G, int burst, int map, int nt, int mode = 0) {
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 6; ++i) {
        Pat p{bytes[i] / 16, S, G, burst / 16, map, nt, mode};
        if (nt) hipLaunchKernelGGL(pattern_kernel<1>, dim3(S * G), dim3(256), 0, 0, (const u32x4*)src[i], (u32x4*)dst[i], p);
        else hipLaunchKernelGGL(pattern_kernel<0>, dim3(S * G), dim3(256), 0, 0, (const u32x4*)src[i], (u32x4*)dst[i], p);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double t = ts[ts.size() / 2] * 1e-3, bytes_moved = mode ? total / 2 : total;
    printf("S %4d  G %4d  wgs %5d  burst %6d  map %d  %s %-5s  %7.1f us per set  %6.0f GB/s  %.3f of 8 TB/s\n", S, G, S * G, burst, map, nt ? "nt   " : "plain",
           mode == 0 ? "copy" : (mode == 1 ? "read" : "write"), t * 1e6, bytes_moved / t / 1e9, bytes_moved / t / 8e12);
    fflush(stdout);
  };
  printf("six tensor pairs of the depthwise forward (%.0f MB read + written per set), median of 11 cold sets; S streams x G cooperating workgroups\n", total / 1e6);
  for (int nt = 0; nt < 2; ++nt) {
    run(1, 256, 16384, 0, nt);                                  // sweep
    run(256, 1, 16384, 0, nt);                                  // banded
    for (int S : {128, 64, 32, 16, 8}) { run(S, 256 / S, 16384, 0, nt); run(S, 256 / S, 16384, 1, nt); }
  }
  for (int burst : {4096, 8192, 9216, 32768, 65536}) run(256, 1, burst, 0, 0);            // banded: burst size
  for (int S : {256, 128, 64, 32}) { run(S, 512 / S, 16384, 0, 0); run(S, 512 / S, 16384, 1, 0); }   // two workgroups per CU
  for (int S : {256, 64, 16}) { run(S, 1024 / S, 16384, 1, 0); }                          // four per CU
  run(64, 4, 36864, 0, 0); run(64, 4, 36864, 1, 0); run(32, 8, 18432, 1, 0);             // four (eight) CUs sharing an image band, 36 (18) KiB per step
  for (int mode : {1, 2}) { run(1, 256, 16384, 0, 0, mode); run(256, 1, 16384, 0, 0, mode); run(64, 4, 16384, 1, 0, mode); }
  // ---- (2) the same sweep / banded pair inside ONE big allocation (the engine's workspace is one 7 GB tensor; the six pairs above are 12 separate
  //          hipMalloc regions): does the gap between the two patterns depend on where the buffers live?
  {
    void* big = nullptr; const size_t sz = 4ull << 30;
    if (hipMalloc(&big, sz) == hipSuccess) {
      hipMemset(big, 3, sz);
      size_t off = 4096 * 37;                                   // (an offset that is not 2 MiB aligned, like the workspace's sub-tensors)
      void *s2[6], *d2[6];
      for (int i = 0; i < 6; ++i) { s2[i] = (char*)big + off; off += bytes[i] + 256 * 13; d2[i] = (char*)big + off; off += bytes[i] + 256 * 7; }
      if (off <= sz) {
        void* keep_s[6]; void* keep_d[6];
        for (int i = 0; i < 6; ++i) { keep_s[i] = src[i]; keep_d[i] = dst[i]; src[i] = s2[i]; dst[i] = d2[i]; }
        printf("the same inside one 4 GiB allocation (sub-buffers at odd offsets):\n");
        run(1, 256, 16384, 0, 0); run(256, 1, 16384, 0, 0); run(1, 256, 16384, 0, 1); run(256, 1, 16384, 0, 1); run(128, 4, 16384, 1, 0);
        for (int i = 0; i < 6; ++i) { src[i] = keep_s[i]; dst[i] = keep_d[i]; }
      }
      hipFree(big);
    }
  }
  // ---- (3) producer -> consumer through the 256 MB last-level cache: a producer copy writes T (123 MB, ascending addresses); the consumer copy reads T
  //          ascending (the producer's oldest lines first) or DESCENDING (its newest first).  Only the consumer is timed.  If the cache keeps the most
  //          recently written half of T, the descending consumer finds it there.
  {
    const long nb = bytes[2];                                   // 122.7 MB
    void *A = src[2], *T = dst[2], *O = src[3];
    for (int desc = 0; desc < 2; ++desc)
      for (int wgs : {256, 2048}) {
        std::vector<float> ts;
        for (int it = 0; it < 10; ++it) {
          // evict: stream two other pairs through the cache
          Pat pe{bytes[0] / 16, 1, 256, 1024, 0, 0, 0};
          hipLaunchKernelGGL(pattern_kernel<0>, dim3(256), dim3(256), 0, 0, (const u32x4*)src[0], (u32x4*)dst[0], pe);
          Pat pe1{bytes[1] / 16, 1, 256, 1024, 0, 0, 0};
          hipLaunchKernelGGL(pattern_kernel<0>, dim3(256), dim3(256), 0, 0, (const u32x4*)src[1], (u32x4*)dst[1], pe1);
          Pat pp{nb / 16, 1, wgs, 1024, 0, 0, 0};               // producer: A -> T, ascending sweep
          hipLaunchKernelGGL(pattern_kernel<0>, dim3(wgs), dim3(256), 0, 0, (const u32x4*)A, (u32x4*)T, pp);
          hipEventRecord(e0, 0);
          Pat pc{nb / 16, 1, wgs, 1024, 0, 0, desc ? 3 : 0};    // consumer: T -> O
          hipLaunchKernelGGL(pattern_kernel<0>, dim3(wgs), dim3(256), 0, 0, (const u32x4*)T, (u32x4*)O, pc);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (it) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double t = ts[ts.size() / 2] * 1e-3;
        printf("consumer of a just-written 123 MB tensor, %4d workgroups, %s: %6.1f us  %6.0f GB/s (read + write)\n", wgs, desc ? "DESCENDING" : "ascending ", t * 1e6,
               2.0 * nb / t / 1e9);
        fflush(stdout);
      }
  }
  // ---- (4) a second pass over a working set LARGER than the last-level cache: pass 1 reads X (491 MB = q2 + dx of block 2's BatchNorm-2 backward) ascending,
  //          pass 2 reads X again and writes Y (245 MB), ascending or DESCENDING.  Only pass 2 is timed.
  {
    void *X = nullptr, *Y = nullptr; const long xb = 2 * bytes[1], yb = bytes[1];
    if (hipMalloc(&X, xb) == hipSuccess && hipMalloc(&Y, yb) == hipSuccess) {
      hipMemset(X, 5, xb); hipMemset(Y, 6, yb);
      for (int desc = 0; desc < 2; ++desc) {
        std::vector<float> ts;
        for (int it = 0; it < 8; ++it) {
          Pat p1{xb / 16, 1, 2048, 1024, 0, 0, 1};              // pass 1: read-only ascending sweep
          hipLaunchKernelGGL(pattern_kernel<0>, dim3(2048), dim3(256), 0, 0, (const u32x4*)X, (u32x4*)Y, p1);
          hipEventRecord(e0, 0);
          Pat p2a{yb / 16, 1, 2048, 1024, 0, 0, desc ? 3 : 0};  // pass 2: the two halves of X -> Y (two launches: 2 x 245 MB read, 2 x 245 MB written)
          if (desc) {
            hipLaunchKernelGGL(pattern_kernel<0>, dim3(2048), dim3(256), 0, 0, (const u32x4*)((char*)X + yb), (u32x4*)Y, p2a);
            hipLaunchKernelGGL(pattern_kernel<0>, dim3(2048), dim3(256), 0, 0, (const u32x4*)X, (u32x4*)Y, p2a);
          } else {
            hipLaunchKernelGGL(pattern_kernel<0>, dim3(2048), dim3(256), 0, 0, (const u32x4*)X, (u32x4*)Y, p2a);
            hipLaunchKernelGGL(pattern_kernel<0>, dim3(2048), dim3(256), 0, 0, (const u32x4*)((char*)X + yb), (u32x4*)Y, p2a);
          }
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (it) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double t = ts[ts.size() / 2] * 1e-3;
        printf("second pass over a 491 MB working set just read ascending, %s: %6.1f us  %6.0f GB/s (read + write)\n", desc ? "DESCENDING" : "ascending ", t * 1e6, 2.0 * xb / t / 1e9);
        fflush(stdout);
      }
    }
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
