/* Measurement / test hooks of the MI355X CRNN build: libcrnn_testhooks.so (csrc/testhooks.hip).  Not part of the drop-in boundary
 * (include/crnn_mi355x.h) and never loaded by the product path: bench.py's copy reference, scripts/ and tests/ only. */
#ifndef CRNN_TESTHOOKS_H
#define CRNN_TESTHOOKS_H
#include <stddef.h>
#include "crnn_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook (tests/test_gpu_ops.py: the give-up path of the persistent recurrences): `blocks` workgroups of 64 threads that each pin
 * `lds_bytes` of LDS (<= 160 KiB: nothing else fits next to one on its CU) and spin for `microseconds` of the constant 100 MHz clock. */
int crnn_debug_occupy(int blocks, int lds_bytes, long microseconds, crnn_stream_t stream);
/* Measurement reference (bench.py "copy_reference"): dst[0..bytes) = src[0..bytes) (16-byte aligned, bytes % 16 == 0) by `workgroups` workgroups
 * of 256 threads.  pattern 0: grid-stride (the resident workgroups sweep one window together); pattern 1: workgroup b copies its own
 * contiguous 1/workgroups of the buffer -- the access pattern of the row-stream depthwise kernels (one image band per workgroup). */
int crnn_debug_copy(const void* src, void* dst, size_t bytes, int pattern, int workgroups, crnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
