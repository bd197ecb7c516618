#!/usr/bin/env python3
"""Tile-GEMM time against the number of tiles (mode 0, fp32 in/out, N = 1024, K = 128): separates per-tile latency from throughput."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for K in (128, 512):
    for M in (128, 512, 2048, 4096, 8192, 13312, 26624, 53248):
        N = 1024
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
        run = lambda: L.crnn_gemm_bf16_ex(0, P(A), P(B), P(C), M, N, K, K, N, N, P(bias), 0, 0, 0, P(scratch), ctypes.c_size_t(scratch.numel()), 0, 0, 0, S())
        for _ in range(3): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print("K=%3d M=%6d tiles %5d: %6.1f us   %.2f TB/s out" % (K, M, (M // 128) * 8, us, M * N * 4 / us / 1e6), flush=True)
