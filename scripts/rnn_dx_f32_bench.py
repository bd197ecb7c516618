#!/usr/bin/env python3
"""Parity mode, the recurrent layers' input gradients (M = 52 * 256 rows, K = 1024 gate columns, two directions): crnn_gemm_nt_f32x2_stream against two accumulating
crnn_gemm_f32x2 launches.  Median of 10 launches with a 512 MiB fill in between (operands cold), us.
(Round 6 also ran dense1's forward, K = 4608, on this stream with three planes: 123 us against 148 for the tile kernel + dropout pass, less the fp32 W^T copy it needs --
every 64-row stripe re-reads the 2.4 MB of fp32 weights, 0.74 GB through the L2s per launch -- and the step did not move: not kept.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, n=10):
    ts = []
    for it in range(n + 2):
        flush.fill_(it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
M = 52 * 256
scr = torch.empty(16 << 20, device="cuda"); sb = ctypes.c_size_t(scr.numel() * 4)
for din in (128, 256):
    G = 1024
    dz = [torch.randn(M, G, device="cuda") * 0.1 for _ in range(2)]; Wp = [torch.randn(din, G, device="cuda") * 0.05 for _ in range(2)]; dx = torch.empty(M, din, device="cuda")
    def s2():
        assert L.crnn_gemm_nt_f32x2_stream(P(dz[0]), P(Wp[0]), P(dz[1]), P(Wp[1]), P(dx), M, din, G, G, G, din, S()) == 0
    def t2():
        assert L.crnn_gemm_f32x2(1, P(dz[0]), P(Wp[0]), P(dx), M, din, G, G, G, din, None, 0, 0, 0, P(scr), sb, S()) == 0
        assert L.crnn_gemm_f32x2(1, P(dz[1]), P(Wp[1]), P(dx), M, din, G, G, G, din, None, 0, 1, 0, P(scr), sb, S()) == 0
    print("recurrent input gradient, din %d, two planes: stripe stream %.1f us, two tile launches %.1f us" % (din, timeit(s2), timeit(t2)))
