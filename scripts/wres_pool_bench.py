#!/usr/bin/env python3
"""Pooled inference pointwise convs (crnn_pwconv_fwd_wres_folded_pool) at batch 1024 on the CRNN's two pooled blocks, stand-alone.
usage: wres_pool_bench.py [lib under scripts/_trace ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
P = lambda t: ctypes.c_void_p(t.data_ptr()); S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = sys.argv[1:] or ["libwres_pool_epi.so"]
for name in libs:
    L = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", name))
    L.crnn_pwconv_fwd_wres_folded_pool.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.crnn_pwconv_fwd_wres_folded.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for (M, N, K, pool) in ((1024 * 104 * 36, 256, 128, 4), (1024 * 52 * 18, 512, 256, 2)):
        a = torch.randn(M, K, device="cuda").abs().bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
        st = torch.cat([torch.zeros(N), torch.ones(N), torch.ones(N), torch.zeros(N)]).cuda()
        y = torch.empty(M // pool, N, dtype=torch.bfloat16, device="cuda"); yf = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        def t(fn):
            for _ in range(2): assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 5 * 1e3
        tp = t(lambda: L.crnn_pwconv_fwd_wres_folded_pool(P(a), P(W), P(y), M, N, K, P(st), pool, S()))
        tf = t(lambda: L.crnn_pwconv_fwd_wres_folded(P(a), P(W), P(yf), M, N, K, P(st), S()))
        print("%-26s M %8d N %3d K %3d pool %d: pooled %6.1f us   folded, un-pooled %6.1f us" % (name, M, N, K, pool, tp, tf))
        del a, W, y, yf
