#!/usr/bin/env python3
"""Ablation of the forward weights-resident pointwise kernel (compile-time masks, scripts/_trace/libwres_exp<mask>.so):
64 no BatchNorm/ReLU6 transform (raw copy), 128 no statistics arithmetic, 4 no global stores, 8 no MFMAs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("b2", 958464, 128, 64), ("b3", 958464, 256, 128), ("b4", 239616, 256, 256), ("b5", 239616, 512, 256), ("b6", 119808, 512, 512)]
masks = [int(a) for a in sys.argv[1:]] or [0, 64, 128, 192, 196, 8, 4]
libs = {m: ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwres_exp%d.so" % m)) for m in masks}
for name, M, N, K in shapes:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); st = torch.randn(4 * K, device="cuda").abs() + 0.5
    line = "%-3s" % name
    for m in masks:
        L = libs[m]
        L.crnn_pwconv_fwd_wres_rows.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
        rows = L.crnn_pwconv_fwd_wres_rows(M, N, K); parts = torch.empty(rows * 2 * N, device="cuda")
        L.crnn_pwconv_bnrelu6_fwd_wres.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        fn = lambda: L.crnn_pwconv_bnrelu6_fwd_wres(P(X), P(st), P(W), P(Y), M, N, K, P(parts), S())
        for _ in range(2): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        line += "  exp%-3d %6.1f" % (m, e0.elapsed_time(e1) / 5 * 1e3)
    print(line, flush=True)
