#!/usr/bin/env python3
"""Ablation of the weights-resident GEMM (experiment builds scripts/_trace/libwres_exp<mask>.so, -DCRNN_WRES_EXP=<mask>, compile-time):
CRNN_WRES_EXP bits: 1 no pixel loads, 2 no fragment reads, 8 no MFMAs, 4 no stores; CRNN_WRES_NLW = loader waves (2 | 4)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
libs = {}
def lib(exp):
    if exp not in libs: libs[exp] = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwres_exp%d.so" % exp))
    return libs[exp]
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("b3 dgrad", 958464, 128, 256), ("b4 dgrad", 239616, 256, 256), ("b7 dgrad", 119808, 512, 512), ("b2 fwd", 958464, 128, 64)]
variants = [(0, 2), (0, 4), (1, 2), (2 | 8, 2), (2 | 8, 4), (8, 2), (4, 2), (2 | 8 | 4, 2), (2 | 8 | 4, 4), (1 | 4, 2)]
if len(sys.argv) > 1: variants = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for name, M, N, K in shapes:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = "%-9s" % name
    for exp, nlw in variants:
        os.environ["CRNN_WRES_EXP"] = str(exp); os.environ["CRNN_WRES_NLW"] = str(nlw)
        L = lib(exp)
        fn = lambda: L.crnn_gemm_wres_bf16(P(X), P(W), P(Y), M, N, K, S())
        for _ in range(2): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        line += "  exp%d/lw%d %6.1f" % (exp, nlw, e0.elapsed_time(e1) / 5 * 1e3)
    print(line, flush=True)
