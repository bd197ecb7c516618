#!/usr/bin/env python3
"""crnn_gemm_nt_bf16 (persistent LDS-DMA kernel) vs crnn_gemm_bf16_ex mode 1 on the pointwise-conv data-gradient / forward shapes
at batch 256: time, achieved HBM bytes/s against the minimal traffic (read X once + write Y once)."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("b3 dgrad", 958464, 128, 256), ("b4 dgrad", 239616, 256, 256), ("b5 dgrad", 239616, 256, 512), ("b6 dgrad", 119808, 512, 512),
          ("b7 dgrad", 119808, 512, 512), ("b3 fwd", 958464, 256, 128), ("b5 fwd", 239616, 512, 256), ("b2 fwd", 958464, 128, 64)]
res = {}
for name, M, N, K in shapes:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def t(fn):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3
    def new():
        r = L.crnn_gemm_nt_bf16(P(X), P(W), P(Y), M, N, K, S()); assert r == 0, r
    def old():
        r = L.crnn_gemm_bf16_ex(1, P(X), P(W), P(Y), M, N, K, K, K, N, None, 0, 0, 0, None, 0, 1, 1, 1, S()); assert r == 0, r
    def wres():
        r = L.crnn_gemm_wres_bf16(P(X), P(W), P(Y), M, N, K, S()); assert r == 0, r
    tw = t(wres) if L.crnn_gemm_wres_supported(N, K) == 0 else float("nan")
    tn, to = t(new), t(old)
    os.environ["CRNN_NT_VARIANT"] = "1"; tn1 = t(new); os.environ["CRNN_NT_VARIANT"] = "2"; tn2 = t(new); os.environ["CRNN_NT_VARIANT"] = "0"
    by = 2.0 * (M * K + M * N)
    res[name] = {"M": M, "N": N, "K": K, "wres_us": round(tw, 1), "wres_TBps": round(by / tw / 1e6, 2), "persist_us": round(tn, 1), "tile_us": round(to, 1), "bn128_r4_us": round(tn1, 1), "bn128_r3_us": round(tn2, 1), "persist_TBps": round(by / tn / 1e6, 2), "tile_TBps": round(by / to / 1e6, 2)}
    print(name, res[name], flush=True)
