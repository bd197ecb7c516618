#!/usr/bin/env python3
"""Ablation of the bf16 NT GEMM (pointwise-conv forward / data-gradient shapes at batch 256) with the experiment build of
gemm.hip (scripts/_trace/libgemm_exp.so, -DCRNN_GEMM_EXP): CRNN_GEMM_EXP bits 1 = no C stores, 2 = no MFMA, 4 = B loaded once,
8 = A loaded once.  Prints time per variant."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", "libgemm_exp.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("b3 fwd", 958464, 256, 128), ("b4 fwd", 239616, 256, 256), ("b6 fwd", 119808, 512, 512), ("b3 dgrad", 958464, 128, 256), ("b5 dgrad", 239616, 256, 512)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); Bm = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for exp in (0, 1, 2, 3, 4, 8, 12, 14, 15):
        os.environ["CRNN_GEMM_EXP"] = str(exp)
        def run():
            r = lib.crnn_gemm_bf16_ex(1, P(A), P(Bm), P(C), M, N, K, K, K, N, None, 0, 0, 0, None, ctypes.c_size_t(0), 1, 1, 1, S())
            assert r == 0, r
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        row.append("%d:%.1f" % (exp, e0.elapsed_time(e1) / 5 * 1e3))
    by = 2.0 * (M * K + M * N)
    print("%-9s M=%7d N=%4d K=%4d  min bytes %.0f MB (%.0f us @5TB/s) | us by variant: %s" % (name, M, N, K, by / 1e6, by / 5e6, "  ".join(row)))
