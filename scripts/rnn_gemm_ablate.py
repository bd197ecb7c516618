#!/usr/bin/env python3
"""Ablation of the tile GEMM on the RNN-projection shapes (experiment build scripts/_trace/libgemm_exp.so): CRNN_GEMM_EXP bits
1 = no C stores, 2 = no MFMA, 4 = B loaded once, 8 = A loaded once."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", "libgemm_exp.so"))
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
TB, G, u, tds = 13312, 1024, 256, 128
scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
mk = lambda *s: torch.randn(*s, device="cuda")
cases = [("xw1", 0, TB, G, tds, mk(TB, tds), mk(tds, G), mk(TB, G), tds, G, G, mk(G)),
         ("dW2", 2, u, G, TB, mk(TB, u), mk(TB, G), mk(u, G), u, G, G, None),
         ("dx2", 1, TB, u, G, mk(TB, G), mk(u, G), mk(TB, u), G, G, u, None)]
for name, mode, M, N, K, A, B, C, lda, ldb, ldc, bias in cases:
    row = []
    for exp in (0, 1, 2, 3, 4, 8, 12, 15):
        os.environ["CRNN_GEMM_EXP"] = str(exp)
        def run():
            r = lib.crnn_gemm_bf16_ex(mode, P(A), P(B), P(C), M, N, K, lda, ldb, ldc, P(bias), 0, 0, 0, P(scratch), ctypes.c_size_t(scratch.numel()), 0, 0, 0, S())
            assert r == 0, r
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        row.append("%d:%.1f" % (exp, e0.elapsed_time(e1) / 5 * 1e3))
    print("%-5s mode %d M=%5d N=%4d K=%5d | us by variant: %s" % (name, mode, M, N, K, "  ".join(row)), flush=True)

# in-tile timeline (workgroup 300, thread 0) of the full kernel
trace = torch.zeros(32, dtype=torch.int64, device="cuda")
os.environ["CRNN_GEMM_EXP"] = "0"; os.environ["CRNN_GEMM_TRACE"] = str(trace.data_ptr())
for name, mode, M, N, K, A, B, C, lda, ldb, ldc, bias in cases:
    trace.zero_()
    for _ in range(3):
        lib.crnn_gemm_bf16_ex(mode, P(A), P(B), P(C), M, N, K, lda, ldb, ldc, P(bias), 0, 0, 0, P(scratch), ctypes.c_size_t(scratch.numel()), 0, 0, 0, S())
    torch.cuda.synchronize()
    t = trace.cpu().numpy(); t = t[t > 0]
    print(name, "stamps (ns since first):", " ".join(str(int((v - t[0]) * 10)) for v in t))
