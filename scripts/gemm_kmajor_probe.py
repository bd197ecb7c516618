#!/usr/bin/env python3
"""Weight-gradient GEMM (both operands k-major, bf16) with a given build of gemm.hip (argv[1] = .so): time for two step shapes."""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(sys.argv[1])
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scr = torch.empty(16 * 1024 * 1024, device="cuda")
for name, Kd, M, N in (("b7 wgrad", 119808, 512, 512), ("b4 wgrad", 239616, 256, 256)):
    A = torch.randn(Kd, M, device="cuda").to(torch.bfloat16); B = torch.randn(Kd, N, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    def run():
        r = lib.crnn_gemm_bf16_ex(2, P(A), P(B), P(C), M, N, Kd, M, N, N, None, 0, 0, 0, P(scr), ctypes.c_size_t(64 * 1024 * 1024), 1, 1, 0, S()); assert r == 0, r
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(sys.argv[1]), name, "%.1f us" % (e0.elapsed_time(e1) / 5 * 1e3), flush=True)
