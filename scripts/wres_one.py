#!/usr/bin/env python3
"""One shape of the weights-resident GEMM under the experiment build, for counter passes: wres_one.py M N K [exp] [nlw]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
L = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwres_exp%s.so" % (sys.argv[4] if len(sys.argv) > 4 and sys.argv[4].isdigit() else "0")))
M, N, K = (int(v) for v in sys.argv[1:4])
os.environ["CRNN_WRES_EXP"] = sys.argv[4] if len(sys.argv) > 4 else "0"
os.environ["CRNN_WRES_NLW"] = sys.argv[5] if len(sys.argv) > 5 else "2"
X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
trace = torch.zeros(64 * 4 + 256, dtype=torch.int64, device="cuda")
if "--trace" in sys.argv: os.environ["CRNN_WRES_TRACE"] = str(trace.data_ptr())
for _ in range(4):
    assert L.crnn_gemm_wres_bf16(ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(Y.data_ptr()), M, N, K, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
if "--trace" in sys.argv:
    c = trace.cpu().numpy()[256:].reshape(-1, 2); t = trace.cpu().numpy()[:256].reshape(64, 4)
    t0 = t[0, 0]
    print("iter: wait-start  waited  barrier  issue   (ns; s_memrealtime ticks of 10 ns)")
    for i in range(40):
        print("%3d  %8d  %6d  %6d  %6d" % (i, (t[i, 0] - t0) * 10, (t[i, 1] - t[i, 0]) * 10, (t[i, 2] - t[i, 1]) * 10, (t[i, 3] - t[i, 2]) * 10))
    print("compute wave 0: stage: end-of-MFMAs (since previous barrier release)  barrier wait")
    for i in range(1, 40):
        print("%3d  work %6d ns   barrier %6d ns" % (i, (c[i, 0] - c[i - 1, 1]) * 10, (c[i, 1] - c[i, 0]) * 10))
