#!/usr/bin/env python3
"""Three-plane GEMM (crnn_gemm_f32x3) against the fp32-MFMA GEMM (crnn_gemm_f32) on the parity mode's pointwise shapes.
usage: gemm_x3_bench.py [lib ...]   (default: the product library; extra names are variant builds under scripts/_trace/)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ARGT = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
# (mode, M, N, K): mode 0 = A[M,K] B[K,N]; 1 = A[M,K] B[N,K]^T (data gradient); 2 = A[K,M]^T B[K,N] (weight gradient, K = pixels)
SHAPES = [(0, 958464, 128, 64), (0, 958464, 256, 128), (0, 239616, 256, 256), (0, 239616, 512, 256), (0, 119808, 512, 512),
          (1, 119808, 512, 512), (1, 958464, 128, 256), (1, 958464, 64, 128), (0, 13312, 1024, 256)]
libs = sys.argv[1:] or ["product"]
def load(name):
    path = os.path.join(ROOT, "crnn-ocr-lite_amd", "libcrnn_mi355x.so") if name == "product" else os.path.join(ROOT, "scripts", "_trace", name)
    L = ctypes.CDLL(path); L.crnn_gemm_f32x3.argtypes = ARGT; L.crnn_gemm_f32.argtypes = ARGT; return L
L = {n: load(n) for n in libs}
def timed(fn, args, n=5):
    for _ in range(2): assert fn(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("%-28s %10s " % ("mode M N K", "fp32 MFMA") + " ".join("%22s" % n for n in libs))
for mode, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    B = (torch.randn(K, N, device="cuda") if mode == 0 else torch.randn(N, K, device="cuda")) * 0.05
    ldb = N if mode == 0 else K
    args = [ctypes.c_int(mode), P(A), P(B), P(C), M, N, K, K, ldb, N, None, 0, 0, 0, None, ctypes.c_size_t(0), S()]
    base = timed(L[libs[0]].crnn_gemm_f32, args)
    ts = [timed(L[n].crnn_gemm_f32x3, args) for n in libs]
    fl = 2.0 * M * N * K * 6
    print("%-28s %8.1f us " % ("%d %d %d %d" % (mode, M, N, K), base) + " ".join("%9.1f us %5.1f%% mfma" % (t, 100 * fl / (t * 1e-6) / 2.5e15) for t in ts))
    del A, B, C
