#!/usr/bin/env python3
"""Micro-benchmark of crnn_gemm_f32 on the exact GEMM shapes of one train step at batch 256 (GPU only)."""
import ctypes, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
FN = L.crnn_gemm_bf16 if '--bf16' in sys.argv else L.crnn_gemm_f32
BFS = '--bf16s' in sys.argv
WBF = '--wbf16' in sys.argv   # weights (B operand of NN/NT) stored as bf16 too
scr = torch.empty(32 * 1024 * 1024, device="cuda")
shapes = []
h, w, cin = 104, 36, 1
for i, (co, ph, pw) in enumerate([(64,1,1),(128,1,1),(256,2,2),(256,1,1),(512,1,2),(512,1,1),(512,1,1)], 1):
    M = B * h * w
    shapes += [("b%d fwd NN" % i, 0, M, co, cin), ("b%d dgrad NT" % i, 1, M, cin, co), ("b%d wgrad TN" % i, 2, cin, co, M)]
    h, w, cin = h // ph, w // pw, co
TB = 52 * B
shapes += [("dense1 fwd", 0, TB, 128, 4608), ("dense1 dgrad", 1, TB, 4608, 128), ("dense1 wgrad", 2, 4608, 128, TB),
           ("rnn1 xw", 0, TB, 1024, 128), ("rnn2 xw", 0, TB, 1024, 256), ("rnn2 dx", 1, TB, 256, 1024), ("rnn2 dW", 2, 256, 1024, TB),
           ("rnn dU", 2, 256, 1024, TB - B)]
tot = 0.0
for name, mode, M, N, K in shapes:
    if mode == 0: A = torch.randn(M, K, device="cuda"); Bm = torch.randn(K, N, device="cuda"); lda, ldb = K, N
    elif mode == 1: A = torch.randn(M, K, device="cuda"); Bm = torch.randn(N, K, device="cuda"); lda, ldb = K, K
    else: A = torch.randn(K, M, device="cuda"); Bm = torch.randn(K, N, device="cuda"); lda, ldb = M, N
    C = torch.empty(M, N, device="cuda")
    if BFS:
        A = A.to(torch.bfloat16)
        if mode == 2 or WBF: Bm = Bm.to(torch.bfloat16)
        else: C = C.to(torch.bfloat16)
    def run():
        if BFS:
            r = L.crnn_gemm_bf16_ex(mode, P(A), P(Bm), P(C), M, N, K, lda, ldb, N, None, 0, 0, 0, P(scr), 128 * 1024 * 1024, 1, int(mode == 2 or WBF), int(mode != 2), S())
        else:
            r = FN(mode, P(A), P(Bm), P(C), M, N, K, lda, ldb, N, None, 0, 0, 0, P(scr), 128 * 1024 * 1024, S())
        assert r == 0
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * M * N * K; by = 4.0 * (M * K + K * N + M * N)
    tot += ms
    print("%-14s M=%7d N=%5d K=%7d  %7.3f ms  %6.1f TF  %6.2f TB/s(min traffic)" % (name, M, N, K, ms, fl / ms / 1e9, by / ms / 1e9))
    del A, Bm, C
print("total %.3f ms" % tot)
