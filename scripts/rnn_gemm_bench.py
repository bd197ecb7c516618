#!/usr/bin/env python3
"""The dense / RNN-projection GEMMs of the step (TB = 52 x 256 rows, G = 4u = 1024, fp32 tensors, bf16 MFMA) stand-alone: time,
achieved bytes/s against the minimal traffic, and the same launches issued on two streams at once."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
TB, G, u, tds, feat, C = 13312, 1024, 256, 128, 4608, 38
scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
def mk(*s): return torch.randn(*s, device="cuda")
cases = []
def case(name, mode, M, N, K, A, B, Cc, lda, ldb, ldc, bias=None, acc=0, minbytes=0):
    cases.append((name, mode, M, N, K, A, B, Cc, lda, ldb, ldc, bias, acc, minbytes))
case("xw1 (dn1.W)", 0, TB, G, tds, mk(TB, tds), mk(tds, G), mk(TB, G), tds, G, G, mk(G), 0, 4 * (TB * tds + TB * G))
case("xw2 (r1.W)", 0, TB, G, u, mk(TB, 2 * u), mk(u, G), mk(TB, G), u, G, G, mk(G), 0, 4 * (TB * u + TB * G))
case("dense2", 0, TB, C, 2 * u, mk(TB, 2 * u), mk(2 * u, C), mk(TB, C), 2 * u, C, C, mk(C), 0, 4 * (TB * 2 * u + TB * C))
case("dW1 (x^T dz)", 2, tds, G, TB, mk(TB, tds), mk(TB, G), mk(tds, G), tds, G, G, None, 0, 4 * (TB * tds + TB * G))
case("dW2 (x^T dz)", 2, u, G, TB, mk(TB, u), mk(TB, G), mk(u, G), u, G, G, None, 0, 4 * (TB * u + TB * G))
case("dU (h^T dz)", 2, u, G, TB - 256, mk(TB, u), mk(TB, G), mk(u, G), u, G, G, None, 0, 4 * (TB * u + TB * G))
case("dx1 (dz.W^T)", 1, TB, tds, G, mk(TB, G), mk(tds, G), mk(TB, tds), G, G, tds, None, 0, 4 * (TB * tds + TB * G))
case("dx2 (dz.W^T)", 1, TB, u, G, mk(TB, G), mk(u, G), mk(TB, u), G, G, u, None, 0, 4 * (TB * u + TB * G))
case("dx2 acc", 1, TB, u, G, mk(TB, G), mk(u, G), mk(TB, u), G, G, u, None, 1, 4 * (2 * TB * u + TB * G))
only = [a for a in sys.argv[1:] if not a.startswith("-")]
if only: cases = [c for c in cases if c[0].split()[0] in only]
s2 = torch.cuda.Stream()
tot = tot2 = 0
for name, mode, M, N, K, A, B, Cc, lda, ldb, ldc, bias, acc, mb in cases:
    def run(stream):
        r = L.crnn_gemm_bf16_ex(mode, P(A), P(B), P(Cc), M, N, K, lda, ldb, ldc, P(bias), 0, acc, 0, P(scratch), ctypes.c_size_t(scratch.numel()), 0, 0, 0,
                                ctypes.c_void_p(stream.cuda_stream))
        assert r == 0, r
    cur = torch.cuda.current_stream()
    for _ in range(2): run(cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): run(cur)
    e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1) / 6 * 1e3
    # the same six launches, alternating over two streams (C tensors alias: timing only)
    e0.record(); s2.wait_stream(cur)
    for i in range(3): run(cur); run(s2)
    cur.wait_stream(s2); e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 6 * 1e3
    tot += t1; tot2 += t2
    print("%-14s mode %d M=%5d N=%4d K=%5d: %6.1f us (%.2f TB/s of the minimal %.0f MB; ideal %.0f us)   two streams: %6.1f us each" % (
        name, mode, M, N, K, t1, mb / t1 / 1e6, mb / 1e6, mb / 5.5e6, t2), flush=True)
print("sum %.0f us, two-stream sum %.0f us" % (tot, tot2))
