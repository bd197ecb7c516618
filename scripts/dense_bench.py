#!/usr/bin/env python3
"""dense2's one-pass backward (crnn_dense_bwd_small, dense.hip) at the headline shape (M = 52 * B rows, K = 512, C = 38): the product library and the
ablation builds scripts/_trace/libdense_*.so (-DCRNN_DSB_EXP=1 no multiply-adds | 2 no LDS reads | 4 no partial store | 8 no dx store | 16 no x DMA | 32 no keep hashing | 64 no W staging), each timed over cold tensors
(the row tensors rotate over 16 copies)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, K, C = 52 * B, 512, 38
L0 = native.lib()
nb = L0.crnn_dense_bwd_small_scratch_bytes(M, K, C)
NR = 16
xs = [torch.randn(M, K, device="cuda") for _ in range(NR)]; dxs = [torch.empty(M, K, device="cuda") for _ in range(NR)]
dy = torch.randn(M, C, device="cuda") * 0.1; W = torch.randn(K, C, device="cuda"); g = torch.zeros(K * C + C, device="cuda"); scr = torch.empty(nb // 4, device="cuda")
keep = torch.zeros(M * K // 8 + 4, dtype=torch.uint8, device="cuda"); L0.crnn_dropout_keep_bytes(P(keep), M * K // 8, 0.2, 77, 9, S())
variants = [("product", L0)] + [(os.path.basename(p)[8:-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libdense_*.so")))]
for name, L in variants:
    fn = L.crnn_dense_bwd_small
    fn.argtypes = L0.crnn_dense_bwd_small.argtypes; fn.restype = ctypes.c_int
    ms = []
    for it in range(NR + 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        i = it % NR
        e0.record()
        rc = fn(P(xs[i]), P(dy), P(W), P(dxs[i]), P(g), P(g[K * C:]), P(scr), nb, M, K, C, K, K, P(keep), 0.2, 77, 9, S())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        if it >= 4: ms.append(e0.elapsed_time(e1))
    print("%-10s %.1f us (median of %d, kernel + second-stage sum, incl. ~2 us of event latency); %d MB of row tensors" % (name, 1e3 * np.median(ms), NR, 2 * M * K * 4 / 1e6 + M * C * 4 / 1e6), flush=True)
    if hasattr(L, "crnn_dense_bwd_small_set_trace"):     # timing build: per-workgroup s_memrealtime stamps (100 MHz) of the last call
        tr = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
        L.crnn_dense_bwd_small_set_trace.argtypes = [ctypes.c_void_p]; L.crnn_dense_bwd_small_set_trace(P(tr))
        fn(P(xs[5]), P(dy), P(W), P(dxs[5]), P(g), P(g[K * C:]), P(scr), nb, M, K, C, K, K, P(keep), 0.2, 77, 9, S()); torch.cuda.synchronize()
        L.crnn_dense_bwd_small_set_trace(None)
        t = tr.cpu().numpy().reshape(256, 8).astype(np.float64) / 100.0      # us
        t0 = t[:, 0].min()
        lab = ["entry", "W staged", "step 0 landed", "rows done", "final barrier", "partials stored"]
        print("  trace (us after the first workgroup's entry; median / max over workgroups): " + "  ".join("%s %.1f / %.1f" % (lab[i], np.median(t[:, i] - t0), (t[:, i] - t0).max()) for i in range(6)), flush=True)
