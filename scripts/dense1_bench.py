#!/usr/bin/env python3
"""dense1's forward on the stripe stream (crnn_dense_fwd_stream) at the headline shape (13312 x 4608 bf16 rows against a bf16 W^T [128][4608]) over cold
operands (four copies of x7 in rotation): the skew of the workgroups' walks over the reduction and the chunks per barrier, from an experiment build with
the knobs live (scripts/_trace/libcrnn_hooks.so: the whole library with -DCRNN_EXPERIMENT_HOOKS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, N, K, T = 52 * B, 128, 4608, 52
L0 = native.lib()
L = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libcrnn_hooks.so"))
fn = L.crnn_dense_fwd_stream; fn.argtypes = L0.crnn_dense_fwd_stream.argtypes; fn.restype = ctypes.c_int
xs = [torch.randn(M, K, device="cuda").relu().bfloat16() for _ in range(4)]
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); bias = torch.randn(N, device="cuda"); Y = torch.empty(M, N, device="cuda")
for kc in (2, 1):
    for skew in (0, 1, 2, 3, 5, 7, 11, 17, 37):
        os.environ["CRNN_NTS_KC"] = str(kc); os.environ["CRNN_NTS_SKEW"] = str(skew)
        ms = []
        for it in range(14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(P(xs[it % 4]), P(W), P(bias), P(Y), M, N, K, K, K, 1, T, 0.4, 1, 8, S()); e1.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            if it >= 2: ms.append(e0.elapsed_time(e1))
        t = float(np.median(ms))
        print("chunks per barrier %d  skew %2d   %.1f us  (%.2f TB/s of x7 rows; incl. ~2 us of event latency)" % (kc, skew, 1e3 * t, M * K * 2 / t / 1e9), flush=True)
