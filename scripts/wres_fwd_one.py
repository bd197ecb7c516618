#!/usr/bin/env python3
"""One shape of the forward weights-resident kernel (product library) for counter passes: wres_fwd_one.py M N K"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
M, N, K = (int(v) for v in sys.argv[1:4])
P = lambda t: ctypes.c_void_p(t.data_ptr())
X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); st = torch.randn(4 * K, device="cuda").abs() + 0.5
rows = L.crnn_pwconv_fwd_wres_rows(M, N, K); parts = torch.empty(rows * 2 * N, device="cuda")
for _ in range(4):
    assert L.crnn_pwconv_bnrelu6_fwd_wres(P(X), P(st), P(W), P(Y), M, N, K, P(parts), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
