#!/usr/bin/env python3
"""crnn_ctc_loss_grad at the benchmark shape (B x 52 x 38, labels up to 23): product library and the phase-ablation builds scripts/_trace/libctc_exp<n>.so
(-DCRNN_CTC_EXP=n: 3 stop before the log-softmax phase, 1 after it, 2 after the two recursions).  usage: ctc_bench.py [B]"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
from bench import synthetic_batch
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, C, L = 52, 38, 23
x, lab, il, ll = synthetic_batch(B, seed=0, T=T)
y = torch.softmax(torch.randn(B, T, C, device="cuda") * 2, -1).contiguous()
labd = torch.from_numpy(lab.astype(np.int32)).cuda(); ild = torch.from_numpy(il.astype(np.int32)).cuda(); lld = torch.from_numpy(ll.astype(np.int32)).cuda()
loss = torch.zeros(B, device="cuda"); dl = torch.zeros(T, B, C, device="cuda")
libs = [("product", native.lib())] + [(os.path.basename(p)[6:-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libctc_*.so")))]
for name, lib in libs:
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.crnn_ctc_loss_grad(P(y), P(labd), P(ild), P(lld), P(loss), P(dl), B, T, C, L, 2, ctypes.c_float(1.0 / B), S())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0, rc
        if it >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    print("%-10s %6.1f us (median of 10, incl. ~2 us of event latency), batch %d" % (name, float(np.median(ts)), B), flush=True)
