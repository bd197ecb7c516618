#!/usr/bin/env python3
"""Fused depthwise-stage backward of blocks 2-7 at batch B: the halo-tile kernel (crnn_dwconv3x3_bwd_fused) against the row-stream kernel
(crnn_dwconv3x3_bwd_stream), the six launches issued round-robin on their own buffers (cold like in the step), each timed with its own
events; 4 tensor passes (d, da, x in; dx out) per launch.  Optional experiment builds scripts/_trace/libdbs_*.so
(-DCRNN_DBS_EXP=1 no DMA | 2 no stores | 4 no dk fmas | 8 no dx fmas, -DCRNN_DBS_D=rows in flight)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 18, 256), (52, 9, 512), (52, 9, 512)]
L0 = native.lib()
bufs = []
for (h, w, c) in shapes:
    x = torch.randn(B, h, w, c, device="cuda").bfloat16(); d = torch.randn_like(x); da = torch.randn_like(x); dx = torch.empty_like(x)
    k = torch.randn(9, c, device="cuda"); dk = torch.empty(9, c, device="cuda")
    st = torch.cat([torch.zeros(c), torch.ones(c), torch.ones(c), torch.ones(c)]).cuda(); coef = torch.zeros(2 * c, device="cuda")
    rows = max(L0.crnn_dwconv_bwd_fused_rows(B, h, w, c), L0.crnn_dwconv_bwd_stream_rows(B, h, w, c), B * 16)
    bufs.append((d, da, st, coef, x, k, dx, dk, torch.empty(rows * 9 * c, device="cuda")))
variants = [("tile", L0, False), ("stream", L0, True)]
for pth in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libdbs_*.so"))):
    variants.append((os.path.basename(pth)[7:-3], ctypes.CDLL(pth), True))
variants.append(("stream", L0, True))      # again, last: order effects
iters = 6
for name, L, stream in variants:
    ms = np.zeros((iters, len(shapes)))
    for it in range(iters + 2):
        evs = []
        for (h, w, c), (d, da, st, coef, x, k, dx, dk, sc) in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn = L.crnn_dwconv3x3_bwd_stream if stream else L.crnn_dwconv3x3_bwd_fused
            e0.record()
            rc = fn(P(d), P(da), P(st), P(coef), P(x), P(k), P(dx), P(dk), P(sc), B, h, w, c, S())
            assert rc == 0, rc
            e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) for a, b in evs]
    med = np.median(ms, 0)
    tot_b = sum(4.0 * B * h * w * c * 2 for h, w, c in shapes)
    line = "%-8s" % name + "".join("  %dx%dx%d %.1f us (%.2f TB/s)" % (h, w, c, 1e3 * m, 4.0 * B * h * w * c * 2 / m / 1e9) for (h, w, c), m in zip(shapes, med))
    print(line + "   six launches %.3f ms = %.2f TB/s (incl. the second-stage sum)" % (med.sum(), tot_b / med.sum() / 1e9), flush=True)
