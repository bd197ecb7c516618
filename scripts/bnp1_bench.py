#!/usr/bin/env python3
"""BatchNorm-backward statistics pass (crnn_bn_bwd_ex with dx = NULL) on the step's shapes at batch 256, rotating buffers (cold), product
library and experiment builds scripts/_trace/libbnb_*.so (-DCRNN_BNB_ROWS4=1: four rows in flight, -DCRNN_BNB_MINCHUNKS=n)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
FULL = '--full' in sys.argv      # all three launches (statistics, finalize, apply) instead of the first two
L0 = native.lib()
shapes = [(104, 36, 64, 1, 1, 0.1), (104, 36, 128, 1, 1, 0.1), (104, 36, 256, 2, 2, 0.1), (52, 18, 256, 1, 1, 0.1), (52, 18, 512, 1, 2, 0.1), (52, 9, 512, 1, 1, 0.1),
          (52, 9, 512, 1, 1, 0.0), (52, 18, 256, 1, 1, 0.0)]
bufs = []
for (h, w, c, ph, pw, rate) in shapes:
    x = torch.randn(B, h, w, c, device="cuda").bfloat16(); g = torch.randn(B, h // ph, w // pw, c, device="cuda").bfloat16()
    st = torch.cat([torch.zeros(c), torch.ones(c), torch.ones(c), torch.full((c,), 0.5)]).cuda()
    bufs.append((x, g, st, torch.ones(c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(c, device="cuda"),
                 torch.empty(8192 * 2 * c, device="cuda"), torch.empty(2 * c, device="cuda"), torch.empty_like(x) if FULL else None))
variants = [("product", L0)] + [(os.path.basename(p)[7:-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libbnb_*.so")))] + [("product", L0)]
for name, L in variants:
    ms = np.zeros((6, len(shapes)))
    for it in range(8):
        evs = []
        for (h, w, c, ph, pw, rate), (x, g, st, gm, dg, db, pp, cf, dx) in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.crnn_bn_bwd_ex(P(x), P(g), P(st), P(gm), P(dx), P(dg), P(db), P(pp), P(cf), B, h, w, c, ph, pw, ctypes.c_float(rate), ctypes.c_uint64(1), ctypes.c_uint32(2), 1, S())
            assert rc == 0
            e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) for a, b in evs]
    med = np.median(ms, 0)
    print("%-10s" % name + "".join("  %dx%dx%d/%dx%d %.0f us (%.2f)" % (h, w, c, ph, pw, 1e3 * m, (B * h * w * c * 2 + B * (h // ph) * (w // pw) * c * 2) / m / 1e9)
                                   for (h, w, c, ph, pw, r), m in zip(shapes, med)) + "   total %.3f ms" % med.sum(), flush=True)
