#!/usr/bin/env python3
"""Depthwise 3x3 forward (with BatchNorm statistics) of blocks 2-7 at batch B: the halo-tile kernel against the row-stream kernel, the six
launches issued round-robin on their own buffers (1.7 GB working set at batch 256: cold like in the step), each timed with its own
events.  Optional experiment builds scripts/_trace/libdws_*.so (-DCRNN_DWS_EXP=1 no DMA | 2 no stores | 4 no fmas)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
IMGW = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32      # image width: 32 (every BASELINE config) | 48 | 64 (round 5: row-stream kernels too)
w2, w4, w6 = IMGW + 4, (IMGW + 4) // 2, (IMGW + 4) // 4
shapes = [(104, w2, 64), (104, w2, 128), (52, w4, 256), (52, w4, 256), (52, w6, 512), (52, w6, 512)]
L0 = native.lib()
bufs = []
for (h, w, c) in shapes:
    x = torch.randn(B, h, w, c, device="cuda").bfloat16(); o = torch.empty_like(x); k = torch.randn(9, c, device="cuda")
    rows = max(L0.crnn_dwconv_num_tiles(B, h, w), L0.crnn_dwconv_fwd_stream_rows(B, h, w, c))
    bufs.append((x, o, k, torch.empty(max(rows, B * 16) * 2 * c, device="cuda")))
variants = [("tile", L0, False), ("stream", L0, True)]
for pth in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libdws_*.so"))):
    variants.append((os.path.basename(pth)[7:-3], ctypes.CDLL(pth), True))
iters = 8
ref_out = {}
for name, L, stream in variants:
    if stream:   # every build of the row-stream kernel must give the product library's bits (outputs) and statistics (to summation order)
        for si, ((h, w, c), (x, o, k, pt)) in enumerate(zip(shapes, bufs)):
            o.zero_(); pt.zero_()
            rc = L.crnn_dwconv3x3_fwd_stream(P(x), P(k), P(o), P(pt), None, B, h, w, c, 0, S())
            assert rc == 0, rc
            rows = L0.crnn_dwconv_fwd_stream_rows(B, h, w, c)
            st = pt[:rows * 2 * c].view(rows, 2, c).double().sum(0)
            if si not in ref_out:
                ref_out[si] = (o.clone(), st.clone())
            elif "exp" not in name:
                same = torch.equal(o, ref_out[si][0]); rel = float(((st - ref_out[si][1]).abs() / (ref_out[si][1].abs() + 1e-3)).max())
                if not same or rel > 1e-4:
                    print("!! %s %dx%dx%d: outputs identical %s, statistics max rel diff %.3g" % (name, h, w, c, same, rel), flush=True)

    ms = np.zeros((iters, len(shapes)))
    for it in range(iters + 2):
        evs = []
        for (h, w, c), (x, o, k, pt) in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if stream: rc = L.crnn_dwconv3x3_fwd_stream(P(x), P(k), P(o), P(pt), None, B, h, w, c, 0, S())
            else: rc = L.crnn_dwconv3x3_fwd_ex(P(x), P(k), P(o), P(pt), B, h, w, c, 0, 1, S())
            assert rc == 0, rc
            e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) for a, b in evs]
    med = np.median(ms, 0)
    tot_b = sum(2.0 * B * h * w * c * 2 for h, w, c in shapes)
    line = "%-8s" % name + "".join("  %dx%dx%d %.1f us (%.2f TB/s)" % (h, w, c, 1e3 * m, 2.0 * B * h * w * c * 2 / m / 1e9) for (h, w, c), m in zip(shapes, med))
    print(line + "   six launches %.3f ms = %.2f TB/s (image width %d, batch %d)" % (med.sum(), tot_b / med.sum() / 1e9, IMGW, B), flush=True)
