#!/usr/bin/env python3
"""Where a launch of the row-stream depthwise forward spends its time: per-workgroup s_memrealtime stamps (timing build scripts/_trace/libdwstrace.so,
-DCRNN_DWS_TRACE): entry, first row landed, last step done, statistics written -- against the launch's event time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 9, 512)]
trace = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
os.environ["CRNN_DWS_TRACE_PTR"] = str(trace.data_ptr())
H = native.parse_header()
L = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libdwstrace.so"))
for name, (ret, args) in H.items():
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype, fn.argtypes = ret, args
bufs = []
for (h, w, c) in shapes:
    x = torch.randn(B, h, w, c, device="cuda").bfloat16(); o = torch.empty_like(x); k = torch.randn(9, c, device="cuda")
    bufs.append((x, o, k, torch.empty(4096 * 2 * c, device="cuda")))
big = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for rep in range(3):
    for (h, w, c), (x, o, k, pt) in zip(shapes, bufs):
        big.fill_(rep)                                   # evict the caches: the launch reads cold inputs
        torch.cuda.synchronize()
        trace.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.crnn_dwconv3x3_fwd_stream(P(x), P(k), P(o), P(pt), None, B, h, w, c, 0, S())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        nwg = L.crnn_dwconv_fwd_stream_rows(B, h, w, c)
        t = trace[:nwg * 4].cpu().numpy().reshape(nwg, 4).astype(np.float64) / 100.0     # us (100 MHz counter)
        t0 = t[:, 0].min()
        if rep == 2:
            q = lambda a: "min %6.1f med %6.1f max %6.1f" % (a.min(), np.median(a), a.max())
            print("%dx%dx%d  event %.1f us  workgroups %d" % (h, w, c, 1e3 * e0.elapsed_time(e1), nwg))
            print("   entry (after the first)     " + q(t[:, 0] - t0))
            print("   entry -> first row landed   " + q(t[:, 1] - t[:, 0]))
            print("   steady state (rows)         " + q(t[:, 2] - t[:, 1]))
            print("   statistics tail             " + q(t[:, 3] - t[:, 2]))
            print("   end (after the first entry) " + q(t[:, 3] - t0) + "   span %.1f us" % (t[:, 3].max() - t0), flush=True)
