#!/usr/bin/env python3
"""Micro-benchmark of the fused depthwise-stage backward on the step's shapes (batch 256), optionally over experiment builds
(scripts/_trace/libfused_*.so: -DFUSED_WPE=waves/SIMD -DFUSED_ABL=1 no global loads | 2 fill only | 3 no weight gradient)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
libs = [("product", native.lib())]
if "--product-only" not in sys.argv:
    libs += [(os.path.basename(p)[8:-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libfused_*.so")))]
shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 9, 512)]
for name, L in libs:
    line = "%-10s" % name; tot = 0
    for (h, w, c) in shapes:
        x = torch.randn(B, h, w, c, device="cuda").bfloat16(); d = torch.randn_like(x); da = torch.randn_like(x); dx = torch.empty_like(x)
        k = torch.randn(9, c, device="cuda"); dk = torch.empty(9, c, device="cuda")
        st = torch.cat([torch.zeros(c), torch.ones(c), torch.ones(c), torch.ones(c)]).cuda(); coef = torch.zeros(2 * c, device="cuda")
        L.crnn_dwconv_bwd_fused_rows.restype = ctypes.c_int
        rows = L.crnn_dwconv_bwd_fused_rows(B, h, w, c)
        parts = torch.empty(rows * 9 * c, device="cuda")
        fn = lambda: L.crnn_dwconv3x3_bwd_fused(P(d), P(da), P(st), P(coef), P(x), P(k), P(dx), P(dk), P(parts), B, h, w, c, S())
        for _ in range(2): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5; tot += ms * (2 if c != 64 and c != 128 else 1)
        line += "  %dx%dx%d %.3f ms (%.2f TB/s)" % (h, w, c, ms, 4.0 * x.numel() * 2 / ms / 1e9)
    print(line + "   step total %.3f" % tot, flush=True)
