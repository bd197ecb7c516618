#!/usr/bin/env python3
"""Micro-benchmark of the depthwise 3x3 kernels on the step's shapes (batch 256)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
BF = '--bf16' in sys.argv
DT = torch.bfloat16 if BF else torch.float32
tot = 0; totb = 0
for (h, w, c) in [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 18, 256), (52, 9, 512), (52, 9, 512)]:
    x = torch.randn(B, h, w, c, device="cuda").to(DT); k = torch.randn(9, c, device="cuda"); o = torch.empty_like(x)
    nt = L.crnn_dwconv_num_tiles(B, h, w)
    parts = torch.empty(nt * 9 * c, device="cuda"); dk = torch.empty(9, c, device="cuda")
    FW = L.crnn_dwconv3x3_fwd_ex
    for name, fn in (("fwd+stats", lambda: FW(P(x), P(k), P(o), P(parts), B, h, w, c, 0, int(BF), S())),
                     ("dgrad", lambda: FW(P(x), P(k), P(o), None, B, h, w, c, 1, int(BF), S())),
                     ("wgrad", lambda: L.crnn_dwconv3x3_wgrad_ex(P(x), P(o), P(dk), P(parts), B, h, w, c, int(BF), S())),
                     ("fwd-stream", lambda: L.crnn_dwconv3x3_fwd_stream(P(x), P(k), P(o), P(parts), None, B, h, w, c, 0, S()))):
        if name == "fwd-stream" and not (BF and L.crnn_dwconv_fwd_stream_supported(B, h, w, c) == 0): continue
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        by = 2.0 * x.numel() * x.element_size()
        if name in ("fwd+stats", "dgrad"): tot += ms; totb += by
        print("%dx%dx%d %-9s %.3f ms  %.2f TB/s" % (h, w, c, name, ms, by / ms / 1e9))
print("fwd+dgrad total %.3f ms  %.2f TB/s" % (tot, totb / tot / 1e9))
