#!/usr/bin/env python3
"""Pointwise-conv weight gradients of the step (batch 256): pixel-streaming kernel vs the tile GEMM; time and input bytes/s."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
tot = [0.0, 0.0]
for name, M, N, K in [("b3", 958464, 256, 128), ("b4", 239616, 256, 256), ("b5", 239616, 512, 256), ("b6", 119808, 512, 512), ("b7", 119808, 512, 512)]:
    d = torch.randn(M, K, device="cuda").to(torch.bfloat16); g = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    st = torch.randn(4 * K, device="cuda").abs() + 0.5; dw = torch.empty(K, N, device="cuda")
    res = []
    fns = [L.crnn_pwconv_bnrelu6_wgrad_stream, L.crnn_pwconv_bnrelu6_wgrad]
    for path in filter(None, os.environ.get("WG_LIBS", "").split(",")):   # variant builds of gemm_wgrad.hip alone (scripts/_trace/libwg_<name>.so)
        lib = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwg_%s.so" % path))
        lib.crnn_pwconv_bnrelu6_wgrad_stream.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        fns.append(lib.crnn_pwconv_bnrelu6_wgrad_stream)
    if "--ablate" in sys.argv:
        for m in (1, 2, 3, 4, 7):
            lib = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwg_exp%d.so" % m))
            lib.crnn_pwconv_bnrelu6_wgrad_stream.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            fns.append(lib.crnn_pwconv_bnrelu6_wgrad_stream)
    for fn in fns:
        run = lambda: fn(P(d), P(st), P(g), P(dw), M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S())
        for _ in range(2): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    by = 2.0 * M * (N + K)
    tot[0] += res[0]; tot[1] += res[1]
    print("%s M=%7d N=%3d K=%3d: stream %6.1f us (%.2f TB/s)   tile %6.1f us (%.2f TB/s)  %s" % (name, M, N, K, res[0], by / res[0] / 1e6, res[1], by / res[1] / 1e6,
          "  ".join("%s %.1f" % (m, t) for m, t in zip(list(filter(None, os.environ.get("WG_LIBS", "").split(","))) + ["exp%d" % m for m in (1, 2, 3, 4, 7)], res[2:]))), flush=True)
print("sum: stream %.0f us, tile %.0f us" % tuple(tot))
