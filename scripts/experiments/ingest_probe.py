#!/usr/bin/env python3
"""Per-CU ingest rate: registers vs LDS-DMA, HBM-resident (span 1 GiB) vs L2-resident (span 2 MiB per XCD-ish) sources."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "..", "_trace", "libingest.so")
L = ctypes.CDLL(so)
L.ingest_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda"); sink = torch.zeros(4, dtype=torch.int32, device="cuda")
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
grid = 256
for label, span, stride, per_wg in [("HBM stream (each WG its own 4 MiB)", 1 << 30, 4 << 20, 4 << 20), ("L2-resident (all WGs the same 1 MiB)", 1 << 20, 0, 4 << 20),
                                     ("L2-resident (16 MiB shared)", 16 << 20, 64 << 10, 4 << 20)]:
    for mode in (0, 1):
        for waves in (2, 4, 8):
            fn = lambda: L.ingest_probe(mode, waves, buf.data_ptr(), span, stride, per_wg, grid, sink.data_ptr(), S())
            for _ in range(2): assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("%-40s %-9s waves %d: %.3f ms  %.2f TB/s  %.1f GB/s per CU" % (label, "regs" if mode == 0 else "LDS-DMA", waves, ms, grid * per_wg / ms / 1e9, per_wg / ms / 1e6), flush=True)
