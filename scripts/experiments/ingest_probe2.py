#!/usr/bin/env python3
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "..", "_trace", "libingest2.so"))
L.ingest_probe2.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(1 << 29, dtype=torch.uint8, device="cuda"); sink = torch.zeros(4, dtype=torch.int32, device="cuda")
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rs in (512, 1024):
    total = 123 << 20 if rs >= 512 else 245 << 20
    nstripes = total // (128 * rs)
    for share in (1, 2, 4):
        for contiguous, swz in ((0, 1), (0, 3)):
            fn = lambda: L.ingest_probe2(buf.data_ptr(), rs, nstripes, share, contiguous, swz, 256, sink.data_ptr(), S())
            for _ in range(2): assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            by = nstripes * 128 * rs
            print("row stride %4d B  share %d  %-32s %.1f us  unique %.2f TB/s  ingest %.2f TB/s  %.1f GB/s per CU" % (
                rs, share, "segments" + (" + s_barrier per stage" if swz & 2 else ""), ms * 1e3, by / ms / 1e9, by * share / ms / 1e9, by * share / 256 / ms / 1e6), flush=True)
