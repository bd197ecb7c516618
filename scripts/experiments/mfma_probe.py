#!/usr/bin/env python3
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "..", "_trace", "libmfma.so"))
out = torch.zeros(256 * 8 * 2, dtype=torch.int64, device="cuda"); sink = torch.zeros(4, device="cuda")
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for threads in (256, 384, 512):
    for n in (120, 1200, 12000):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); assert L.mfma_probe(256, threads, n, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sink.data_ptr()), S()) == 0; e1.record(); torch.cuda.synchronize()
        o = out.cpu().numpy().reshape(-1, 2)[: 256 * threads // 64]
        cyc, ticks = o[:, 0].mean(), o[:, 1].mean()
        print("threads %d  n=%6d (%7d MFMA/wave): %.1f us  s_memtime %.1f cycles/MFMA  realtime %.2f ns/MFMA  -> shader clock %.2f GHz  (PF/s %.2f)" % (
            threads, n, 16 * n, e0.elapsed_time(e1) * 1e3, cyc / (16 * n), ticks * 10 / (16 * n), cyc / (ticks * 10), 256 * threads / 64 * 16 * n * 32768 / (e0.elapsed_time(e1) * 1e-3) / 1e15))
