#!/usr/bin/env python3
"""Probe for tests/test_gpu_ops.py::test_persistent_lstm_reports_a_lost_cluster: does crnn_debug_occupy keep workgroups of the persistent LSTM
off the CUs it pins, and how long does the recurrence take / does it report give-ups?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np, torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
cus = torch.cuda.get_device_properties(0).multi_processor_count
B, T, u = 16, 22, 256; G = 4 * u
rs = np.random.RandomState(0)
U = [torch.from_numpy((rs.normal(size=(u, G)) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
ut = [x.t().contiguous().to(torch.bfloat16) for x in U]
xw = [torch.from_numpy(rs.normal(size=(T, B, G)).astype(np.float32)).cuda() for _ in range(2)]
hcat = torch.zeros(T, B, 2 * u, device="cuda"); cs = [torch.zeros(T, B, u, device="cuda") for _ in range(2)]
gt = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]
nbytes = L.crnn_lstm_persist_xbuf_bytes(T, B, u, 1)
xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u)
def lstm():
    return L.crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, 1, P(xbuf), nbytes, 0, 0, S())
assert lstm() == 0; torch.cuda.synchronize()
print("cus", cus, "clean run: status %d giveups %d" % (int(xbuf[4].item()), int(xbuf[0].item())), flush=True)
side = torch.cuda.Stream()
for blocks, lds, secs in ((cus - 2, 150 * 1024, 6), (cus, 150 * 1024, 3), (cus - 2, 100 * 1024, 6), (2 * cus, 64 * 1024, 6)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = native.hooks().crnn_debug_occupy(blocks, lds, secs * 1000 * 1000, ctypes.c_void_p(side.cuda_stream))
    time.sleep(0.3)
    a = torch.ones(4, device="cuda") + 1; a.cpu()                      # does anything run beside the spinners?
    t1 = time.perf_counter()
    g0 = int(xbuf[0].item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc2 = lstm(); e1.record(); e1.synchronize()
    t2 = time.perf_counter()
    print("occupy(%d blocks, %d KiB, %d s) rc %d | tiny op beside it after %.2f s | lstm rc %d took %.3f s (event %.1f ms) | status %d giveups +%d"
          % (blocks, lds // 1024, secs, rc, t1 - t0, rc2, t2 - t1, e0.elapsed_time(e1), int(xbuf[4].item()), int(xbuf[0].item()) - g0), flush=True)
    torch.cuda.synchronize()
    print("   spinner done after %.2f s" % (time.perf_counter() - t0), flush=True)
