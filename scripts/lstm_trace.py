#!/usr/bin/env python3
"""In-kernel timeline of the persistent LSTM forward (trace build of rnn_persist.hip, scripts/_trace/librnn_trace.so built with
-DCRNN_RNN_TRACE): per step, for two workgroups, 100-MHz timestamps at step start / tile gathered / after barrier 1 /
after MFMA + barrier 2 / after publish.  Prints the mean duration of each phase in ns."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", "librnn_trace.so"))
lib.crnn_lstm_persist_xbuf_bytes.restype = ctypes.c_size_t
B, T, u = int(os.environ.get("B", 256)), 52, 256
G = 4 * u
mt = int(os.environ.get("MT", 1)); uw = int(os.environ.get("UW", 1)) | (0x100 if os.environ.get("XCD", "0") == "1" else 0)   # XCD=1: XCD-local clusters
rs = np.random.RandomState(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
U = [torch.from_numpy((rs.normal(size=(u, G)) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
ut = [x.t().contiguous().to(torch.bfloat16) for x in U]
xw = [torch.from_numpy(rs.normal(size=(T, B, G)).astype(np.float32)).cuda() for _ in range(2)]
hcat = torch.zeros(T, B, 2 * u, device="cuda"); cs = [torch.zeros(T, B, u, device="cuda") for _ in range(2)]
gt = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]
nbytes = lib.crnn_lstm_persist_xbuf_bytes(T, B, u, 1)
xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u)
for rep in range(3):
    rc = lib.crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, 1,
                                   P(xbuf), ctypes.c_size_t(nbytes), mt, uw, S)
    assert rc == 0, rc
    torch.cuda.synchronize()
TRACE_OFF = 256 + 4096                                          # kTraceOff of rnn_exchange.h (status words + hello table)
raw = xbuf[TRACE_OFF // 4:(TRACE_OFF + 65536) // 4].cpu().numpy().view(np.uint64)
out = {"xcd_local": bool(uw & 0x100), "uw": uw & 0xff, "status": int(int(xbuf[4].item()) != -1), "giveups": int(xbuf[0].item())}
for wg in range(2):
    t = raw[wg * 128 * 8:(wg * 128 + T) * 8].reshape(T, 8)[:, :5].astype(np.int64) * 10     # ns
    s = slice(2, T - 1)
    out["wg%d" % wg] = {"poll_ns": float(np.mean(t[s, 1] - t[s, 0])), "barrier1_ns": float(np.mean(t[s, 2] - t[s, 1])),
                        "mfma_barrier2_ns": float(np.mean(t[s, 3] - t[s, 2])), "epilogue_publish_ns": float(np.mean(t[s, 4] - t[s, 3])),
                        "step_ns": float(np.mean(np.diff(t[1:, 0]))), "total_us": float((t[-1, 4] - t[0, 0]) / 1e3)}
print(json.dumps(out))
