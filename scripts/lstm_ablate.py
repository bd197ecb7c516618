#!/usr/bin/env python3
"""Persistent LSTM recurrence of one Bidirectional layer (B = 256, T = 52, u = 256; the step's variant: 16-row tiles, two unit groups per workgroup, XCD-local
clusters), forward and BPTT, bf16 and fp32 recurrent weights: the product library against variant builds of rnn_persist.hip alone
(RNN_LIBS=name,name -> scripts/_trace/librnn_<name>.so).  Median of 20 launches, us."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
L = native.lib()
libs = [("product", L)]
for n in filter(None, os.environ.get("RNN_LIBS", "").split(",")):
    lib = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/librnn_%s.so" % n))
    for f in ("crnn_lstm_fwd_persist", "crnn_lstm_bwd_persist_db", "crnn_lstm_persist_xbuf_bytes"):
        getattr(lib, f).argtypes = getattr(L, f).argtypes; getattr(lib, f).restype = getattr(L, f).restype
    libs.append((n, lib))
B, T, u = int(os.environ.get("B", 256)), int(os.environ.get("T", 52)), 256
G = 4 * u
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rs = np.random.RandomState(0)
for bf16 in (True, False):
    dt = 1 if bf16 else 0
    wdt = torch.bfloat16 if bf16 else torch.float32
    U = [torch.from_numpy((rs.normal(size=(u, G)) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    ut = [x.t().contiguous().to(wdt) for x in U]; Ud = [x.to(wdt).contiguous() for x in U]
    xw = [torch.from_numpy(rs.normal(size=(T, B, G)).astype(np.float32)).cuda() for _ in range(2)]
    gd = torch.from_numpy(rs.normal(size=(T, B, 2 * u)).astype(np.float32)).cuda()
    hcat = torch.zeros(T, B, 2 * u, device="cuda"); cs = [torch.zeros(T, B, u, device="cuda") for _ in range(2)]
    gt = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]; dz = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]
    dbp = [torch.zeros(L.crnn_rnn_db_rows(B) * G, device="cuda") for _ in range(2)]
    nbytes = L.crnn_lstm_persist_xbuf_bytes(T, B, u, dt)
    xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u); gb = ctypes.c_void_p(gd.data_ptr() + 4 * u)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for name, lib in libs:
        fwd = lambda: lib.crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, dt,
                                                P(xbuf), nbytes, 1, 0x102, S())
        bwd = lambda: lib.crnn_lstm_bwd_persist_db(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dbp[0]), P(dbp[1]),
                                                   T, B, u, dt, P(xbuf), nbytes, 1, 0x102, S())
        out = []
        for cold in (False, True):
            for fn in (fwd, bwd):
                ts = []
                for it in range(23):
                    if cold: flush.fill_(it)                     # operands out of the caches, as after the step's other kernels
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
                    assert rc == 0, rc
                    if it >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
                out.append(float(np.median(ts)))
        print("%-5s %-10s fwd %6.1f us  bwd %6.1f us | after a 512 MiB fill: fwd %6.1f  bwd %6.1f   (give-ups %d)" % ("bf16" if bf16 else "fp32", name, out[0], out[1], out[2], out[3], int(xbuf[0].item())), flush=True)
