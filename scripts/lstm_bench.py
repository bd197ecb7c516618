#!/usr/bin/env python3
"""Micro-benchmark of one Bidirectional(LSTM) layer's recurrence (forward + BPTT): T per-step launches (crnn_lstm_*_ex)
vs the persistent one-launch kernels (crnn_lstm_*_persist, 16- and 32-row batch tiles).  Prints one JSON object.

Recurrent-GEMM FLOPs per layer: forward 2 dirs x T x 2 x B x u x 4u; backward the same (dh = dz U^T).  The MFMA fraction is
those FLOPs / time / the dense peak of the multiply type (bf16 2.5 PF, fp32 157.3 TF)."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from crnn_mi355x import native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--T", type=int, default=52)
    ap.add_argument("--units", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    L = native.lib()
    B, T, u = args.batch, args.T, args.units
    G = 4 * u
    rs = np.random.RandomState(0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {"batch": B, "T": T, "units": u}
    for bf16 in (True, False):
        dt = 1 if bf16 else 0
        wdt = torch.bfloat16 if bf16 else torch.float32
        U = [torch.from_numpy((rs.normal(size=(u, G)) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
        ut = [x.t().contiguous().to(wdt) for x in U]; Ud = [x.to(wdt).contiguous() for x in U]
        xw = [torch.from_numpy(rs.normal(size=(T, B, G)).astype(np.float32)).cuda() for _ in range(2)]
        gd = torch.from_numpy(rs.normal(size=(T, B, 2 * u)).astype(np.float32)).cuda()
        hcat = torch.zeros(T, B, 2 * u, device="cuda"); cs = [torch.zeros(T, B, u, device="cuda") for _ in range(2)]
        gt = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]; dz = [torch.zeros(T, B, G, device="cuda") for _ in range(2)]
        dc = [torch.zeros(B, u, device="cuda") for _ in range(2)]
        nbytes = L.crnn_lstm_persist_xbuf_bytes(T, B, u, dt)
        xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
        hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u); gb = ctypes.c_void_p(gd.data_ptr() + 4 * u)

        def fwd(kind, mt, uw=0):
            if kind == "step":
                return L.crnn_lstm_fwd_ex(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, dt, S())
            return L.crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, dt,
                                           P(xbuf), nbytes, mt, uw, S())

        def bwd(kind, mt, uw=0):
            if kind == "step":
                return L.crnn_lstm_bwd_ex(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dc[0]), P(dc[1]),
                                          T, B, u, dt, S())
            return L.crnn_lstm_bwd_persist(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), T, B, u, dt,
                                           P(xbuf), nbytes, mt, uw, S())

        flops = 2.0 * T * 2 * B * u * G
        peak = 2500e12 if bf16 else 157.3e12
        mode = {}
        variants = (("step", 0, 0), ("persist", 1, 1), ("persist", 1, 2), ("persist", 1, 4), ("persist", 2, 2), ("persist", 2, 4), ("persist", 0, 0), ("persist", 0, 0x100), ("persist", 1, 0x102), ("persist", 0, 0))
        for kind, mt, uw in variants:
            row = {}
            xbuf[:4].zero_()            # the sticky give-up counter belongs to this variant
            for name, fn in (("fwd", fwd), ("bwd", bwd)):
                ts = []
                for it in range(args.iters + 3):
                    if it == 1 and kind == "persist" and int(xbuf[0].item()) != 0:
                        break                                   # this variant loses its hand-offs (bounded waits gave up): do not time it
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(kind, mt, uw)
                    e1.record()
                    torch.cuda.synchronize()
                    assert rc == 0, (kind, mt, uw, name, rc)
                    if it >= 3:
                        ts.append(e0.elapsed_time(e1) * 1e-3)
                t = float(np.median(ts)) if ts else float('nan')
                row[name + "_us"] = round(t * 1e6, 1)
                row[name + "_us_per_step"] = round(t * 1e6 / T, 2)
                row[name + "_mfma_frac"] = round(flops / t / peak, 4)
            row["status"] = int(int(xbuf[4].item()) != -1); row["giveups"] = int(xbuf[0].item())
            mode["%s%s" % (kind, ("_mt%d_uw%d%s" % (mt, uw & 0xff, "_xcd" if uw & 0x100 else "")) if kind == "persist" else "") + ("_again" if (kind, mt, uw) == ("persist", 0, 0) and "persist_mt0_uw0" in mode else "")] = row
        res["bf16" if bf16 else "fp32"] = mode
    print(json.dumps(res))


if __name__ == "__main__":
    main()
