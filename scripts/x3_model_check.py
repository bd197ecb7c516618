#!/usr/bin/env python3
"""Parity mode, one forward + backward on the same batch with the three-plane GEMMs (default) and with the fp32 MFMA (CRNN_FLAG_F32_MFMA_GEMMS):
posteriors, CTC costs and gradients side by side, for shapes the test-suite's model-level comparison does not run.
usage: x3_model_check.py [B imgh max_len gru]..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import model as M
from crnn_mi355x import native
from crnn_mi355x.engine import Engine
cases = [(48, 100, 23, 0), (64, 200, 21, 0), (40, 100, 23, 1), (256, 100, 23, 0)]
if len(sys.argv) > 1:
    v = [int(x) for x in sys.argv[1:]]; cases = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]
for B, imgh, max_len, gru in cases:
    imgw, ncls, tds, u = 32, 38, 128, 256
    cfg = M.Config(imgh=imgh, imgw=imgw, max_len=max_len, time_dense_size=tds, n_units=u, num_classes=ncls, gru=bool(gru))
    p, bn = M.init_params(cfg, seed=9, dtype=np.float64)
    p = M.randomize_params(cfg, p)
    x, lab, il, ll = M.synthetic_batch(cfg, B, seed=3, dtype=np.float64)
    out = {}
    for flags in (native.FLAG_F32_MFMA_GEMMS, 0):
        eng = Engine(B, imgh, imgw, ncls, max_len, tds, u, gru=bool(gru), stn=True, dropout=False, precision="fp32", flags=flags)
        eng.set_params(p, bn)
        eng.grads.zero_()
        y = eng.forward(x.astype(np.float32), train=True, seed=1).clone()
        loss = eng.backward(lab, il, ll, seed=1).clone()
        out[flags] = (y, loss, eng.grads.clone())
        del eng
    (y0, l0, g0), (y1, l1, g1) = out[native.FLAG_F32_MFMA_GEMMS], out[0]
    dy = float((y0 - y1).abs().max()); dl = float(((l0 - l1).abs() / l0.abs().clamp_min(1.0)).max())
    dg = float((g0.double() - g1.double()).norm() / g0.double().norm())
    print("B %3d imgh %3d %s: max |dy| %.3g, max rel dloss %.3g, gradient rel L2 %.3g, mean loss %.6f / %.6f" % (B, imgh, "GRU " if gru else "LSTM", dy, dl, dg, float(l0.mean()), float(l1.mean())))
