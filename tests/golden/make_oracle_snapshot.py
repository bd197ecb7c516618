#!/usr/bin/env python3
"""Snapshot of the oracle itself (tests/golden/oracle_snapshot.npz): posteriors, CTC losses, every gradient tensor's
norm plus leading entries, greedy / beam decodes and one Adam step of two tiny models (LSTM and GRU cells), from fixed
seeds.  The oracle is the checker of every GPU parity test; this file pins IT, so a later edit of oracle/ cannot move
the reference point silently (tests/test_oracle_cpu.py::test_oracle_matches_its_committed_snapshot).
Run: python tests/golden/make_oracle_snapshot.py   (no reference checkout needed; data only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ctc, model as M  # noqa: E402


def snapshot():
    out = {}
    for tag, gru in (("lstm", False), ("gru", True)):
        cfg = M.Config(imgh=40, imgw=32, max_len=6, time_dense_size=32, n_units=64, gru=gru)
        p, bn = M.init_params(cfg, seed=11, dtype=np.float64)
        p = M.randomize_params(cfg, p)
        x, lab, il, ll = M.synthetic_batch(cfg, 3, seed=4, dtype=np.float64)
        loss, lb, g, c = M.loss_and_grads(cfg, p, bn, x, lab, il, ll)
        y = c["y_pred"]
        out[tag + "/y_pred"] = y
        out[tag + "/loss_per_sample"] = np.asarray(lb)
        names = sorted(k for k in g if k in p)
        out[tag + "/grad_names"] = np.array(names)
        out[tag + "/grad_norms"] = np.array([np.linalg.norm(g[k]) for k in names])
        out[tag + "/grad_heads"] = np.stack([np.resize(g[k].ravel()[:6], 6) for k in names])
        go, gl = ctc.ctc_greedy_decode(y, np.full(3, y.shape[1]))
        bo, bl, bs = ctc.ctc_beam_decode(y, beam_width=5)
        out[tag + "/greedy"] = go; out[tag + "/greedy_len"] = gl
        out[tag + "/beam"] = bo; out[tag + "/beam_len"] = bl; out[tag + "/beam_score"] = np.asarray(bs)
        opt = M.Adam(lr=1e-3, beta_1=0.5, beta_2=0.999, epsilon=1e-7, clipnorm=5.0)
        p2 = opt.step({k: v.copy() for k, v in p.items()}, g)
        out[tag + "/adam_delta_norms"] = np.array([np.linalg.norm(p2[k] - p[k]) for k in names])
        y_inf, _ = M.forward(cfg, p, bn, x, train=False)
        out[tag + "/y_inference"] = y_inf
    return out


if __name__ == "__main__":
    snap = snapshot()
    np.savez_compressed(os.path.join(HERE, "oracle_snapshot.npz"), **snap)
    print("wrote oracle_snapshot.npz:", len(snap), "arrays")
