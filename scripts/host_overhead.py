#!/usr/bin/env python3
"""Host enqueue cost vs GPU time of the train step at several batch sizes: how long the host needs to enqueue one step
(perf_counter around the call, no synchronisation) next to the synchronised step time."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np, torch
from bench import synthetic_batch
from crnn_mi355x.engine import Engine
from crnn_mi355x.init import initial_parameters
from crnn_mi355x.optimizers import Adam
out = {}
for B in (256, 64, 16):
    eng = Engine(B, dropout=True, precision="bf16s")
    eng.set_params(initial_parameters(eng.layout, 256, False, seed=1))
    x, lab, il, ll = synthetic_batch(B, 0)
    xd = torch.from_numpy(x).cuda(); labd = torch.from_numpy(lab.astype(np.int32)).cuda()
    ild = torch.from_numpy(il.astype(np.int32)).cuda(); lld = torch.from_numpy(ll.astype(np.int32)).cuda()
    opt = Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5)
    for it in range(5): eng.train_step(xd, labd, ild, lld, opt, it)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for it in range(n): eng.train_step(xd, labd, ild, lld, opt, 5 + it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # pieces of the host path
    torch.cuda.synchronize(); a = time.perf_counter()
    for it in range(10): eng.forward(xd, train=True, seed=it)
    b = time.perf_counter(); torch.cuda.synchronize()
    for it in range(10): eng.backward(labd, ild, lld, seed=it)
    c = time.perf_counter(); torch.cuda.synchronize()
    out[B] = {"host_enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 3), "synced_ms_per_step": round((t2 - t0) / n * 1e3, 3),
              "host_forward_ms": round((b - a) / 10 * 1e3, 3), "host_backward_ms": round((c - b) / 10 * 1e3, 3)}
    del eng
    torch.cuda.empty_cache()
print(json.dumps(out))
