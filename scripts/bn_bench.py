#!/usr/bin/env python3
"""Micro-benchmark of the BatchNorm elementwise kernels (apply+ReLU6+pool+dropout forward; two-pass backward) on the
step's shapes (batch 256), next to a plain device copy of the same tensor as the streaming yardstick."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 256
BF = '--fp32' not in sys.argv
DT = torch.bfloat16 if BF else torch.float32


RING = 6     # every call works on a different buffer set (> 256 MB MALL in total), as in the train step


def timeit(fn, n=12):
    for i in range(RING): fn(i % RING)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % RING)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tf = tb = 0.0
# the BatchNorm sites of the batch-256 step: (map, channels, pool after, dropout after)
for (h, w, c, ph, pw, rate) in [(104, 36, 64, 1, 1, 0.0), (104, 36, 128, 1, 1, 0.0), (104, 36, 256, 2, 2, 0.2), (52, 18, 256, 1, 1, 0.0),
                                (52, 18, 512, 1, 2, 0.2), (52, 9, 512, 1, 1, 0.0)]:
    xs = [torch.randn(B, h, w, c, device="cuda").to(DT) for _ in range(RING)]
    ys = [torch.empty(B, h // ph, w // pw, c, device="cuda", dtype=DT) for _ in range(RING)]
    gs = [torch.randn_like(ys[0]) for _ in range(RING)]; dxs = [torch.empty_like(xs[0]) for _ in range(RING)]
    x, y, g = xs[0], ys[0], gs[0]
    st = torch.cat([torch.zeros(c), torch.ones(c), torch.ones(c), torch.full((c,), 0.5)]).cuda()
    gamma = torch.ones(c, device="cuda"); dg = torch.empty(c, device="cuda"); db = torch.empty(c, device="cuda")
    pp = torch.empty(L.crnn_bn_bwd_chunks(B * h * w) * 2 * c, device="cuda"); coef = torch.empty(2 * c, device="cuda")
    es = x.element_size()
    t_copy = timeit(lambda i: dxs[i].copy_(xs[i]))
    t_f = timeit(lambda i: L.crnn_bn_act_pool_drop_ex(P(xs[i]), P(st), P(ys[i]), B, h, w, c, ph, pw, rate, 1, 2, int(BF), int(BF), S()))
    t_b = timeit(lambda i: L.crnn_bn_bwd_ex(P(xs[i]), P(gs[i]), P(st), P(gamma), P(dxs[i]), P(dg), P(db), P(pp), P(coef), B, h, w, c, ph, pw, rate, 1, 2, int(BF), S()))
    by_f = (x.numel() + y.numel()) * es
    by_b = (3 * x.numel() + 2 * g.numel()) * es
    tf += t_f; tb += t_b
    print("%dx%dx%d pool %dx%d  copy %.1f us %.2f TB/s | fwd %.1f us %.2f TB/s | bwd(3 launches) %.1f us %.2f TB/s" % (
        h, w, c, ph, pw, t_copy * 1e3, 2 * x.numel() * es / t_copy / 1e9, t_f * 1e3, by_f / t_f / 1e9, t_b * 1e3, by_b / t_b / 1e9))
print("total fwd %.3f ms  bwd %.3f ms" % (tf, tb))
