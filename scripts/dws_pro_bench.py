#!/usr/bin/env python3
"""Prologue form of the depthwise row-stream kernels against the two-pass path, per shape (the un-pooled block outputs feeding blocks 2, 3, 5, 7)
at batch B: forward = crnn_bn_act_pool_drop_ex + crnn_dwconv3x3_fwd_stream vs crnn_dwconv3x3_fwd_stream_pro (rate .1 and 0); backward =
crnn_dwconv3x3_bwd_stream vs crnn_dwconv3x3_bwd_stream_pro.  Optional experiment builds scripts/_trace/libdwsp_*.so / libdbsp_*.so."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 9, 512)]
L0 = native.lib()
H = native.parse_header()
def load(path):
    L = ctypes.CDLL(path)
    for name, (ret, args) in H.items():
        if hasattr(L, name):
            fn = getattr(L, name); fn.restype, fn.argtypes = ret, args
    return L
bufs = []
for (h, w, c) in shapes:
    n = B * h * w * c
    q = torch.randn(n, device="cuda").bfloat16(); x = torch.empty_like(q); d = torch.empty_like(q); da = torch.randn(n, device="cuda").bfloat16(); dx = torch.empty_like(q)
    k = torch.randn(9, c, device="cuda"); dk = torch.zeros(9, c, device="cuda")
    st2 = torch.cat([torch.randn(c), 1 + torch.rand(c), 1 + 0.5 * torch.randn(c), 1.5 + 1.5 * torch.randn(c)]).cuda()
    st1 = torch.cat([torch.randn(c) * 0.1, 1 + torch.rand(c), 1 + 0.3 * torch.randn(c), 1.0 + 0.5 * torch.randn(c)]).cuda()
    coef = (torch.randn(2 * c) * 1e-3).cuda()
    keep = torch.zeros(n // 8 + 64, dtype=torch.uint8, device="cuda"); L0.crnn_dropout_keep_bytes(P(keep), n // 8, 0.1, 7, 3, S())
    rows = max(L0.crnn_dwconv_fwd_stream_rows(B, h, w, c) * 2, L0.crnn_dwconv_bwd_stream_rows(B, h, w, c) * 9)
    bufs.append((q, x, d, da, dx, k, dk, st1, st2, coef, torch.empty(rows * c + 64, device="cuda"), keep))
def run(L, name, fn, iters=6):
    ms = np.zeros((iters, len(shapes)))
    for it in range(iters + 2):
        evs = []
        for sh, bf in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(L, sh, bf); e1.record(); evs.append((e0, e1))
            assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) for a, b in evs]
    med = np.median(ms, 0)
    print("%-34s" % name + "".join("  %dx%dx%d %6.1f us" % (h, w, c, 1e3 * m) for (h, w, c), m in zip(shapes, med)) + "   sum %.3f ms" % med.sum(), flush=True)
def f_act(L, sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
    return L.crnn_bn_act_pool_drop_ex(P(q), P(st2), P(x), B, h, w, c, 1, 1, 0.1, 7, 3, 1, 1, S())
def f_stream(L, sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
    return L.crnn_dwconv3x3_fwd_stream(P(x), P(k), P(d), P(pt), None, B, h, w, c, 0, S())
def f_pair(L, sh, bf):
    return f_act(L, sh, bf) or f_stream(L, sh, bf)
def f_pro(rate):
    def f(L, sh, bf):
        h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
        return L.crnn_dwconv3x3_fwd_stream_pro(P(q), P(st2), rate, P(keep), P(k), P(d), P(pt), B, h, w, c, S())
    return f
def b_plain(L, sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
    return L.crnn_dwconv3x3_bwd_stream(P(d), P(da), P(st1), P(coef), P(x), P(k), P(dx), P(dk), P(pt), B, h, w, c, S())
stat_parts = torch.empty(4096 * 2 * 512, device="cuda")
def b_pro(rate, stats=False):
    def f(L, sh, bf):
        h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
        return L.crnn_dwconv3x3_bwd_stream_pro(P(d), P(da), P(st1), P(coef), P(q), P(st2), rate, P(keep), P(k), P(dx), P(dk), P(pt), P(stat_parts) if stats else None,
                                               B, h, w, c, S())
    return f
def b_bn1(L, sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep = bf
    return L.crnn_bn_bwd_ex(P(q), P(dx), P(st2), P(st2), None, P(dk), P(dk), P(pt), P(coef), B, h, w, c, 1, 1, 0.1, 7, 3, 1, S())
def f_keep(L, sh, bf):
    h, w, c = sh; keep = bf[-1]
    return L.crnn_dropout_keep_bytes(P(keep), B * h * w * c // 8, 0.1, 7, 3, S())
run(L0, "keep bytes", f_keep)
run(L0, "fwd bn_act alone", f_act)
run(L0, "fwd stream alone", f_stream)
run(L0, "fwd bn_act + stream", f_pair)
run(L0, "fwd pro rate .1", f_pro(0.1))
run(L0, "fwd pro rate 0", f_pro(0.0))
run(L0, "bwd stream", b_plain)
run(L0, "bwd pro rate .1", b_pro(0.1))
run(L0, "bwd pro rate 0", b_pro(0.0))
run(L0, "bwd pro .1 + BN2 statistics", b_pro(0.1, True))
run(L0, "BN2 bwd statistics pass alone", b_bn1)
for pth in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libdwsp_*.so"))):
    L = load(pth); nm = os.path.basename(pth)[8:-3]
    run(L, "fwd pro .1 [%s]" % nm, f_pro(0.1))
    if "noxf" not in nm: run(L, "fwd pro 0  [%s]" % nm, f_pro(0.0))
for pth in sorted(glob.glob(os.path.join(ROOT, "scripts/_trace/libdbsp_*.so"))):
    L = load(pth); nm = os.path.basename(pth)[8:-3]
    run(L, "bwd pro .1 [%s]" % nm, b_pro(0.1))
