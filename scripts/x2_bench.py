#!/usr/bin/env python3
"""Pointwise GEMMs of the parity mode per conv-stack block at batch B: three planes per operand (six products, fp32-accurate) vs two planes (three
products, 16 significant bits per factor) -- time per launch and the deviation of the two-plane result from the three-plane one."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
shapes = [(104 * 36, 64, 128), (104 * 36, 128, 256), (52 * 18, 256, 256), (52 * 18, 256, 512), (52 * 9, 512, 512)]   # (pixels, ci, co) of blocks 2..6 (7 = 6)
L = native.lib()
bufs = []
scr = torch.empty(16 * 1024 * 1024, device="cuda"); sb = ctypes.c_size_t(scr.numel() * 4)
for (px, ci, co) in shapes:
    M = B * px
    d = torch.randn(M, ci, device="cuda"); q = torch.empty(M, co, device="cuda"); dq = torch.randn(M, co, device="cuda"); da = torch.empty(M, ci, device="cuda")
    w = torch.randn(ci, co, device="cuda") * 0.1; dw = torch.empty(ci, co, device="cuda")
    st = torch.cat([torch.randn(ci) * 0.1, 1 + torch.rand(ci), 1 + 0.3 * torch.randn(ci), 1.0 + 0.5 * torch.randn(ci)]).cuda()
    parts = torch.empty(max(L.crnn_pwconv_stat_rows(M) * 2 * co, L.crnn_gemm_f32x3_bnstats_rows(M) * 2 * ci) + 64, device="cuda")
    bufs.append((M, d, q, dq, da, w, st, parts, dw))
def run(name, fn, iters=6):
    ms = np.zeros((iters, len(shapes)))
    for it in range(iters + 2):
        evs = []
        for sh, bf in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(sh, bf); e1.record(); evs.append((e0, e1))
            assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) for a, b in evs]
    med = np.median(ms, 0)
    print("%-34s" % name + "".join("  %d>%d %6.1f us" % (ci, co, 1e3 * m) for (px, ci, co), m in zip(shapes, med)) + "   sum %.3f ms" % med.sum(), flush=True)
    return med
def mk(kind, x):
    def f(sh, bf):
        px, ci, co = sh; M, d, q, dq, da, w, st, parts, dw = bf
        if kind == "fwd":
            fn = L.crnn_pwconv_bnrelu6_fwd_f32x3 if x == 3 else L.crnn_pwconv_bnrelu6_fwd_f32x2
            return fn(P(d), P(st), P(w), P(q), M, co, ci, P(parts), S())
        if kind == "dgrad":
            fn = L.crnn_gemm_f32x3_bnstats if x == 3 else L.crnn_gemm_f32x2_bnstats
            return fn(P(dq), P(w), P(da), M, ci, co, P(d), P(st), P(parts), S())
        fn = L.crnn_pwconv_bnrelu6_wgrad_f32x3 if x == 3 else L.crnn_pwconv_bnrelu6_wgrad_f32x2
        return fn(P(d), P(st), P(dq), P(dw), M, co, ci, P(scr), sb, S())
    return f
outs = {"fwd": 2, "dgrad": 4, "wgrad": 8}
for kind in ("fwd", "dgrad", "wgrad"):
    t3 = run("%s, three planes" % kind, mk(kind, 3))
    ref = [bf[outs[kind]].clone() for bf in bufs]
    t2 = run("%s, two planes" % kind, mk(kind, 2))
    dev = []
    for r, bf in zip(ref, bufs):
        o = bf[outs[kind]]
        dev.append(float((o.double() - r.double()).abs().max() / r.double().abs().max()))
    print("   two-plane / three-plane time %.3f; max |x2 - x3| / max |x3| per shape: %s" % (t2.sum() / t3.sum(), " ".join("%.2e" % v for v in dev)), flush=True)
