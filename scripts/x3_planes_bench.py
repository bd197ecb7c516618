#!/usr/bin/env python3
"""Three-plane pointwise GEMMs of the parity mode per conv-stack block at batch B: operands split per tile (fp32 in) vs handed over as bf16 planes
(crnn_split3_planes): forward (weights), data gradient + BatchNorm-backward statistics (weights; weights and the incoming gradient)."""
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
for (px, ci, co) in shapes:
    M = B * px
    d = torch.randn(M, ci, device="cuda"); q = torch.empty(M, co, device="cuda"); dq = torch.randn(M, co, device="cuda"); da = torch.empty(M, ci, device="cuda")
    w = torch.randn(ci, co, device="cuda") * 0.1
    st = torch.cat([torch.randn(ci) * 0.1, 1 + torch.rand(ci), 1 + 0.3 * torch.randn(ci), 1.0 + 0.5 * torch.randn(ci)]).cuda()
    parts = torch.empty(max(L.crnn_pwconv_stat_rows(M) * 2 * co, L.crnn_gemm_f32x3_bnstats_rows(M) * 2 * ci) + 64, device="cuda")
    wpl = torch.empty(3 * ci * co, dtype=torch.int16, device="cuda"); qpl = torch.empty(3 * M * co, dtype=torch.int16, device="cuda")
    assert L.crnn_split3_planes(P(w), P(wpl), ci * co, ci * co, S()) == 0 and L.crnn_split3_planes(P(dq), P(qpl), M * co, M * co, S()) == 0
    bufs.append((M, d, q, dq, da, w, st, parts, wpl, qpl))
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
    print("%-44s" % name + "".join("  %d>%d %6.1f us" % (ci, co, 1e3 * m) for (px, ci, co), m in zip(shapes, med)) + "   sum %.3f ms" % med.sum(), flush=True)
def fwd(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_pwconv_bnrelu6_fwd_f32x3(P(d), P(st), P(w), P(q), M, co, ci, P(parts), S())
def fwd_pl(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_pwconv_bnrelu6_fwd_f32x3_pl(P(d), P(st), P(w), P(wpl), ci * co, P(q), M, co, ci, P(parts), S())
def dg(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_gemm_f32x3_bnstats(P(dq), P(w), P(da), M, ci, co, P(d), P(st), P(parts), S())
def dg_w(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_gemm_f32x3_bnstats_pl(P(dq), None, 0, P(w), P(wpl), ci * co, P(da), M, ci, co, P(d), P(st), P(parts), S())
def dg_wq(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_gemm_f32x3_bnstats_pl(P(dq), P(qpl), M * co, P(w), P(wpl), ci * co, P(da), M, ci, co, P(d), P(st), P(parts), S())
def split_q(sh, bf):
    px, ci, co = sh; M, d, q, dq, da, w, st, parts, wpl, qpl = bf
    return L.crnn_split3_planes(P(dq), P(qpl), M * co, M * co, S())
for rep in range(2):
    run("forward, weights split per tile", fwd)
    run("forward, weight planes", fwd_pl)
    run("data gradient, split per tile", dg)
    run("data gradient, weight planes", dg_w)
    run("data gradient, weight + gradient planes", dg_wq)
run("crnn_split3_planes of the gradient", split_q)
