#!/usr/bin/env python3
"""Pointwise GEMMs of the parity mode per conv-stack block at batch B: the tile kernel (gemm_x3p_kernel) against the weights-resident plane kernels
(gemm_wres3.hip) -- forward (BatchNorm-1 + ReLU6 while staging, BatchNorm-2 statistics) with three planes, data gradient (+ BatchNorm-1 backward
statistics) with two and three planes.  Time per launch (median of 6, back to back over the blocks: each launch's operands are cold), the bound
(six / three bf16 products per MAC at 2.5 PFLOP/s; fp32 traffic at 8 TB/s) and the largest deviation between the two kernels' results."""
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
# W3_LIB: a variant build of gemm_wres3.hip alone (scripts/_trace/libw3_<name>.so) for the resident calls
R = L
if os.environ.get('W3_LIB'):
    R = ctypes.CDLL(os.environ['W3_LIB'])
    for n in ('crnn_pwconv_bnrelu6_fwd_wres3', 'crnn_gemm_wres3_bnstats', 'crnn_gemm_wres3_stat_rows', 'crnn_pwconv_bnrelu6_wgrad_planes_stream'):
        if not hasattr(R, n): setattr(R, n, getattr(L, n)); continue
        getattr(R, n).argtypes = getattr(L, n).argtypes; getattr(R, n).restype = getattr(L, n).restype
bufs = []
for (px, ci, co) in shapes:
    M = B * px
    d = torch.randn(M, ci, device="cuda"); q = torch.empty(M, co, device="cuda"); dq = torch.randn(M, co, device="cuda"); da = torch.empty(M, ci, device="cuda")
    w = torch.randn(ci, co, device="cuda") * 0.1
    st = torch.cat([torch.randn(ci) * 0.1, 1 + torch.rand(ci), 1 + 0.3 * torch.randn(ci), 1.0 + 0.5 * torch.randn(ci)]).cuda()
    rows = max(L.crnn_pwconv_stat_rows(M), L.crnn_gemm_f32x3_bnstats_rows(M), 1024)
    parts = torch.empty(rows * 2 * max(ci, co) + 64, device="cuda")
    dw = torch.empty(ci, co, device="cuda")
    bufs.append((M, d, q, dq, da, w, st, parts, dw))
def run(name, fn, iters=6):
    ms = np.full((iters, len(shapes)), np.nan)
    for it in range(iters + 2):
        evs = []
        for sh, bf in zip(shapes, bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(sh, bf); e1.record(); evs.append((e0, e1, rc))
        torch.cuda.synchronize()
        if it >= 2: ms[it - 2] = [a.elapsed_time(b) if rc == 0 else np.nan for a, b, rc in evs]
    med = np.median(ms, 0)
    print("%-40s" % name + "".join("  %d>%d %6.1f us" % (ci, co, 1e3 * m) for (px, ci, co), m in zip(shapes, med)) + "   sum %.3f ms" % np.nansum(med), flush=True)
    return med
def mk(kind, planes, res):
    def f(sh, bf):
        px, ci, co = sh; M, d, q, dq, da, w, st, parts, dw = bf
        if kind == "wgrad":
            if res: return R.crnn_pwconv_bnrelu6_wgrad_planes_stream(P(d), P(st), P(dq), P(dw), M, co, ci, P(scr), sb, S()) if planes == 2 else -3
            return (L.crnn_pwconv_bnrelu6_wgrad_f32x3 if planes == 3 else L.crnn_pwconv_bnrelu6_wgrad_f32x2)(P(d), P(st), P(dq), P(dw), M, co, ci, P(scr), sb, S())
        if kind == "fwd":
            if res: return R.crnn_pwconv_bnrelu6_fwd_wres3(P(d), P(st), P(w), P(q), M, co, ci, planes, P(parts), S())
            return (L.crnn_pwconv_bnrelu6_fwd_f32x3 if planes == 3 else L.crnn_pwconv_bnrelu6_fwd_f32x2)(P(d), P(st), P(w), P(q), M, co, ci, P(parts), S())
        if res: return R.crnn_gemm_wres3_bnstats(P(dq), P(w), P(da), M, ci, co, planes, P(d), P(st), P(parts), S())
        return (L.crnn_gemm_f32x3_bnstats if planes == 3 else L.crnn_gemm_f32x2_bnstats)(P(dq), P(w), P(da), M, ci, co, P(d), P(st), P(parts), S())
    return f
print("batch %d; bounds per block (us): " % B + "  ".join("%d>%d mfma6 %.0f mfma3 %.0f fwd-hbm %.0f dgrad-hbm %.0f" % (
    ci, co, 2e6 * B * px * ci * co * 6 / 2.5e15, 2e6 * B * px * ci * co * 3 / 2.5e15, 1e6 * B * px * (ci + co) * 4 / 8e12, 1e6 * B * px * (2 * ci + co) * 4 / 8e12) for px, ci, co in shapes))
outs = {"fwd": 2, "dgrad": 4, "wgrad": 8}
scr = torch.empty(16 * 1024 * 1024, device="cuda"); sb = ctypes.c_size_t(scr.numel() * 4)
kinds = (("fwd", 3), ("fwd", 2), ("dgrad", 2), ("dgrad", 3), ("wgrad", 2))
if os.environ.get("W3_ONLY"): kinds = tuple((k.split(":")[0], int(k.split(":")[1])) for k in os.environ["W3_ONLY"].split(","))
for kind, planes in kinds:
    t0 = run("%s, %d planes, tile kernel" % (kind, planes), mk(kind, planes, False)) if not os.environ.get("W3_LIB") else np.ones(len(shapes))
    ref = [bf[outs[kind]].clone() for bf in bufs]
    t1 = run("%s, %d planes, weights resident" % (kind, planes), mk(kind, planes, True))
    dev = []
    for r, bf, m in zip(ref, bufs, t1):
        o = bf[outs[kind]]
        dev.append(float("nan") if np.isnan(m) else float((o.double() - r.double()).abs().max() / r.double().abs().max()))
    print("   resident / tile time %.3f (supported shapes); max |resident - tile| / max |tile| per shape: %s" % (
        np.nansum(t1) / np.nansum(np.where(np.isnan(t1), np.nan, t0)), " ".join("%.2e" % v for v in dev)), flush=True)
