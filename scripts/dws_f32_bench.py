#!/usr/bin/env python3
"""fp32 forms of the depthwise row-stream kernels (parity mode) per shape at batch B: forward = halo-tile kernel vs row stream vs
crnn_bn_act_pool_drop_ex + row stream vs the prologue form; backward = the three-kernel sequence vs crnn_dwconv3x3_bwd_stream_ex vs the prologue
form (+ the BatchNorm-2 backward statistics), and the statistics pass the latter replaces."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 9, 512)]
L = native.lib()
F = 0
bufs = []
for (h, w, c) in shapes:
    n = B * h * w * c
    q = torch.randn(n, device="cuda"); x = torch.empty_like(q); d = torch.empty_like(q); da = torch.randn(n, device="cuda"); dx = torch.empty_like(q); dd = torch.empty_like(q)
    k = torch.randn(9, c, device="cuda"); dk = torch.zeros(9, c, device="cuda")
    st2 = torch.cat([torch.randn(c), 1 + torch.rand(c), 1 + 0.5 * torch.randn(c), 1.5 + 1.5 * torch.randn(c)]).cuda()
    st1 = torch.cat([torch.randn(c) * 0.1, 1 + torch.rand(c), 1 + 0.3 * torch.randn(c), 1.0 + 0.5 * torch.randn(c)]).cuda()
    coef = (torch.randn(2 * c) * 1e-3).cuda()
    keep = torch.zeros(n // 8 + 64, dtype=torch.uint8, device="cuda"); L.crnn_dropout_keep_bytes(P(keep), n // 8, 0.1, 7, 3, S())
    rows = max(L.crnn_dwconv_fwd_stream_rows_ex(B, h, w, c, F) * 2, L.crnn_dwconv_bwd_stream_rows_ex(B, h, w, c, F) * 9, L.crnn_dwconv_num_tiles(B, h, w) * 9,
               L.crnn_bn_bwd_chunks(B * h * w) * 2)
    bufs.append((q, x, d, da, dx, k, dk, st1, st2, coef, torch.empty(rows * c + 64, device="cuda"), keep, dd))
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
    print("%-40s" % name + "".join("  %dx%dx%d %6.1f us" % (h, w, c, 1e3 * m) for (h, w, c), m in zip(shapes, med)) + "   sum %.3f ms" % med.sum(), flush=True)
def f_act(sh, bf):
    h, w, c = sh; q, x = bf[0], bf[1]; st2 = bf[8]
    return L.crnn_bn_act_pool_drop_ex(P(q), P(st2), P(x), B, h, w, c, 1, 1, 0.1, 7, 3, F, F, S())
def f_tile(sh, bf):
    h, w, c = sh
    return L.crnn_dwconv3x3_fwd_ex(P(bf[1]), P(bf[5]), P(bf[2]), P(bf[10]), B, h, w, c, 0, F, S())
def f_stream(sh, bf):
    h, w, c = sh
    return L.crnn_dwconv3x3_fwd_stream_dt(P(bf[1]), P(bf[5]), P(bf[2]), P(bf[10]), B, h, w, c, 0, F, S())
def f_pro(rate):
    def f(sh, bf):
        h, w, c = sh
        return L.crnn_dwconv3x3_fwd_stream_pro_ex(P(bf[0]), P(bf[8]), rate, P(bf[11]), P(bf[5]), P(bf[2]), P(bf[10]), B, h, w, c, F, S())
    return f
def b_seq(sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep, dd = bf
    return (L.crnn_bn_bwd_apply_ex(P(d), P(da), P(st1), P(coef), P(dd), B, h, w, c, 1, 1, 0.0, 0, 0, F, S()) or
            L.crnn_dwconv3x3_wgrad_ex(P(x), P(dd), P(dk), P(pt), B, h, w, c, F, S()) or
            L.crnn_dwconv3x3_fwd_ex(P(dd), P(k), P(dx), None, B, h, w, c, 1, F, S()))
def b_stream(sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep, dd = bf
    return L.crnn_dwconv3x3_bwd_stream_ex(P(d), P(da), P(st1), P(coef), P(x), P(k), P(dx), P(dk), P(pt), B, h, w, c, F, S())
stat_parts = torch.empty(4096 * 2 * 512, device="cuda")
def b_pro(rate, stats=False):
    def f(sh, bf):
        h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep, dd = bf
        return L.crnn_dwconv3x3_bwd_stream_pro_ex(P(d), P(da), P(st1), P(coef), P(q), P(st2), rate, P(keep), P(k), P(dx), P(dk), P(pt),
                                                  P(stat_parts) if stats else None, B, h, w, c, F, S())
    return f
def b_bn2(sh, bf):
    h, w, c = sh; q, x, d, da, dx, k, dk, st1, st2, coef, pt, keep, dd = bf
    return L.crnn_bn_bwd_ex(P(q), P(dx), P(st2), P(st2), None, P(dk), P(dk), P(pt), P(coef), B, h, w, c, 1, 1, 0.1, 7, 3, F, S())
run("fwd bn_act alone", f_act)
run("fwd tile alone", f_tile)
run("fwd stream alone", f_stream)
run("fwd bn_act + stream", lambda sh, bf: f_act(sh, bf) or f_stream(sh, bf))
run("fwd pro rate .1", f_pro(0.1))
run("fwd pro rate 0", f_pro(0.0))
run("bwd three-kernel sequence", b_seq)
run("bwd stream", b_stream)
run("bwd pro rate .1", b_pro(0.1))
run("bwd pro rate 0", b_pro(0.0))
run("bwd pro .1 + BN2 statistics", b_pro(0.1, True))
run("BN2 bwd statistics pass alone", b_bn2)
