#!/usr/bin/env python3
"""Parity mode, data gradient of the pointwise convs from pre-split planes (gemm_pres.hip) against the tile kernel (crnn_gemm_f32x2/3_bnstats) and the
weights-resident fp32-operand kernel (gemm_wres3.hip): time per launch at batch B (median of 6, back to back over the blocks: operands cold), the result's
deviation from the tile kernel's, the statistics' deviation (column sums over the partial rows, float64), and the planes of `a` against crnn_split3_planes of
the activated tensor (bit for bit).  PRES_LIB: a variant build of gemm_pres.hip alone for the plane calls."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np
import torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
planes_list = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
shapes = [(104 * 36, 128, 256), (52 * 18, 256, 256), (52 * 18, 256, 512), (52 * 9, 512, 512)]   # (pixels, ci, co) of blocks 3..6 (7 = 6): da [M][ci] = dq [M][co] . W^T
L = native.lib()
R = L
if os.environ.get("PRES_LIB"):
    R = ctypes.CDLL(os.environ["PRES_LIB"])
    for n in ("crnn_gemm_pres_supported", "crnn_gemm_pres_stat_rows", "crnn_gemm_pres_bnstats"):
        getattr(R, n).argtypes = getattr(L, n).argtypes; getattr(R, n).restype = getattr(L, n).restype
torch.manual_seed(1)
bufs = []
for (px, ci, co) in shapes:
    M = B * px
    d = torch.randn(M, ci, device="cuda") * 1.5 + 0.4
    dq = torch.randn(M, co, device="cuda") * 1e-3
    w = torch.randn(ci, co, device="cuda") * 0.05
    st = torch.cat([torch.randn(ci) * 0.1, 1 + torch.rand(ci), 1 + 0.3 * torch.randn(ci), 1.0 + 0.5 * torch.randn(ci)]).cuda()
    pl = torch.empty(3 * M * co, dtype=torch.int16, device="cuda")
    assert L.crnn_split3_planes(P(dq), P(pl), M * co, M * co, S()) == 0
    rows = max(L.crnn_gemm_f32x3_bnstats_rows(M), 2048)
    bufs.append(dict(M=M, d=d, dq=dq, w=w, st=st, pl=pl, da=torch.empty(M, ci, device="cuda"), da2=torch.empty(M, ci, device="cuda"),
                     parts=torch.zeros(rows * 2 * ci + 64, device="cuda"), parts2=torch.zeros(rows * 2 * ci + 64, device="cuda"),
                     ap=torch.zeros(3 * M * ci, dtype=torch.int16, device="cuda")))
torch.cuda.synchronize()

def run(name, fn, iters=6):
    ms = np.full((iters, len(shapes)), np.nan)
    for it in range(iters + 2):
        for i, (sh, bf) in enumerate(zip(shapes, bufs)):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(sh, bf); e1.record(); torch.cuda.synchronize()
            if rc == 0 and it >= 2: ms[it - 2, i] = e0.elapsed_time(e1)
            elif rc not in (0, -3): raise SystemExit("%s: rc %d at %s" % (name, rc, sh))
    med = np.median(ms, axis=0)
    print("%-46s" % name + "  ".join("%3d<%-3d %6.1f us" % (sh[1], sh[2], m * 1e3) for sh, m in zip(shapes, med)) + "   sum %.3f ms" % np.nansum(med), flush=True)
    return med

print("batch %d; bounds per block (us): " % B + "  ".join("%d<%d mfma3 %.0f mfma6 %.0f hbm %.0f" % (
    ci, co, 2e6 * B * px * ci * co * 3 / 2.5e15, 2e6 * B * px * ci * co * 6 / 2.5e15, 1e6 * B * px * (2 * ci + co) * 4 / 8e12) for px, ci, co in shapes))
for planes in planes_list:
    tile = L.crnn_gemm_f32x2_bnstats if planes == 2 else L.crnn_gemm_f32x3_bnstats
    t0 = run("dgrad, %d planes, tile kernel (fp32 dq)" % planes,
             lambda sh, bf: tile(P(bf["dq"]), P(bf["w"]), P(bf["da"]), bf["M"], sh[1], sh[2], P(bf["d"]), P(bf["st"]), P(bf["parts"]), S()))
    tw = run("dgrad, %d planes, weights resident (fp32 dq)" % planes,
             lambda sh, bf: L.crnn_gemm_wres3_bnstats(P(bf["dq"]), P(bf["w"]), P(bf["da2"]), bf["M"], sh[1], sh[2], planes, P(bf["d"]), P(bf["st"]), P(bf["parts2"]), S()))
    for emit in (0, planes):
        def f(sh, bf):
            M = bf["M"]
            return R.crnn_gemm_pres_bnstats(P(bf["pl"]), M * sh[2], P(bf["w"]), P(bf["da2"]), M, sh[1], sh[2], planes, P(bf["d"]), P(bf["st"]), P(bf["parts2"]),
                                            P(bf["ap"]) if emit else None, M * sh[1], emit, S())
        for bf in bufs: bf["da2"].zero_(); bf["parts2"].zero_()
        t1 = run("dgrad, %d planes, from planes%s" % (planes, ", writes %d planes of a" % emit if emit else ""), f)
        if os.environ.get("PRES_TRACE"):   # a PRES_EXP & 128 build left [workgroup][cycles, 100 MHz ticks, stripes] of its main loop in da2
            for sh, bf in zip(shapes, bufs):
                g = min(2048, bf["da2"].numel() // 8)
                tr = bf["da2"].view(-1).view(torch.int64)[:4 * g].view(g, 4)[:, :4].cpu().numpy().astype(np.float64)
                tr = tr[(tr[:, 2] > 0) & (tr[:, 2] < 1e6) & (tr[:, 1] > 0) & (tr[:, 1] < 1e9)][:512]
                print("      %d<%d: %d workgroups, stripes %.1f, prologue %.1f us, main loop %.1f us, %.0f cycles per stripe, clock %.2f GHz" % (
                    sh[1], sh[2], len(tr), tr[:, 2].mean(), tr[:, 3].mean() / 100, tr[:, 1].mean() / 100, (tr[:, 0] / tr[:, 2]).mean(), (tr[:, 0] / tr[:, 1]).mean() / 10))
        dev, sdev, adev = [], [], []
        for sh, bf, m in zip(shapes, bufs, t1):
            if np.isnan(m): dev.append(float("nan")); sdev.append(float("nan")); adev.append(float("nan")); continue
            M, ci = bf["M"], sh[1]
            dev.append(float((bf["da2"].double() - bf["da"].double()).abs().max() / bf["da"].double().abs().max()))
            r0 = L.crnn_gemm_f32x3_bnstats_rows(M); r1 = R.crnn_gemm_pres_stat_rows(M, ci, sh[2], planes)
            s0 = bf["parts"][:r0 * 2 * ci].double().view(r0, 2, ci).sum(0); s1 = bf["parts2"][:r1 * 2 * ci].double().view(r1, 2, ci).sum(0)
            sdev.append(float(((s1 - s0).abs().max(1).values / s0.abs().max(1).values).max()))
            if emit:
                ref = torch.empty(3 * M * ci, dtype=torch.int16, device="cuda")
                av = torch.empty(M, ci, device="cuda")
                assert L.crnn_bn_act_pool_drop_ex(P(bf["d"]), P(bf["st"]), P(av), 1, 1, M, ci, 1, 1, 0.0, 0, 0, 0, 0, S()) == 0
                assert L.crnn_split3_planes(P(av), P(ref), M * ci, M * ci, S()) == 0
                adev.append(int((ref[:emit * M * ci] != bf["ap"][:emit * M * ci]).sum()))
        print("   planes / tile time %.3f, planes / resident %.3f; max |planes - tile| / max |tile|: %s; statistics: %s%s" % (
            np.nansum(t1) / np.nansum(np.where(np.isnan(t1), np.nan, t0)), np.nansum(t1) / np.nansum(np.where(np.isnan(t1), np.nan, tw)),
            " ".join("%.2e" % v for v in dev), " ".join("%.2e" % v for v in sdev),
            ("; words of the a planes that differ: " + " ".join(str(v) for v in adev)) if emit else ""), flush=True)
