#!/usr/bin/env python3
"""Launch sequence for the HBM-traffic counters of the fp32 row-stream kernels (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python scripts/dw_f32_pmc.py):
per shape of the un-pooled block outputs at batch 256, three launches each of the plain forward, the prologue forward (dropout .1), the plain backward and
the prologue backward with the BatchNorm-2 statistics.  scripts/pmc_f32_summary.py folds the counter files into profiles/<round>_pmc_dw_f32.json."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, F, REP = 256, 0, 3
SHAPES = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 9, 512)]
if __name__ == "__main__":
    stat = torch.empty(4096 * 2 * 512, device="cuda")
    for (h, w, c) in SHAPES:
        n = B * h * w * c
        q = torch.randn(n, device="cuda"); x = torch.randn(n, device="cuda"); d = torch.empty(n, device="cuda"); da = torch.randn(n, device="cuda"); dx = torch.empty(n, device="cuda")
        k = torch.randn(9, c, device="cuda"); dk = torch.zeros(9, c, device="cuda")
        st2 = torch.cat([torch.randn(c), 1 + torch.rand(c), 1 + 0.5 * torch.randn(c), 1.5 + 1.5 * torch.randn(c)]).cuda()
        st1 = torch.cat([torch.randn(c) * 0.1, 1 + torch.rand(c), 1 + 0.3 * torch.randn(c), 1.0 + 0.5 * torch.randn(c)]).cuda()
        coef = (torch.randn(2 * c) * 1e-3).cuda()
        keep = torch.zeros(n // 8 + 64, dtype=torch.uint8, device="cuda"); L.crnn_dropout_keep_bytes(P(keep), n // 8, 0.1, 7, 3, S())
        rows = max(L.crnn_dwconv_fwd_stream_rows_ex(B, h, w, c, F) * 2, L.crnn_dwconv_bwd_stream_rows_ex(B, h, w, c, F) * 9)
        pt = torch.empty(rows * c + 64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(REP): assert L.crnn_dwconv3x3_fwd_stream_dt(P(x), P(k), P(d), P(pt), B, h, w, c, 0, F, S()) == 0
        for _ in range(REP): assert L.crnn_dwconv3x3_fwd_stream_pro_ex(P(q), P(st2), 0.1, P(keep), P(k), P(d), P(pt), B, h, w, c, F, S()) == 0
        for _ in range(REP): assert L.crnn_dwconv3x3_bwd_stream_ex(P(d), P(da), P(st1), P(coef), P(x), P(k), P(dx), P(dk), P(pt), B, h, w, c, F, S()) == 0
        for _ in range(REP): assert L.crnn_dwconv3x3_bwd_stream_pro_ex(P(d), P(da), P(st1), P(coef), P(q), P(st2), 0.1, P(keep), P(k), P(dx), P(dk), P(pt), P(stat), B, h, w, c, F, S()) == 0
        torch.cuda.synchronize()
