import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np, torch
from crnn_mi355x import native
L = native.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, C, rate) in [(3, 104, 36, 64, 0.0), (3, 104, 36, 64, 0.1), (2, 52, 9, 512, 0.1)]:
    n = B * H * W * C
    g = torch.Generator("cuda").manual_seed(1)
    q = (torch.randn(n, device="cuda", generator=g) * 1.5).bfloat16()
    d = torch.randn(n, device="cuda", generator=g).bfloat16(); da = torch.randn(n, device="cuda", generator=g).bfloat16()
    k = torch.randn(9, C, device="cuda"); dk = torch.zeros(9, C, device="cuda"); dx = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    st2 = torch.cat([torch.randn(C), 1 + torch.rand(C), 1 + 0.5 * torch.randn(C), 1.5 + 1.5 * torch.randn(C)]).cuda()
    st1 = torch.cat([torch.randn(C) * 0.1, 1 + torch.rand(C), 1 + 0.3 * torch.randn(C), 1.0 + 0.5 * torch.randn(C)]).cuda()
    coef = (torch.randn(2 * C) * 1e-3).cuda()
    keep = torch.zeros(n // 8 + 64, dtype=torch.uint8, device="cuda"); L.crnn_dropout_keep_bytes(P(keep), n // 8, rate, 7, 3, S())
    rows = L.crnn_dwconv_bwd_stream_rows(B, H, W, C)
    sc = torch.zeros(rows * 9 * C + 64, device="cuda"); parts = torch.zeros(rows, 2, C, device="cuda")
    rc = L.crnn_dwconv3x3_bwd_stream_pro(P(d), P(da), P(st1), P(coef), P(q), P(st2), rate, P(keep), P(k), P(dx), P(dk), P(sc), P(parts), B, H, W, C, S())
    torch.cuda.synchronize()
    # host reference of gy per (image,row)
    qf = q.float().view(B, H, W, C); dxf = dx.float().view(B, H, W, C)
    t = torch.addcmul(st2[3 * C:], qf, st2[2 * C:3 * C])      # fma not bit exact, fine
    bits = ((keep[:n // 8].to(torch.int32).unsqueeze(1) >> torch.arange(8, device="cuda", dtype=torch.int32)) & 1).reshape(B, H, W, C).float()
    ik = 1.0 / (1.0 - rate) if rate > 0 else 1.0
    gy = dxf * bits * ik * ((t > 0) & (t < 6)).float()
    nwgb = rows // B; HB = H // nwgb
    ref = gy.view(B, nwgb, HB, W, C).sum((2, 3)).view(rows, C)
    got = parts[:, 0, :]
    err = (got - ref).abs().max(1).values
    print(B, H, W, C, rate, "rc", rc, "rows", rows, "HB", HB, "max err per band (first 8):", [round(float(e), 3) for e in err[:8]], "scale", float(ref.abs().max()))
    xh = (qf - st2[:C]) / torch.sqrt(st2[C:2 * C] + 1e-3)
    ref2 = (gy * xh).view(B, nwgb, HB, W, C).sum((2, 3)).view(rows, C)
    print("   xhat sums err", float((parts[:, 1, :] - ref2).abs().max()), "scale", float(ref2.abs().max()))
    # which rows contribute? compare with the reference shifted by +-1 row
    for sh in (-1, 1):
        gys = torch.roll(dxf, sh, 1) * bits * ik * ((t > 0) & (t < 6)).float()
        r2 = gys.view(B, nwgb, HB, W, C).sum((2, 3)).view(rows, C)
        print("   vs dx shifted by", sh, "row: err", float((got - r2).abs().max()))
