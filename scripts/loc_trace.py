#!/usr/bin/env python3
"""Phase times of the per-sample localisation-net backward kernel (timing build scripts/_trace/libloctrace.so, -DCRNN_LOC_TRACE): workgroup 0's
s_memrealtime stamps after every phase."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]
import numpy as np, torch
from crnn_mi355x import native
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L0 = native.lib(); H = native.parse_header()
L = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libloctrace.so"))
for name, (ret, args) in H.items():
    if hasattr(L, name):
        fn = getattr(L, name); fn.restype, fn.argtypes = ret, args
B, H0, W0 = 256, 100, 32
Hs1, Ws1 = H0 // 2, W0 // 2; Ho1, Wo1 = Hs1 - 4, Ws1 - 4; Hs2, Ws2 = Ho1 // 2, Wo1 // 2; Ho2, Wo2 = Hs2 - 4, Ws2 - 4; F = Ho2 * Wo2 * 20
r = lambda *s: torch.randn(*s, device="cuda")
x = torch.rand(B, H0, W0, device="cuda"); k1 = r(5, 5, 1, 20) * 0.3; bc1 = r(20) * 0.1; k2 = r(5, 5, 20, 20) * 0.1; bc2 = r(20) * 0.1
w1 = r(F, 50) * 0.1; b1 = r(50) * 0.1; w2 = r(50, 6) * 0.3; b2 = r(6)
p1 = torch.zeros(B, Hs1, Ws1, device="cuda"); c1 = torch.zeros(B, Ho1, Wo1, 20, device="cuda"); p2 = torch.zeros(B, Hs2, Ws2, 20, device="cuda")
fl = torch.zeros(B, F, device="cuda"); f1 = torch.zeros(B, 50, device="cuda"); th = torch.zeros(B, 6, device="cuda")
assert L0.crnn_loc_net_fwd(P(x), P(k1), P(bc1), P(k2), P(bc2), P(w1), P(b1), P(w2), P(b2), P(p1), P(c1), P(p2), P(fl), P(f1), P(th), B, H0, W0, S()) == 0
dth = r(B, 6); dfc1 = torch.zeros(B, 50, device="cuda")
terms = torch.zeros(L0.crnn_loc_net_bwd_scratch(B) + 64, device="cuda")
g = [torch.zeros(n, device="cuda") for n in (500, 20, 10000, 20, F * 50, 50, 300, 6)]
for it in range(3):
    assert L.crnn_loc_net_bwd(P(dth), P(fl), P(f1), P(p1), P(c1), P(p2), P(w1), P(w2), P(k2), P(dfc1), P(terms), *[P(t) for t in g], B, H0, W0, S()) == 0
    torch.cuda.synchronize()
st = terms[L0.crnn_loc_net_bwd_scratch(B):].view(torch.int64)[:6].cpu().numpy().astype(np.float64) / 100.0
names = ["loads + first maxima + dfc1", "dflat", "conv2 dgrad + dk2 terms", "dk1 quarters", "combine + biases"]
for n, a, b in zip(names, st[:-1], st[1:]): print("%-30s %6.2f us" % (n, b - a))
print("total %.2f us" % (st[-1] - st[0]))
