#!/usr/bin/env python3
"""In-kernel timeline of the streaming weight-gradient kernel (trace build scripts/_trace/libwg_trace.so): workgroup 8, IO wave 0 and
MFMA wave 0, the first 40 chunks."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
L = ctypes.CDLL(os.path.join(ROOT, "scripts/_trace/libwg_trace.so"))
L.crnn_pwconv_bnrelu6_wgrad_stream.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
P = lambda t: ctypes.c_void_p(t.data_ptr())
M, N, K = 119808, 512, 512
d = torch.randn(M, K, device="cuda").to(torch.bfloat16); g = torch.randn(M, N, device="cuda").to(torch.bfloat16)
st = torch.randn(4 * K, device="cuda").abs() + 0.5; dw = torch.empty(K, N, device="cuda"); scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
trace = torch.zeros(512, dtype=torch.int64, device="cuda")
os.environ["CRNN_WG_TRACE"] = str(trace.data_ptr())
for _ in range(3):
    assert L.crnn_pwconv_bnrelu6_wgrad_stream(P(d), P(st), P(g), P(dw), M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
torch.cuda.synchronize()
t = trace.cpu().numpy(); io = t[:256].reshape(64, 4); mm = t[256:].reshape(64, 4)
t0 = io[0, 0]
print("chunk | IO wave: start  wait-lgkm  barrier  write+load-issue | MFMA wave: barrier-wait  work   (ns)")
for s in range(3, 40):
    print("%3d | %7d %5d %5d %5d | %5d %5d" % (s, (io[s, 0] - t0) * 10, (io[s, 1] - io[s, 0]) * 10, (io[s, 2] - io[s, 1]) * 10, (io[s, 3] - io[s, 2]) * 10,
                                             (mm[s, 1] - mm[s, 0]) * 10, (mm[s + 1, 0] - mm[s, 1]) * 10))
