#!/usr/bin/env python3
"""In-kernel timeline of the three-plane GEMM (crnn_gemm_f32x3) on the parity mode's deep-K pointwise shape: s_memrealtime stamps of one
workgroup (trace build scripts/_trace/libgemm_exp.so, -DCRNN_GEMM_EXP -DCRNN_EXPERIMENT_HOOKS).  Per chunk: planes stored | barrier |
96 MFMAs with the next chunk's split in their shadow | barrier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_trace", os.environ.get("GEMM_LIB", "libgemm_exp.so")))
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 119808, 512, 512
A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda") * 0.05; C = torch.empty(M, N, device="cuda")
trace = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
os.environ["CRNN_GEMM_TRACE"] = hex(trace.data_ptr())
args = [ctypes.c_int(0), P(A), P(B), P(C), M, N, K, K, N, N, None, 0, 0, 0, None, ctypes.c_size_t(0), S()]
lib.crnn_gemm_f32x3.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
for _ in range(3):
    rc = lib.crnn_gemm_f32x3(*args); assert rc == 0, rc
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): lib.crnn_gemm_f32x3(*args)
e1.record(); torch.cuda.synchronize()
t = trace.cpu().numpy()
print("kernel %.1f us" % (e0.elapsed_time(e1) / 5 * 1e3))
if t.max() == 0: sys.exit(0)      # a build without stamps: the timing only
if os.environ.get("GEMM_LIB", "").startswith("libgemm_x3p"):
    # producer-wave kernel: 64 stamps per wave (waves 0-3 multiply, 4-7 stage); ns relative to the workgroup's first stamp
    t = t.reshape(8, 64); t0 = t[t > 0].min()
    for w in (0, 3, 4, 7):
        r = t[w][t[w] > 0]
        print("wave %d (%s): first %d ns; deltas (ns): %s" % (w, "mfma: [MFMAs issued | barrier passed]..." if w < 4 else "stage: start | [planes stored | barrier passed]...",
              (r[0] - t0) * 10, " ".join("%d" % x for x in np.diff(r) * 10)))
else:
    t = t[t > 0]
    d = np.diff(t) * 10
    print("stamps of workgroup 300 (ns between stamps):")
    print(" ".join("%d" % x for x in d))
