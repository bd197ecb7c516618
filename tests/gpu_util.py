"""Helpers for the -m gpu parity tests: call libcrnn_mi355x through its C ABI on torch device buffers."""
import ctypes

import numpy as np
import torch

from crnn_mi355x import native


def L():
    return native.lib()


_KEEP = []  # device buffers stay referenced until the test ends: a temporary passed as P(dev(x)) would be
#             returned to torch's caching allocator (and re-used by the next dev()) before the kernel runs


def dev(a, dtype=np.float32):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()
    _KEEP.append(t)
    return t


def release():
    torch.cuda.synchronize()
    del _KEEP[:]


def zeros(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device="cuda")


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ok(code):
    assert code == 0, "libcrnn_mi355x returned %d" % code
    torch.cuda.synchronize()


def host(t):
    return t.detach().cpu().numpy()


def maxerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()), float(np.abs(b).max())


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=""):
    err, scale = maxerr(a, b)
    assert err <= atol + rtol * scale, "%s: max|diff| %.3e vs scale %.3e" % (what, err, scale)


def gemm(mode, A, B, M, N, K, lda, ldb, ldc, bias=None, act=0, acc=0, perm=0, C=None, scratch_mb=64, rows_out=None):
    rows_out = rows_out or M
    Cd = zeros(rows_out, ldc) if C is None else C
    scr = zeros(scratch_mb * 1024 * 1024 // 4) if scratch_mb else None
    ok(L().crnn_gemm_f32(mode, P(A), P(B), P(Cd), M, N, K, lda, ldb, ldc, P(bias), act, acc, perm, P(scr),
                         (scratch_mb * 1024 * 1024) if scratch_mb else 0, S()))
    return Cd
