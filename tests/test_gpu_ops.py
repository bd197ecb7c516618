"""-m gpu parity tests, one HIP operator at a time, through the C ABI, against the CPU oracle.
Floating point: tolerance 1e-4 relative (fp32 kernels vs the fp64 oracle); integer outputs bit-exact."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ops, ctc, model as M
from gpu_util import L, dev, zeros, P, S, ok, host, assert_close, gemm
from crnn_mi355x import native

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(300, 130, 70), (128, 64, 32), (1000, 38, 512), (77, 20, 25), (513, 257, 129),
                                   (4, 6, 50), (260, 1, 64), (1, 64, 900)])
def test_gemm_modes(mode, shape):
    Mm, N, K = shape
    rs = np.random.RandomState(Mm + N + K + mode)
    A = rs.normal(size=(Mm, K)); B = rs.normal(size=(K, N))
    bias = rs.normal(size=N)
    ref = A @ B
    if mode == 0:
        Ad, Bd, lda, ldb = dev(A), dev(B), K, N
    elif mode == 1:
        Ad, Bd, lda, ldb = dev(A), dev(B.T), K, K
    else:
        Ad, Bd, lda, ldb = dev(A.T), dev(B), Mm, N
    C = gemm(mode, Ad, Bd, Mm, N, K, lda, ldb, N)
    assert_close(host(C), ref, what="plain")
    C = gemm(mode, Ad, Bd, Mm, N, K, lda, ldb, N, bias=dev(bias), act=1)
    assert_close(host(C), np.maximum(ref + bias, 0), what="bias+relu")
    C0 = rs.normal(size=(Mm, N))
    C = gemm(mode, Ad, Bd, Mm, N, K, lda, ldb, N, acc=1, C=dev(C0))
    assert_close(host(C), ref + C0, what="accumulate")


def _gemm_x3(mode, A, B, M, N, K, lda, ldb, ldc, bias=None, act=0, acc=0, perm=0, C=None, scratch_mb=64):
    Cd = zeros(M, ldc) if C is None else C
    scr = zeros(scratch_mb * 1024 * 1024 // 4) if scratch_mb else None
    ok(L().crnn_gemm_f32x3(mode, P(A), P(B), P(Cd), M, N, K, lda, ldb, ldc, P(bias), act, acc, perm, P(scr),
                           (scratch_mb * 1024 * 1024) if scratch_mb else 0, S()))
    return Cd


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(300, 130, 70), (128, 64, 32), (1000, 38, 512), (77, 20, 25), (513, 257, 129), (4, 6, 50), (260, 1, 64), (1, 64, 900),
                                   (1024, 256, 512), (2048, 512, 128), (640, 128, 4608)])
def test_gemm_three_plane_products_are_fp32_accurate(mode, shape):
    """crnn_gemm_f32x3 (the parity mode's GEMMs): fp32 operands split into three bf16 planes, six bf16 MFMAs per k-step.  Against an fp64
    product its error must be of the order of the fp32-MFMA kernel's own (an fmaf chain), i.e. fp32 round-off -- far inside the 1e-3 /
    1e-4 parity tolerances -- in all three operand modes, ragged and whole tiles (guarded and unguarded instantiations), K of 25 .. 4608
    (split reductions included), with the bias / ReLU / accumulate epilogues."""
    Mm, N, K = shape
    rs = np.random.RandomState(Mm + N + K + mode)
    A = rs.normal(size=(Mm, K)) * np.exp(rs.normal(size=(Mm, 1))); B = rs.normal(size=(K, N)) * np.exp(rs.normal(size=(1, N)) * 2)   # wide dynamic range
    A = A.astype(np.float32).astype(np.float64); B = B.astype(np.float32).astype(np.float64)
    bias = rs.normal(size=N)
    ref = A @ B
    mag = np.abs(A) @ np.abs(B)                                                # scale of the rounding error of every output
    if mode == 0:
        Ad, Bd, lda, ldb = dev(A), dev(B), K, N
    elif mode == 1:
        Ad, Bd, lda, ldb = dev(A), dev(B.T), K, K
    else:
        Ad, Bd, lda, ldb = dev(A.T), dev(B), Mm, N
    C3 = host(_gemm_x3(mode, Ad, Bd, Mm, N, K, lda, ldb, N)).astype(np.float64)
    C1 = host(gemm(mode, Ad, Bd, Mm, N, K, lda, ldb, N)).astype(np.float64)
    e3 = float((np.abs(C3 - ref) / mag).max()); e1 = float((np.abs(C1 - ref) / mag).max())
    print("x3 error %.3g of sum|a||b| (fp32 MFMA kernel: %.3g)" % (e3, e1))
    assert e3 < 4 * e1 + 3e-7, (e3, e1)                                        # fp32 round-off: 6e-8 per operation
    C = _gemm_x3(mode, Ad, Bd, Mm, N, K, lda, ldb, N, bias=dev(bias), act=1)
    assert_close(host(C), np.maximum(ref + bias, 0), rtol=1e-5, atol=3e-6 * mag.max(), what="bias+relu")
    C0 = rs.normal(size=(Mm, N))
    C = _gemm_x3(mode, Ad, Bd, Mm, N, K, lda, ldb, N, acc=1, C=dev(C0))
    assert_close(host(C), ref + C0, rtol=1e-5, atol=3e-6 * mag.max(), what="accumulate")


def test_gemm_three_plane_split_reduction_permute_and_statistics():
    rs = np.random.RandomState(3)
    K, Mm, N = 40000, 64, 128
    X = rs.normal(size=(K, Mm)); G = rs.normal(size=(K, N))
    C = _gemm_x3(2, dev(X), dev(G), Mm, N, K, Mm, N, N)
    assert_close(host(C), X.T @ G, rtol=1e-5, atol=1e-3, what="TN split")
    A = rs.normal(size=(130, 4608)); W = rs.normal(size=(4608, 128)) * 0.02; b = rs.normal(size=128)
    C3 = _gemm_x3(0, dev(A), dev(W), 130, 128, 4608, 4608, 128, 128, bias=dev(b), act=1, perm=10)
    ref = np.maximum(A @ W + b, 0).reshape(13, 10, 128).transpose(1, 0, 2).reshape(130, 128)
    assert_close(host(C3), ref, rtol=1e-5, atol=1e-5, what="NN split+epilogue")
    # pointwise conv with the BatchNorm statistics epilogue (crnn_pwconv_fwd, product selector 2) against the fp32-MFMA path
    M, K2, N2 = 128 * 37 + 5, 64, 128
    a = rs.normal(size=(M, K2)); w = rs.normal(size=(K2, N2)) * 0.2
    rows = L().crnn_pwconv_stat_rows(M)
    q3, q1 = zeros(M, N2), zeros(M, N2); st3, st1 = zeros(rows * 2 * N2), zeros(rows * 2 * N2)
    ok(L().crnn_pwconv_fwd(P(dev(a)), P(dev(w)), P(q3), M, N2, K2, P(st3), None, 2, 0, 0, 0, 0, S()))
    ok(L().crnn_pwconv_fwd(P(dev(a)), P(dev(w)), P(q1), M, N2, K2, P(st1), None, 0, 0, 0, 0, 0, S()))
    assert_close(host(q3), host(q1), rtol=1e-5, atol=2e-6 * float(np.abs(host(q1)).max()), what="pwconv x3 vs fp32")
    s3 = host(st3).reshape(rows, 2, N2).sum(0); s1 = host(st1).reshape(rows, 2, N2).sum(0)
    assert_close(s3, s1, rtol=1e-5, atol=1e-4 * float(np.abs(s1).max()), what="pwconv statistics")


def test_gemm_row_permute_and_ld():
    rs = np.random.RandomState(0)
    Bn, T, K, N = 6, 5, 40, 24
    A = rs.normal(size=(Bn * T, K)); W = rs.normal(size=(K, N))
    C = gemm(0, dev(A), dev(W), Bn * T, N, K, K, N, N, perm=T)
    ref = (A @ W).reshape(Bn, T, N).transpose(1, 0, 2).reshape(T * Bn, N)
    assert_close(host(C), ref, what="perm")
    # strided output (ldc > N) and strided A (lda > K)
    Apad = np.zeros((Bn * T, K + 8)); Apad[:, :K] = A
    Cd = zeros(Bn * T, N + 16)
    gemm(0, dev(Apad), dev(W), Bn * T, N, K, K + 8, N, N + 16, C=Cd)
    assert_close(host(Cd)[:, :N], A @ W, what="ld")
    assert np.all(host(Cd)[:, N:] == 0)


def test_gemm_split_reduction_large_k():
    rs = np.random.RandomState(1)
    K, Mm, N = 40000, 64, 128
    X = rs.normal(size=(K, Mm)); G = rs.normal(size=(K, N))
    C = gemm(2, dev(X), dev(G), Mm, N, K, Mm, N, N)
    assert_close(host(C), X.T @ G, rtol=2e-4, what="TN split")
    C2 = gemm(2, dev(X), dev(G), Mm, N, K, Mm, N, N, scratch_mb=0)
    assert_close(host(C2), X.T @ G, rtol=2e-4, what="TN unsplit")
    # NN with long K and few tiles (dense1 shape class) + bias/relu/permute through the split path
    A = rs.normal(size=(130, 4608)); W = rs.normal(size=(4608, 128)) * 0.02; b = rs.normal(size=128)
    C3 = gemm(0, dev(A), dev(W), 130, 128, 4608, 4608, 128, 128, bias=dev(b), act=1, perm=10)
    ref = np.maximum(A @ W + b, 0).reshape(13, 10, 128).transpose(1, 0, 2).reshape(130, 128)
    assert_close(host(C3), ref, rtol=2e-4, what="NN split+epilogue")


def _bf16_round(a):
    """round-to-nearest-even to bfloat16, returned as float64 (what v_cvt_pk_bf16_f32 does to the staged operands)"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32).astype(np.float64)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(300, 130, 70), (128, 64, 64), (1000, 38, 512), (77, 20, 25), (513, 257, 129), (260, 1, 64),
                                   (64, 128, 5000)])
def test_gemm_bf16_mode_equals_fp32_accumulation_of_bf16_rounded_operands(mode, shape):
    Mm, N, K = shape
    rs = np.random.RandomState(Mm + N + K + mode)
    A = rs.normal(size=(Mm, K)).astype(np.float32); B = rs.normal(size=(K, N)).astype(np.float32)
    ref = _bf16_round(A) @ _bf16_round(B)
    if mode == 0:
        Ad, Bd, lda, ldb = dev(A), dev(B), K, N
    elif mode == 1:
        Ad, Bd, lda, ldb = dev(A), dev(B.T), K, K
    else:
        Ad, Bd, lda, ldb = dev(A.T), dev(B), Mm, N
    C = zeros(Mm, N); scr = zeros(16 * 1024 * 1024)
    bias = rs.normal(size=N)
    ok(L().crnn_gemm_bf16(mode, P(Ad), P(Bd), P(C), Mm, N, K, lda, ldb, N, P(dev(bias)), 1, 0, 0, P(scr), 64 * 1024 * 1024, S()))
    assert_close(host(C), np.maximum(ref + bias, 0), rtol=2e-5, atol=1e-4, what="bf16 products, fp32 accumulate")
    # and it is within bf16 round-off of the exact product
    ok(L().crnn_gemm_bf16(mode, P(Ad), P(Bd), P(C), Mm, N, K, lda, ldb, N, None, 0, 0, 0, P(scr), 64 * 1024 * 1024, S()))
    exact = A.astype(np.float64) @ B.astype(np.float64)
    assert np.abs(host(C) - exact).max() < 2.0 ** -7 * np.sqrt(K) * 4


# ------------------------------------------------------------------------------------------------ depthwise conv
@pytest.mark.parametrize("shape", [(3, 20, 12, 64), (2, 9, 7, 32), (2, 13, 36, 128), (2, 10, 6, 1), (1, 5, 9, 96)])
def test_dwconv_fwd_flip_wgrad_stats(shape):
    B, H, W, C = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.normal(size=shape); k = rs.normal(size=(3, 3, C)); g = rs.normal(size=shape)
    ref = ops.dwconv_fwd(x, k)
    dx_ref, dk_ref = ops.dwconv_bwd(x, k, g)
    xd, kd, gd = dev(x), dev(k), dev(g)
    out = zeros(*shape)
    ntiles = L().crnn_dwconv_num_tiles(B, H, W)
    parts = zeros(ntiles, 2, C) if C % 32 == 0 else None
    ok(L().crnn_dwconv3x3_fwd(P(xd), P(kd), P(out), P(parts), B, H, W, C, 0, S()))
    assert_close(host(out), ref, what="fwd")
    if parts is not None:
        st = host(parts).sum(0)
        assert_close(st[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-3, what="sum")
        assert_close(st[1], (ref ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-3, what="sumsq")
    dx = zeros(*shape)
    ok(L().crnn_dwconv3x3_fwd(P(gd), P(kd), P(dx), None, B, H, W, C, 1, S()))
    assert_close(host(dx), dx_ref, what="dgrad")
    dk = zeros(9, C); scr = zeros(max(1, ntiles * 9 * C))
    ok(L().crnn_dwconv3x3_wgrad(P(xd), P(gd), P(dk), P(scr), B, H, W, C, S()))
    assert_close(host(dk).reshape(3, 3, C), dk_ref, rtol=1e-4, atol=1e-4, what="wgrad")


@pytest.mark.parametrize("shape", [(2, 104, 36, 128), (3, 52, 18, 256), (2, 52, 9, 512), (2, 13, 7, 64), (1, 5, 61, 64), (5, 33, 20, 128), (1, 3, 100, 64), (2, 1, 1, 64), (1, 40, 25, 64),
                                   (3, 104, 36, 64), (2, 204, 36, 128), (40, 52, 18, 256), (128, 52, 9, 512), (2, 7, 36, 64), (1, 3, 18, 256),
                                   # round 5: image widths 48 (ranges of 208 columns: three and a quarter of the five waves busy) and 64 (272 columns) on the row-stream kernel
                                   (3, 64, 52, 64), (2, 64, 52, 128), (4, 32, 26, 256), (40, 32, 13, 512), (2, 104, 68, 64), (2, 104, 68, 128), (3, 52, 34, 256), (2, 52, 17, 512)])
def test_fused_depthwise_stage_backward_equals_the_three_kernel_sequence(shape):
    """crnn_dwconv3x3_bwd_fused (BatchNorm-backward pass 2 formed in the halo-tile fill, depthwise weight and data gradients from one
    pass over the tiles) against crnn_bn_bwd_ex + crnn_dwconv3x3_wgrad_ex + crnn_dwconv3x3_fwd_ex(flip=1) on bf16 tensors: the data
    gradient bit for bit, the weight gradient to fp32 summation-order round-off; both against the fp64 oracle.  Several bands per
    workgroup, ragged widths and heights, band counts that do not divide."""
    B, H, W, C = shape
    rs = np.random.RandomState(sum(shape))
    x = _bf16_round(rs.normal(size=shape)); k = rs.normal(size=(3, 3, C))
    d = _bf16_round(ops.dwconv_fwd(x, k))
    gamma, beta = rs.normal(size=C) * 0.3 + 1.0, rs.normal(size=C) * 0.5 + 1.0
    da = _bf16_round(rs.normal(size=shape))
    xd, dd, dad, kd = _to_bf16_dev(x), _to_bf16_dev(d), _to_bf16_dev(da), dev(k)
    M = B * H * W
    mean = d.reshape(M, C).mean(0); var = d.reshape(M, C).var(0)
    scale = gamma / np.sqrt(var + 1e-3); shift = beta - mean * scale
    st = dev(np.concatenate([mean, var, scale, shift]))
    gd = dev(gamma)
    nparts = max(L().crnn_bn_bwd_chunks(M), L().crnn_dwconv_num_tiles(B, H, W) * 9, L().crnn_dwconv_bwd_fused_rows(B, H, W, C) * 9)
    # reference sequence on the device
    gin = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    dg1, db1 = zeros(C), zeros(C); parts = zeros(nparts * 2 * C + 9 * C * nparts); coef = zeros(2 * C)
    ok(L().crnn_bn_bwd_ex(P(dd), P(dad), P(st), P(gd), P(gin), P(dg1), P(db1), P(parts), P(coef), B, H, W, C, 1, 1, 0.0, 0, 0, 1, S()))
    dk1 = zeros(9, C)
    ok(L().crnn_dwconv3x3_wgrad_ex(P(xd), P(gin), P(dk1), P(parts), B, H, W, C, 1, S()))
    dx1 = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_dwconv3x3_fwd_ex(P(gin), P(kd), P(dx1), None, B, H, W, C, 1, 1, S()))
    # fused: statistics pass only, then the one kernel
    dg2, db2 = zeros(C), zeros(C); coef2 = zeros(2 * C)
    ok(L().crnn_bn_bwd_ex(P(dd), P(dad), P(st), P(gd), None, P(dg2), P(db2), P(parts), P(coef2), B, H, W, C, 1, 1, 0.0, 0, 0, 1, S()))
    assert torch.equal(coef, coef2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    assert L().crnn_dwconv_bwd_fused_supported(H, W, C) == 0
    dx2 = torch.full((B, H, W, C), 9.0, dtype=torch.bfloat16, device="cuda"); dk2 = zeros(9, C)
    ok(L().crnn_dwconv3x3_bwd_fused(P(dd), P(dad), P(st), P(coef2), P(xd), P(kd), P(dx2), P(dk2), P(parts), B, H, W, C, S()))
    assert torch.equal(dx1, dx2), "dx: max diff %g" % float((dx1.float() - dx2.float()).abs().max())
    assert_close(host(dk2), host(dk1), rtol=2e-5, atol=2e-4 * np.abs(host(dk1)).max(), what="dk fused vs sequence")
    # fp64 oracle of the same stage (gradient w.r.t. d through BN + ReLU6, then the conv gradients); dd is stored as bf16 on both paths
    xhat = (d - mean) / np.sqrt(var + 1e-3); y = xhat * gamma + beta
    gy = da * ((y > 0) & (y < 6))
    ddn = scale * (gy - gy.reshape(M, C).mean(0) - xhat * (gy * xhat).reshape(M, C).mean(0))
    dx_ref, dk_ref = ops.dwconv_bwd(x, k, _bf16_round(ddn))
    assert_close(_f(dx2), dx_ref, rtol=2.0 ** -7, atol=2e-2 * np.abs(dx_ref).max(), what="dx vs oracle")
    assert_close(host(dk2).reshape(3, 3, C), dk_ref, rtol=1e-2, atol=1e-2 * np.abs(dk_ref).max(), what="dk vs oracle")
    assert L().crnn_dwconv_bwd_fused_supported(H, W, 32) == -3 and L().crnn_dwconv_bwd_fused_supported(H, W, 96) == -3
    # the row-stream kernel of the same stage, where its shape rule holds: dx bit for bit, dk to summation round-off
    if W in (52, 26, 13, 68, 34, 17):
        assert L().crnn_dwconv_bwd_stream_supported(B, H, W, C) == 0, "image widths 48 / 64 take the row-stream backward since round 5"
    if L().crnn_dwconv_bwd_stream_supported(B, H, W, C) == 0:
        rows = L().crnn_dwconv_bwd_stream_rows(B, H, W, C)
        dx3 = torch.full((B * H * W * C + 64,), 9.0, dtype=torch.bfloat16, device="cuda"); dk3 = zeros(9, C)
        sc3 = torch.full((rows * 9 * C + 16,), 5.0, device="cuda")
        ok(L().crnn_dwconv3x3_bwd_stream(P(dd), P(dad), P(st), P(coef2), P(xd), P(kd), P(dx3), P(dk3), P(sc3), B, H, W, C, S()))
        assert torch.equal(dx3[:-64].view(torch.int16), dx1.reshape(-1).view(torch.int16)), "stream dx: max diff %g" % float((dx1.reshape(-1).float() - dx3[:-64].float()).abs().max())
        assert bool((dx3[-64:] == 9.0).all()) and bool((sc3[-16:] == 5.0).all())
        assert_close(host(dk3), host(dk1), rtol=2e-5, atol=2e-4 * np.abs(host(dk1)).max(), what="dk stream vs sequence")
        dx4 = torch.zeros_like(dx3); dk4 = zeros(9, C)
        ok(L().crnn_dwconv3x3_bwd_stream(P(dd), P(dad), P(st), P(coef2), P(xd), P(kd), P(dx4), P(dk4), P(sc3), B, H, W, C, S()))
        assert torch.equal(dx4[:-64], dx3[:-64]) and torch.equal(dk3, dk4), "repeat launches differ"
    else:
        assert L().crnn_dwconv_bwd_stream_rows(B, H, W, C) == 0


@pytest.mark.parametrize("shape", [(2, 104, 36, 128), (3, 52, 18, 256), (2, 52, 9, 512), (3, 104, 36, 64), (2, 204, 36, 128), (40, 52, 18, 256), (70, 52, 9, 512),
                                   (1, 40, 36, 64), (2, 13, 7, 64), (1, 3, 18, 256), (3, 64, 52, 64), (4, 32, 26, 256), (2, 104, 68, 128), (2, 52, 17, 512)])
def test_fp32_row_stream_depthwise_stage_backward_equals_the_three_kernel_sequence(shape):
    """crnn_dwconv3x3_bwd_stream_ex(dtype = fp32) -- the parity mode's depthwise-stage backward in one pass (round 4) -- against
    crnn_bn_bwd_ex + crnn_dwconv3x3_wgrad_ex + crnn_dwconv3x3_fwd_ex(flip = 1) on fp32 tensors: the data gradient bit for bit, the weight
    gradient to fp32 summation-order round-off; both against the fp64 oracle at the fp32 kernels' tolerance.  Shapes the rule refuses report so."""
    B, H, W, C = shape
    rs = np.random.RandomState(sum(shape) + 1)
    x = rs.normal(size=shape).astype(np.float32).astype(np.float64); k = rs.normal(size=(3, 3, C))
    d = ops.dwconv_fwd(x, k).astype(np.float32).astype(np.float64)
    gamma, beta = rs.normal(size=C) * 0.3 + 1.0, rs.normal(size=C) * 0.5 + 1.0
    da = rs.normal(size=shape).astype(np.float32).astype(np.float64)
    xd, dd, dad, kd = dev(x), dev(d), dev(da), dev(k)
    M = B * H * W
    mean = d.reshape(M, C).mean(0); var = d.reshape(M, C).var(0)
    scale = gamma / np.sqrt(var + 1e-3); shift = beta - mean * scale
    st = dev(np.concatenate([mean, var, scale, shift])); gd = dev(gamma)
    sup = L().crnn_dwconv_bwd_stream_supported_ex(B, H, W, C, 0)
    rows = L().crnn_dwconv_bwd_stream_rows_ex(B, H, W, C, 0)
    if sup != 0:
        assert sup == -3 and rows == 0
        assert L().crnn_dwconv3x3_bwd_stream_ex(P(dd), P(dad), P(st), P(st), P(xd), P(kd), P(dd), P(kd), P(kd), B, H, W, C, 0, S()) == -3
        return
    assert rows >= B
    nparts = max(L().crnn_bn_bwd_chunks(M), L().crnn_dwconv_num_tiles(B, H, W) * 9, rows * 9)
    gin = zeros(B, H, W, C); dg1, db1 = zeros(C), zeros(C); parts = zeros(nparts * 2 * C + 9 * C * nparts); coef = zeros(2 * C)
    ok(L().crnn_bn_bwd_ex(P(dd), P(dad), P(st), P(gd), P(gin), P(dg1), P(db1), P(parts), P(coef), B, H, W, C, 1, 1, 0.0, 0, 0, 0, S()))
    dk1 = zeros(9, C)
    ok(L().crnn_dwconv3x3_wgrad_ex(P(xd), P(gin), P(dk1), P(parts), B, H, W, C, 0, S()))
    dx1 = zeros(B, H, W, C)
    ok(L().crnn_dwconv3x3_fwd_ex(P(gin), P(kd), P(dx1), None, B, H, W, C, 1, 0, S()))
    dx2 = torch.full((B * H * W * C + 64,), 9.0, device="cuda"); dk2 = zeros(9, C)
    sc2 = torch.full((rows * 9 * C + 16,), 5.0, device="cuda")
    ok(L().crnn_dwconv3x3_bwd_stream_ex(P(dd), P(dad), P(st), P(coef), P(xd), P(kd), P(dx2), P(dk2), P(sc2), B, H, W, C, 0, S()))
    assert torch.equal(dx2[:-64].view(torch.int32), dx1.reshape(-1).view(torch.int32)), "stream dx: max diff %g" % float((dx1.reshape(-1) - dx2[:-64]).abs().max())
    assert bool((dx2[-64:] == 9.0).all()) and bool((sc2[-16:] == 5.0).all())
    assert_close(host(dk2), host(dk1), rtol=2e-5, atol=2e-5 * np.abs(host(dk1)).max(), what="dk stream vs sequence")
    dx3 = torch.zeros_like(dx2); dk3 = zeros(9, C)
    ok(L().crnn_dwconv3x3_bwd_stream_ex(P(dd), P(dad), P(st), P(coef), P(xd), P(kd), P(dx3), P(dk3), P(sc2), B, H, W, C, 0, S()))
    assert torch.equal(dx3[:-64], dx2[:-64]) and torch.equal(dk2, dk3), "repeat launches differ"
    # fp64 oracle of the same stage
    xhat = (d - mean) / np.sqrt(var + 1e-3); y = xhat * gamma + beta
    gy = da * ((y > 0) & (y < 6))
    ddn = scale * (gy - gy.reshape(M, C).mean(0) - xhat * (gy * xhat).reshape(M, C).mean(0))
    dx_ref, dk_ref = ops.dwconv_bwd(x, k, ddn)
    near = np.minimum(np.abs(y), np.abs(y - 6)) < 1e-4          # (activations within round-off of a ReLU6 threshold may gate differently)
    if not near.any():
        assert_close(host(dx2[:-64]).reshape(shape), dx_ref, rtol=1e-4, atol=1e-4 * np.abs(dx_ref).max(), what="dx vs oracle")
        assert_close(host(dk2).reshape(3, 3, C), dk_ref, rtol=1e-4, atol=1e-4 * np.abs(dk_ref).max(), what="dk vs oracle")
    # the bf16 entry points through the same dispatcher
    assert L().crnn_dwconv_bwd_stream_supported_ex(B, H, W, C, 1) == L().crnn_dwconv_bwd_stream_supported(B, H, W, C)
    assert L().crnn_dwconv_bwd_stream_rows_ex(B, H, W, C, 1) == L().crnn_dwconv_bwd_stream_rows(B, H, W, C)
    assert L().crnn_dwconv_bwd_stream_supported_ex(B, H, W, C, 7) == -2


# ------------------------------------------------------------------------------------------------ BatchNorm chain
def _bn_state(x, gamma, beta):
    Mrows = x.size // x.shape[-1]
    C = x.shape[-1]
    xd = dev(x)
    chunks = L().crnn_colreduce_chunks(Mrows)
    parts = zeros(chunks, 2, C)
    ok(L().crnn_colreduce(P(xd), P(parts), Mrows, C, C, 2, S()))
    st = zeros(4 * C)
    ok(L().crnn_bn_finalize(P(parts), chunks, C, Mrows, P(dev(gamma)), P(dev(beta)), P(st), S()))
    return xd, st


@pytest.mark.parametrize("shape,pool,rate", [((3, 8, 6, 64), (1, 1), 0.0), ((2, 8, 6, 32), (2, 2), 0.0), ((2, 6, 8, 128), (1, 2), 0.1),
                                            ((2, 7, 5, 1), (1, 1), 0.0), ((2, 9, 7, 36), (2, 2), 0.1)])
def test_bn_relu6_pool_dropout_fwd_bwd(shape, pool, rate):
    B, H, W, C = shape
    ph, pw = pool
    rs = np.random.RandomState(sum(shape) + ph)
    x = rs.normal(size=shape) * 3 + 1.0
    x = np.round(x * 4) / 4  # coarse grid => ties / saturated ReLU6 values inside pool windows
    gamma = 1 + 0.3 * rs.normal(size=C); beta = 0.5 * rs.normal(size=C) + 1.5
    xd, st = _bn_state(x, gamma, beta)
    y_bn, mean, var = ops.bn_train_fwd(x, gamma, beta)
    sth = host(st)
    assert_close(sth[:C], mean, what="mean"); assert_close(sth[C:2 * C], var, what="var")
    Ho, Wo = H // ph, W // pw
    seed, layer = 12345, 3
    mask = zeros(B * Ho * Wo * C)
    ok(L().crnn_dropout_mask(P(mask), mask.numel(), rate, seed, layer, S()))
    mk = host(mask).reshape(B, Ho, Wo, C)
    if rate > 0:
        keep = (mk > 0).mean()
        assert abs(keep - (1 - rate)) < 0.08 and set(np.unique(mk)).issubset({0.0, np.float32(1 / (1 - rate))})
    r = ops.relu6_fwd(y_bn)
    ref = ops.maxpool_fwd(r, ph, pw) * mk
    y = zeros(B, Ho, Wo, C)
    ok(L().crnn_bn_act_pool_drop(P(xd), P(st), P(y), B, H, W, C, ph, pw, rate, seed, layer, S()))
    assert_close(host(y), ref, what="fwd")
    # backward
    g = rs.normal(size=(B, Ho, Wo, C))
    gr = ops.maxpool_bwd(r, g * mk, ph, pw)
    gr = ops.relu6_bwd_from_out(r, gr)
    dx_ref, dg_ref, db_ref = ops.bn_train_bwd(x, gamma, mean, var, gr)
    dx = zeros(*shape); dgm = zeros(C); dbt = zeros(C)
    parts = zeros(L().crnn_bn_bwd_chunks(B * H * W), 2, C); coef = zeros(2 * C)
    ok(L().crnn_bn_bwd(P(xd), P(dev(g)), P(st), P(dev(gamma)), P(dx), P(dgm), P(dbt), P(parts), P(coef), B, H, W, C, ph, pw, rate, seed,
                       layer, S()))
    assert_close(host(dgm), dg_ref, rtol=2e-4, atol=1e-4, what="dgamma")
    assert_close(host(dbt), db_ref, rtol=2e-4, atol=1e-4, what="dbeta")
    assert_close(host(dx), dx_ref, rtol=2e-4, atol=1e-5, what="dx")


@pytest.mark.parametrize("pattern", [0, 1])
def test_measurement_reference_copy_is_a_copy(pattern):
    """crnn_debug_copy (bench.py copy_reference): both access patterns move exactly the bytes asked for, for sizes that do not divide by
    the workgroup count or the 16 KiB iteration, and refuse misaligned requests."""
    for nbytes, wgs in ((16, 1), (16 * 1000 + 16, 7), (9216 * 104, 256), (4 << 20, 1024)):
        src = torch.randint(0, 256, (nbytes + 64,), dtype=torch.uint8, device="cuda")
        dst = torch.zeros_like(src)
        ok(native.hooks().crnn_debug_copy(P(src), P(dst), nbytes, pattern, wgs, S()))
        assert torch.equal(dst[:nbytes], src[:nbytes]) and int(dst[nbytes:].sum()) == 0, (nbytes, wgs)
    assert native.hooks().crnn_debug_copy(P(src), P(dst), 24, pattern, 4, S()) != 0
    assert native.hooks().crnn_debug_copy(P(src), P(dst), 32, 2, 4, S()) != 0


def test_bn_inference_states_in_one_launch_equal_the_single_launches():
    """crnn_bn_infer_state_batch (the predict path's 14 BatchNorm layers in one launch) writes what n crnn_bn_infer_state calls write."""
    rs = np.random.RandomState(5)
    Cs = [1, 64, 64, 128, 128, 256, 512, 512, 38, 300]
    arrs = [[dev(rs.normal(size=C)), dev(rs.uniform(0.1, 2.0, size=C)), dev(rs.normal(size=C) * 0.3 + 1.0), dev(rs.normal(size=C))] for C in Cs]
    one = [zeros(4 * C) for C in Cs]; many = [torch.full((4 * C + 8,), 7.0, device="cuda") for C in Cs]
    for (mm, mv, g, b), C, o in zip(arrs, Cs, one):
        ok(L().crnn_bn_infer_state(P(mm), P(mv), P(g), P(b), C, P(o), S()))
    n = len(Cs)
    PA = ctypes.c_void_p * n
    col = lambda k: PA(*[a[k].data_ptr() for a in arrs])
    ok(L().crnn_bn_infer_state_batch(n, col(0), col(1), col(2), col(3), (ctypes.c_int * n)(*Cs), PA(*[t.data_ptr() for t in many]), S()))
    for o, m, C in zip(one, many, Cs):
        assert torch.equal(o, m[:4 * C]) and bool((m[4 * C:] == 7.0).all())
    assert L().crnn_bn_infer_state_batch(17, col(0), col(1), col(2), col(3), (ctypes.c_int * n)(*Cs), PA(*[t.data_ptr() for t in many]), S()) != 0


def test_bn_inference_state_and_colsum():
    rs = np.random.RandomState(5)
    C = 48
    mm, mv = rs.normal(size=C), rs.uniform(0.5, 2, size=C)
    g, b = rs.normal(size=C), rs.normal(size=C)
    st = zeros(4 * C)
    ok(L().crnn_bn_infer_state(P(dev(mm)), P(dev(mv)), P(dev(g)), P(dev(b)), C, P(st), S()))
    x = rs.normal(size=(50, C))
    y = zeros(50, C)
    ok(L().crnn_bn_act(P(dev(x)), P(st), P(y), 50, C, S()))
    assert_close(host(y), ops.relu6_fwd(ops.bn_infer_fwd(x, g, b, mm, mv)), what="infer")
    for (Mr, Cc) in [(5000, 38), (300, 1024), (64, 6), (7000, 20)]:
        z = rs.normal(size=(Mr, Cc))
        ch = L().crnn_colreduce_chunks(Mr)
        parts = zeros(ch, Cc); out = zeros(Cc)
        ok(L().crnn_colreduce(P(dev(z)), P(parts), Mr, Cc, Cc, 1, S()))
        ok(L().crnn_partials_sum(P(parts), ch, Cc, P(out), 1.0, S()))
        assert_close(host(out), z.sum(0), rtol=1e-4, atol=1e-3, what="colsum")


# ------------------------------------------------------------------------------------------------ STN pieces
def test_sampler_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "sampler_golden.npz"))
    for k in ("mj", "iam"):
        for tn in ("ident", "pert", "wild"):
            img, th, ref = z[f"{k}_{tn}_img"], z[f"{k}_{tn}_theta"], z[f"{k}_{tn}_out"]
            B, H, W, _ = img.shape
            out = zeros(B, H + 4, W + 4)
            ok(L().crnn_sampler_fwd(P(dev(img)), P(dev(th)), P(out), B, H, W, 2, S()))
            o = host(out)
            assert_close(o[:, 2:-2, 2:-2], ref[..., 0], rtol=1e-4, atol=2e-4, what=f"{k}_{tn}")
            assert np.all(o[:, :2] == 0) and np.all(o[:, -2:] == 0) and np.all(o[:, :, :2] == 0) and np.all(o[:, :, -2:] == 0)


def test_sampler_bwd_maxpool_im2col():
    rs = np.random.RandomState(2)
    B, H, W = 3, 20, 14
    img = rs.normal(size=(B, H, W, 1)); th = np.tile([1, 0, 0, 0, 1, 0], (B, 1)) + rs.uniform(-.2, .2, (B, 6))
    g = rs.normal(size=(B, H, W, 1))
    ref = ops.sampler_bwd(img, th, g)
    gp = np.pad(g[..., 0], ((0, 0), (2, 2), (2, 2)))
    dth = zeros(B, 6)
    ok(L().crnn_sampler_bwd(P(dev(img)), P(dev(th)), P(dev(gp)), P(dth), B, H, W, 2, S()))
    assert_close(host(dth), ref, rtol=5e-4, atol=1e-3, what="dtheta")
    x = np.round(rs.normal(size=(2, 9, 8, 5)) * 2) / 2
    y = zeros(2, 4, 4, 5)
    ok(L().crnn_maxpool_fwd(P(dev(x)), P(y), 2, 9, 8, 5, 2, 2, S()))
    assert np.array_equal(host(y), ops.maxpool_fwd(x, 2, 2).astype(np.float32))
    gy = rs.normal(size=(2, 4, 4, 5)); gx = zeros(2, 9, 8, 5)
    ok(L().crnn_maxpool_bwd(P(dev(x)), P(dev(gy)), P(gx), 2, 9, 8, 5, 2, 2, S()))
    assert_close(host(gx), ops.maxpool_bwd(x, gy, 2, 2), what="maxpool_bwd")
    xi = rs.normal(size=(2, 9, 8, 3))
    col = zeros(2 * 5 * 4, 75)
    ok(L().crnn_im2col(P(dev(xi)), P(col), 2, 9, 8, 3, 5, S()))
    assert np.array_equal(host(col), ops.im2col(xi, 5, 5).astype(np.float32))
    dc = rs.normal(size=(2 * 5 * 4, 75)); dxi = zeros(2, 9, 8, 3)
    ok(L().crnn_col2im(P(dev(dc)), P(dxi), 2, 9, 8, 3, 5, S()))
    assert_close(host(dxi), ops.col2im(dc, (2, 9, 8, 3), 5, 5), what="col2im")


# ------------------------------------------------------------------------------------------------ LSTM
@pytest.mark.parametrize("B,T,u,din", [(5, 7, 64, 24), (33, 6, 128, 40), (520, 3, 64, 8)])   # 520: two batch tiles per workgroup, ragged
def test_bilstm_fwd_bwd(B, T, u, din):
    rs = np.random.RandomState(B + T)
    x = rs.normal(size=(B, T, din))
    G = 4 * u
    Wt = [rs.normal(size=(din, G)) * 0.3 for _ in range(2)]
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    bb = [rs.normal(size=G) * 0.2 for _ in range(2)]
    Hs, caches = [], []
    for d in range(2):
        h, c = ops.lstm_fwd(x, Wt[d], U[d], bb[d], reverse=(d == 1))
        Hs.append(h); caches.append(c)
    tm = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))  # (B,T,.) -> (T,B,.)
    xw = [dev(tm(x @ Wt[d] + bb[d])) for d in range(2)]
    ut = [dev(U[d].T) for d in range(2)]
    hcat = zeros(T, B, 2 * u); cs = [zeros(T, B, u) for _ in range(2)]; gt = [zeros(T, B, G) for _ in range(2)]
    hb = hcat.view(-1)[u:]
    ok(L().crnn_lstm_fwd(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(cs[0]), P(cs[1]),
                         P(gt[0]), P(gt[1]), T, B, u, S()))
    hh = host(hcat)
    for d in range(2):
        assert_close(hh[:, :, d * u:(d + 1) * u], tm(Hs[d]), rtol=1e-4, atol=1e-5, what=f"h dir{d}")
        assert_close(host(cs[d]), tm(caches[d][4]), rtol=1e-4, atol=1e-5, what=f"c dir{d}")
        assert_close(host(gt[d]), tm(caches[d][5]), rtol=1e-4, atol=1e-5, what=f"gates dir{d}")
    # backward: upstream gradient on the concatenated output
    gH = rs.normal(size=(B, T, 2 * u))
    dz_ref = []
    for d in range(2):
        # recover dZ from the oracle: rerun its loop (lstm_bwd returns dx = dZ W^T; solve via separate call)
        dx, dW, dU, db = ops.lstm_bwd(caches[d], gH[..., d * u:(d + 1) * u])
        dz_ref.append((dx, dW, dU, db))
    gd = dev(tm(gH))
    dz = [zeros(T, B, G) for _ in range(2)]; dc = [zeros(B, u) for _ in range(2)]
    Ud = [dev(U[d]) for d in range(2)]
    ok(L().crnn_lstm_bwd(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), ctypes.c_void_p(gd.data_ptr() + 4 * u), 2 * u,
                         P(dz[0]), P(dz[1]), P(dc[0]), P(dc[1]), T, B, u, S()))
    for d in range(2):
        dzh = np.swapaxes(host(dz[d]).astype(np.float64), 0, 1)  # (B,T,G)
        dx, dW, dU, db = dz_ref[d]
        assert_close(dzh @ Wt[d].T, dx, rtol=2e-4, atol=1e-5, what=f"dx dir{d}")
        assert_close(x.reshape(B * T, -1).T @ dzh.reshape(B * T, G), dW, rtol=2e-4, atol=1e-4, what=f"dW dir{d}")
        assert_close(dzh.reshape(B * T, G).sum(0), db, rtol=2e-4, atol=1e-4, what=f"db dir{d}")


def _lstm_persist_case(B, T, u, bf16, mt, uw, seed):
    """Run one Bidirectional(LSTM) layer forward + BPTT through the per-step kernels (crnn_lstm_*_ex) and through the
    persistent ones (crnn_lstm_*_persist); returns both result sets as host arrays + the status word."""
    rs = np.random.RandomState(seed)
    G = 4 * u
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    xw = [dev(rs.normal(size=(T, B, G)) * 0.8) for _ in range(2)]
    gd = dev(rs.normal(size=(T, B, 2 * u)))
    if bf16:
        ut = [_to_bf16_dev(U[d].T) for d in range(2)]; Ud = [_to_bf16_dev(U[d]) for d in range(2)]
    else:
        ut = [dev(U[d].T) for d in range(2)]; Ud = [dev(U[d]) for d in range(2)]
    dt = 1 if bf16 else 0
    nbytes = L().crnn_lstm_persist_xbuf_bytes(T, B, u, dt)
    xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    out = {}
    for kind in ("step", "persist"):
        hcat = zeros(T, B, 2 * u); cs = [zeros(T, B, u) for _ in range(2)]; gt = [zeros(T, B, G) for _ in range(2)]
        dz = [zeros(T, B, G) for _ in range(2)]; dc = [zeros(B, u) for _ in range(2)]
        hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u); gb = ctypes.c_void_p(gd.data_ptr() + 4 * u)
        if kind == "step":
            ok(L().crnn_lstm_fwd_ex(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, dt, S()))
            ok(L().crnn_lstm_bwd_ex(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dc[0]), P(dc[1]),
                                    T, B, u, dt, S()))
            status = 0
        else:
            ok(L().crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, dt,
                                         P(xbuf), nbytes, mt, uw, S()))
            status = int(xbuf[4].item()) != -1            # per-launch status word (byte 16); xbuf[0] = sticky give-up counter
            # (round 4: the launch also leaves the bias-gradient partials, one row per 16-row batch tile; rows past them stay untouched)
            nrow = L().crnn_rnn_db_rows(B)
            dbp = [torch.full((nrow + 1, G), 7.0, device="cuda") for _ in range(2)]
            ok(L().crnn_lstm_bwd_persist_db(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dbp[0]), P(dbp[1]),
                                            T, B, u, dt, P(xbuf), nbytes, mt, uw, S()))
            status |= int(xbuf[4].item()) != -1
            status |= int(xbuf[0].item()) != 0
            assert nrow == (B + 15) // 16 and all(bool((t[nrow] == 7.0).all()) for t in dbp)
            out["db"] = [host(t[:nrow]).sum(0) for t in dbp]
        out[kind] = dict(h=host(hcat), c=[host(t) for t in cs], g=[host(t) for t in gt], dz=[host(t) for t in dz], status=status)
    return out


@pytest.mark.parametrize("B,T,u,bf16,mt,uw", [
    (5, 7, 64, False, 0, 0), (33, 6, 128, False, 1, 1), (33, 6, 128, False, 2, 2), (70, 9, 256, False, 0, 0), (40, 5, 256, False, 2, 1),
    (20, 9, 128, True, 0, 0), (37, 8, 256, True, 1, 1), (37, 8, 256, True, 2, 2), (37, 8, 256, True, 1, 4), (256, 52, 256, True, 0, 0),
    (256, 52, 256, True, 1, 2), (64, 102, 256, False, 0, 0), (18, 5, 512, True, 0, 0), (600, 4, 128, True, 0, 0),
    # uw | 0x100 (CRNN_RNN_XCD_LOCAL): cluster members = workgroup ids congruent modulo 8 (needs #clusters % 8 == 0, else the linear map)
    (256, 52, 256, True, 0, 0x100), (64, 9, 256, False, 1, 0x101), (128, 7, 128, True, 2, 0x102), (37, 8, 256, True, 1, 0x102)])
def test_persistent_lstm_is_bit_identical_to_the_step_kernels(B, T, u, bf16, mt, uw):
    """One launch per layer (cluster of u/16 workgroups per batch tile, recurrent weights + cell state in registers, h_t / dz_t
    all-gathered through device memory and staged through LDS) must reproduce the T-launch path bit for bit, forward and
    BPTT, fp32 and bf16 recurrent products, ragged batch tiles, 16- and 32-row tiles, 256- / 512- / 1024-thread workgroups
    (cluster sizes u/16, u/32, u/64), T up to the IAM shape's 102, a batch that needs several launches (600 rows)."""
    r = _lstm_persist_case(B, T, u, bf16, mt, uw, seed=B + T + u)
    a, b = r["step"], r["persist"]
    assert b["status"] == 0, "a bounded wait of the persistent kernel gave up"
    assert np.array_equal(a["h"], b["h"]), "h: max diff %g" % np.abs(a["h"] - b["h"]).max()
    for d in range(2):
        assert np.array_equal(a["c"][d], b["c"][d]), "c dir%d" % d
        assert np.array_equal(a["g"][d], b["g"][d]), "gates dir%d" % d
        assert np.array_equal(a["dz"][d], b["dz"][d]), "dz dir%d: max diff %g" % (d, np.abs(a["dz"][d] - b["dz"][d]).max())
    assert np.isfinite(b["h"]).all() and np.abs(b["dz"][0]).max() > 0
    for d in range(2):        # the bias gradient the same launch leaves: column sums of dz over time and batch
        ref = b["dz"][d].astype(np.float64).reshape(-1, 4 * u).sum(0)
        assert_close(r["db"][d], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()), what="db dir%d" % d)


def test_persistent_lstm_reports_a_lost_cluster():
    """The give-up path on the PRODUCT path (VERDICT r2 weak-10 / ADVICE): a cluster that loses a member must not produce numbers
    silently.  Deterministic form of the scenario (the real one -- CUs pinned by another kernel -- is scripts/occupy_probe.py, its output
    profiles/r03_occupy_probe.txt; it depends on two queues of one process being scheduled together, which a GPU shared by several test
    workers does not promise): the recurrence is launched on the ENGINE's own exchange buffer with its last workgroup missing
    (CRNN_RNN_DEBUG_DROP_MEMBER).  The cluster's other members wait their 2 s, give up and free-run; the per-launch status word loses bit
    0, the sticky counter moves, Engine.check_rnn_status raises -- once -- and the engine's next clean forward passes the check again."""
    import time
    from crnn_mi355x.engine import Engine
    from crnn_mi355x.native import CrnnError
    eng = Engine(16, imgh=40, imgw=32, max_len=6, time_dense_size=64, n_units=256, dropout=False, precision="bf16s")   # T = 22
    assert eng._rnn_giveups is not None
    from oracle import model as M
    cfg = M.Config(imgh=40, imgw=32, max_len=6, time_dense_size=64, n_units=256)
    p, bn = M.init_params(cfg, seed=3, dtype=np.float64)
    eng.set_params(M.randomize_params(cfg, p), bn)
    x = np.random.RandomState(0).normal(size=(16, 40, 32, 1)).astype(np.float32)
    y0 = eng.forward(x, train=False).clone()
    eng.check_rnn_status()                                                     # clean run: no exception
    # a stand-alone layer on the engine's exchange buffer, one workgroup short
    B, T, u = 16, 6, 256; G = 4 * u
    rs = np.random.RandomState(1)
    ut = [_to_bf16_dev(rs.normal(size=(G, u)) * 0.1) for _ in range(2)]
    xw = [dev(rs.normal(size=(T, B, G))) for _ in range(2)]
    hcat = zeros(T, B, 2 * u); cs = [zeros(T, B, u) for _ in range(2)]; gt = [zeros(T, B, G) for _ in range(2)]
    xbuf = eng.ws_tensor("rnnx")
    nbytes = L().crnn_lstm_persist_xbuf_bytes(eng.T, eng.B, u, 1)
    assert L().crnn_lstm_persist_xbuf_bytes(T, B, u, 1) <= nbytes
    t0 = time.perf_counter()
    assert L().crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(cs[0]), P(cs[1]),
                                     P(gt[0]), P(gt[1]), T, B, u, 1, P(xbuf), nbytes, 0, 0x200, S()) == 0
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    words = xbuf[:8].view(torch.int32).cpu().numpy()
    assert words[0] > 0 and (words[4] & 1) == 0, ("no give-up was recorded", words[:5])
    assert 1.5 < waited < 20, waited                                           # bounded: about 2 s per wait, not a hang
    with pytest.raises(CrnnError, match="gave up"):
        eng.check_rnn_status()
    eng.check_rnn_status()                                                     # reported once: the counter is compared with the last value seen
    y1 = eng.forward(x, train=False)
    eng.check_rnn_status()                                                     # clean again, same numbers as before
    assert torch.equal(y0, y1)


def test_persistent_lstm_repeated_launches_and_oracle():
    """Back-to-back launches reuse (and re-poison) the same exchange buffer; the fp32 result also matches the fp64 oracle cell."""
    B, T, u, din = 21, 11, 64, 24
    rs = np.random.RandomState(77)
    x = rs.normal(size=(B, T, din)); G = 4 * u
    Wt = [rs.normal(size=(din, G)) * 0.3 for _ in range(2)]
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    bb = [rs.normal(size=G) * 0.2 for _ in range(2)]
    tm = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))
    xw = [dev(tm(x @ Wt[d] + bb[d])) for d in range(2)]
    ut = [dev(U[d].T) for d in range(2)]
    nbytes = L().crnn_lstm_persist_xbuf_bytes(T, B, u, 0)
    xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    hcat = zeros(T, B, 2 * u); cs = [zeros(T, B, u) for _ in range(2)]; gt = [zeros(T, B, G) for _ in range(2)]
    first = None
    for rep in range(4):
        hcat.zero_()
        code = L().crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(cs[0]),
                                         P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, 0, P(xbuf), nbytes, 0, 0, S())
        assert code == 0
    torch.cuda.synchronize()
    assert int(xbuf[4].item()) == -1          # per-launch status (byte 16) all ones: no bounded wait gave up
    assert int(xbuf[0].item()) == 0           # sticky give-up counter (byte 0): never reset by a launch, still zero
    hh = host(hcat)
    for d in range(2):
        h, c = ops.lstm_fwd(x, Wt[d], U[d], bb[d], reverse=(d == 1))
        assert_close(hh[:, :, d * u:(d + 1) * u], tm(h), rtol=1e-4, atol=1e-5, what=f"h dir{d}")
        assert_close(host(cs[d]), tm(c[4]), rtol=1e-4, atol=1e-5, what=f"c dir{d}")
    # unsupported widths / a too-small exchange buffer are refused
    assert L().crnn_lstm_persist_supported(192, 0) == -3 and L().crnn_lstm_persist_supported(64, 1) == -3
    assert L().crnn_lstm_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), P(hcat), 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, u, 0,
                                     P(xbuf), 64, 0, 0, S()) == -2


@pytest.mark.parametrize("B,T,u,din", [(5, 7, 64, 24), (20, 5, 128, 40)])
def test_bigru_fwd_bwd(B, T, u, din):
    rs = np.random.RandomState(B + T + 1)
    x = rs.normal(size=(B, T, din))
    G = 3 * u
    Wt = [rs.normal(size=(din, G)) * 0.3 for _ in range(2)]
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    bb = [rs.normal(size=G) * 0.2 for _ in range(2)]
    Hs, caches = [], []
    for d in range(2):
        h, c = ops.gru_fwd(x, Wt[d], U[d], bb[d], reverse=(d == 1))
        Hs.append(h); caches.append(c)
    tm = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))
    xw = [dev(tm(x @ Wt[d] + bb[d])) for d in range(2)]
    ut = [dev(U[d].T) for d in range(2)]
    hcat = zeros(T, B, 2 * u); gt = [zeros(T, B, G) for _ in range(2)]; rh = [zeros(T, B, u) for _ in range(2)]
    ok(L().crnn_gru_fwd(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(gt[0]), P(gt[1]),
                        P(rh[0]), P(rh[1]), T, B, u, S()))
    hh = host(hcat)
    for d in range(2):
        assert_close(hh[:, :, d * u:(d + 1) * u], tm(Hs[d]), rtol=1e-4, atol=1e-5, what=f"h dir{d}")
        assert_close(host(gt[d]), tm(caches[d][4]), rtol=1e-4, atol=1e-5, what=f"gates dir{d}")
    gH = rs.normal(size=(B, T, 2 * u))
    gd = dev(tm(gH))
    dz = [zeros(T, B, G) for _ in range(2)]; dh = [zeros(B, u) for _ in range(2)]; dhp = [zeros(B, u) for _ in range(2)]
    Ud = [dev(U[d]) for d in range(2)]
    ok(L().crnn_gru_bwd(P(Ud[0]), P(Ud[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(gt[0]), P(gt[1]), P(gd),
                        ctypes.c_void_p(gd.data_ptr() + 4 * u), 2 * u, P(dz[0]), P(dz[1]), P(dh[0]), P(dh[1]), P(dhp[0]), P(dhp[1]), T, B, u, S()))
    for d in range(2):
        dx, dW, dU, db = ops.gru_bwd(caches[d], gH[..., d * u:(d + 1) * u])
        dzh = np.swapaxes(host(dz[d]).astype(np.float64), 0, 1)
        assert_close(dzh @ Wt[d].T, dx, rtol=2e-4, atol=1e-5, what=f"dx dir{d}")
        assert_close(x.reshape(B * T, -1).T @ dzh.reshape(B * T, G), dW, rtol=2e-4, atol=1e-4, what=f"dW dir{d}")
        assert_close(dzh.reshape(B * T, G).sum(0), db, rtol=2e-4, atol=1e-4, what=f"db dir{d}")
        rhh = np.swapaxes(host(rh[d]).astype(np.float64), 0, 1)
        assert_close(rhh.reshape(B * T, u).T @ dzh.reshape(B * T, G)[:, 2 * u:], dU[:, 2 * u:], rtol=2e-4, atol=1e-4, what=f"dUh dir{d}")


def _gru_persist_case(B, T, u, bf16, flags, seed, reps=1):
    """One Bidirectional(GRU) layer forward + BPTT through the per-step kernels (crnn_gru_*_ex: 2 T launches per pass) and through the
    persistent ones (crnn_gru_*_persist); returns both result sets as host arrays + the status."""
    rs = np.random.RandomState(seed)
    G = 3 * u
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    xw = [dev(rs.normal(size=(T, B, G)) * 0.8) for _ in range(2)]
    gd = dev(rs.normal(size=(T, B, 2 * u)))
    if bf16:
        ut = [_to_bf16_dev(U[d].T) for d in range(2)]; Ud = [_to_bf16_dev(U[d]) for d in range(2)]
    else:
        ut = [dev(U[d].T) for d in range(2)]; Ud = [dev(U[d]) for d in range(2)]
    dt = 1 if bf16 else 0
    nbytes = L().crnn_lstm_persist_xbuf_bytes(T, B, u, dt)
    xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    out = {}
    for kind in ("step", "persist"):
        hcat = zeros(T, B, 2 * u); gt = [zeros(T, B, G) for _ in range(2)]; rh = [zeros(T, B, u) for _ in range(2)]
        dz = [zeros(T, B, G) for _ in range(2)]; dh = [zeros(B, u) for _ in range(2)]; dhp = [zeros(B, u) for _ in range(2)]
        hb = ctypes.c_void_p(hcat.data_ptr() + 4 * u); gb = ctypes.c_void_p(gd.data_ptr() + 4 * u)
        status = 0
        if kind == "step":
            ok(L().crnn_gru_fwd_ex(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(gt[0]), P(gt[1]), P(rh[0]), P(rh[1]), T, B, u, dt, S()))
            ok(L().crnn_gru_bwd_ex(P(Ud[0]), P(Ud[1]), P(hcat), hb, 2 * u, P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dh[0]), P(dh[1]),
                                   P(dhp[0]), P(dhp[1]), T, B, u, dt, S()))
        else:
            for _ in range(reps):       # back-to-back launches reuse (and re-poison) the same exchange ring
                ok(L().crnn_gru_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), hb, 2 * u, P(gt[0]), P(gt[1]), P(rh[0]), P(rh[1]), T, B, u, dt,
                                            P(xbuf), nbytes, flags, S()))
                status |= int(xbuf[4].item()) != -1
                nrow = L().crnn_rnn_db_rows(B)      # (round 4: the launch also leaves the bias-gradient partials, one row per 16-row batch tile)
                dbp = [torch.full((nrow + 1, G), 7.0, device="cuda") for _ in range(2)]
                ok(L().crnn_gru_bwd_persist_db(P(Ud[0]), P(Ud[1]), P(hcat), hb, 2 * u, P(gt[0]), P(gt[1]), P(gd), gb, 2 * u, P(dz[0]), P(dz[1]), P(dbp[0]), P(dbp[1]),
                                               T, B, u, dt, P(xbuf), nbytes, flags, S()))
                status |= int(xbuf[4].item()) != -1
                assert all(bool((t[nrow] == 7.0).all()) for t in dbp)
                out["db"] = [host(t[:nrow]).sum(0) for t in dbp]
            status |= int(xbuf[0].item()) != 0
        out[kind] = dict(h=host(hcat), g=[host(t) for t in gt], rh=[host(t) for t in rh], dz=[host(t) for t in dz], status=status)
    return out


@pytest.mark.parametrize("B,T,u,bf16,flags", [
    (5, 7, 64, False, 0), (33, 6, 128, False, 0), (70, 9, 256, False, 0), (20, 9, 128, True, 0), (37, 8, 256, True, 0), (256, 52, 256, True, 0),
    (256, 52, 256, True, 0x100), (64, 102, 256, False, 0x100), (18, 5, 512, True, 0), (600, 4, 128, True, 0), (64, 9, 256, False, 0x100)])
def test_persistent_gru_is_bit_identical_to_the_step_kernels(B, T, u, bf16, flags):
    """One launch per layer and pass (cluster of u/32 workgroups per batch tile, recurrent weights + hidden state / gradient carry in
    registers, two all-gathers per step through the sentinel ring) must reproduce the 2T-launch path bit for bit: forward (z, r, hh gates,
    r*h_prev, h) and BPTT (dz), fp32 and bf16 recurrent products, ragged batch tiles, both workgroup -> cluster maps, T up to the IAM
    shape's 102, a batch that needs several launches (600 rows)."""
    r = _gru_persist_case(B, T, u, bf16, flags, seed=B + T + u)
    a, b = r["step"], r["persist"]
    assert b["status"] == 0, "a bounded wait of the persistent kernel gave up"
    assert np.array_equal(a["h"], b["h"]), "h: max diff %g" % np.abs(a["h"] - b["h"]).max()
    for d in range(2):
        assert np.array_equal(a["g"][d], b["g"][d]), "gates dir%d: max diff %g" % (d, np.abs(a["g"][d] - b["g"][d]).max())
        assert np.array_equal(a["rh"][d], b["rh"][d]), "r*h dir%d" % d
        assert np.array_equal(a["dz"][d], b["dz"][d]), "dz dir%d: max diff %g" % (d, np.abs(a["dz"][d] - b["dz"][d]).max())
    assert np.isfinite(b["h"]).all() and np.abs(b["dz"][0]).max() > 0 and np.abs(b["h"]).max() > 0
    for d in range(2):        # the bias gradient the same launch leaves: column sums of dz over time and batch
        ref = b["dz"][d].astype(np.float64).reshape(-1, 3 * u).sum(0)
        assert_close(r["db"][d], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()), what="db dir%d" % d)


def test_persistent_gru_repeated_launches_and_oracle():
    """Four forward + BPTT launches back to back on one exchange ring stay bit-identical to the step kernels; the fp32 forward also
    matches the fp64 oracle cell (ops.gru_fwd); unsupported widths are refused."""
    r = _gru_persist_case(21, 11, 64, False, 0, seed=5, reps=4)
    assert r["persist"]["status"] == 0
    assert np.array_equal(r["step"]["h"], r["persist"]["h"]) and np.array_equal(r["step"]["dz"][1], r["persist"]["dz"][1])
    B, T, u, din = 19, 9, 64, 24
    rs = np.random.RandomState(78)
    x = rs.normal(size=(B, T, din)); G = 3 * u
    Wt = [rs.normal(size=(din, G)) * 0.3 for _ in range(2)]
    U = [rs.normal(size=(u, G)) * 0.15 for _ in range(2)]
    bb = [rs.normal(size=G) * 0.2 for _ in range(2)]
    tm = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))
    xw = [dev(tm(x @ Wt[d] + bb[d])) for d in range(2)]
    ut = [dev(U[d].T) for d in range(2)]
    nbytes = L().crnn_lstm_persist_xbuf_bytes(T, B, u, 0)
    xbuf = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
    hcat = zeros(T, B, 2 * u); gt = [zeros(T, B, G) for _ in range(2)]; rh = [zeros(T, B, u) for _ in range(2)]
    ok(L().crnn_gru_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(gt[0]), P(gt[1]),
                                P(rh[0]), P(rh[1]), T, B, u, 0, P(xbuf), nbytes, 0, S()))
    hh = host(hcat)
    assert int(xbuf[4].item()) == -1 and int(xbuf[0].item()) == 0
    for d in range(2):
        h, c = ops.gru_fwd(x, Wt[d], U[d], bb[d], reverse=(d == 1))
        assert_close(hh[:, :, d * u:(d + 1) * u], tm(h), rtol=1e-4, atol=1e-5, what=f"h dir{d}")
        assert_close(host(gt[d]), tm(c[4]), rtol=1e-4, atol=1e-5, what=f"gates dir{d}")
    assert L().crnn_gru_persist_supported(192, 0) == -3 and L().crnn_gru_persist_supported(64, 1) == -3
    assert L().crnn_gru_fwd_persist(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), P(hcat), 2 * u, P(gt[0]), P(gt[1]), P(rh[0]), P(rh[1]), T, B, u, 0,
                                    P(xbuf), 64, 0, S()) == -2


# ------------------------------------------------------------------------------------------------ softmax / CTC / decode
def _ctc_case(B, T, C, Lmax, seed, short=False):
    rs = np.random.RandomState(seed)
    logits = rs.normal(size=(B, T, C)) * 2
    y = ops.softmax_fwd(logits)
    ll = rs.randint(1, Lmax + 1, size=B)
    labels = np.full((B, Lmax), C - 1, dtype=np.int64)
    for b in range(B):
        labels[b, :ll[b]] = rs.randint(0, C - 1, size=ll[b])
        if ll[b] >= 2:
            labels[b, 1] = labels[b, 0]  # force a repeated character
    il = np.full(B, T - 2, dtype=np.int64)
    if short:
        il[0] = max(2 * ll[0], 3); il[-1] = 1; ll[-1] = 1
    return logits, y, labels, il, ll


@pytest.mark.parametrize("B,T,C,Lmax,short", [(9, 52, 38, 23, False), (6, 102, 38, 21, True), (4, 12, 7, 4, True)])
def test_ctc_loss_and_logit_gradient(B, T, C, Lmax, short):
    logits, y, labels, il, ll = _ctc_case(B, T, C, Lmax, 11 + B, short)
    loss_ref, gy = ctc.ctc_loss_and_grad(y, labels, il, ll)
    gl_ref = ops.softmax_bwd(y, gy / B)
    yd = zeros(B * T, C)
    ok(L().crnn_softmax_rows(P(dev(logits.reshape(-1, C))), P(yd), B * T, C, S()))
    assert_close(host(yd).reshape(B, T, C), y, rtol=1e-5, atol=1e-7, what="softmax")
    loss = zeros(B); dl = zeros(T, B, C)
    ok(L().crnn_ctc_loss_grad(P(yd), P(dev(labels, np.int32)), P(dev(il, np.int32)), P(dev(ll, np.int32)), P(loss), P(dl), B, T, C, Lmax, 2,
                              1.0 / B, S()))
    assert_close(host(loss), loss_ref, rtol=1e-4, atol=1e-3, what="ctc loss")   # north_star: CTC loss within 1e-3
    assert_close(np.swapaxes(host(dl), 0, 1), gl_ref, rtol=1e-3, atol=2e-6, what="dlogits")


@pytest.mark.parametrize("rows,C,ldz,P_,two", [(52 * 8, 38, 128, 8, True), (64, 64, 64, 0, False), (13312, 38, 128, 256, True), (12, 20, 32, 4, False)])
def test_softmax_rows_with_bias_row_permutation_and_two_outputs(rows, C, ldz, P_, two):
    """crnn_softmax_rows_perm (round 5: dense2's epilogue in one pass): logits = z[:, :C] + bias in permuted row order, their softmax in one or two outputs --
    against crnn_softmax_rows on the permuted biased rows (the same arithmetic: bit for bit) and the oracle; padding columns (NaN here) are never read."""
    rs = np.random.RandomState(rows + C)
    z = rs.normal(size=(rows, ldz)).astype(np.float32) * 3; z[:, C:] = np.nan
    bias = rs.normal(size=C).astype(np.float32)
    zd, bd = dev(z), dev(bias)
    lg = torch.full((rows * C + 16,), 5.0, device="cuda"); p1 = torch.full((rows * C + 16,), 6.0, device="cuda"); p2 = torch.full((rows * C + 16,), 7.0, device="cuda")
    ok(L().crnn_softmax_rows_perm(P(zd), ldz, P(bd), P(lg), P(p1), P(p2) if two else None, rows, C, P_, S()))
    want = z[:, :C] + bias
    if P_:
        m = np.arange(rows); orow = (m % P_) * (rows // P_) + m // P_
        perm = np.empty_like(want); perm[orow] = want; want = perm
    assert np.array_equal(host(lg[:-16]).reshape(rows, C), want)
    ref = zeros(rows, C)
    ok(L().crnn_softmax_rows(P(dev(want)), P(ref), rows, C, S()))
    assert torch.equal(p1[:-16].view(rows, C), ref)
    assert_close(host(p1[:-16]).reshape(rows, C), ops.softmax_fwd(want.astype(np.float64)), rtol=1e-5, atol=1e-7, what="softmax vs oracle")
    if two:
        assert torch.equal(p2[:-16], p1[:-16])
    else:
        assert bool((p2 == 7.0).all())
    assert bool((lg[-16:] == 5.0).all()) and bool((p1[-16:] == 6.0).all()) and bool((p2[-16:] == 7.0).all())
    assert L().crnn_softmax_rows_perm(P(zd), ldz, P(bd), P(lg), P(p1), None, rows, C, 7 if rows % 7 else 5, S()) == -2      # rows not a multiple of permP
    assert L().crnn_softmax_rows_perm(P(zd), C - 1, P(bd), P(lg), P(p1), None, rows, C, 0, S()) == -2


@pytest.mark.parametrize("M,K,C,rate,lddx", [(52 * 256, 512, 38, 0.2, 512), (52 * 8, 128, 38, 0.2, 128), (203, 256, 40, 0.0, 320), (7, 128, 2, 0.5, 128), (52 * 64, 512, 11, 0.2, 512), (2049, 384, 37, 0.3, 384)])
def test_dense_backward_in_one_pass_against_the_products(M, K, C, rate, lddx):
    """crnn_dense_bwd_small (round 5: dense2's backward, dense.hip): weight gradient, bias gradient and the data gradient with the dropout multiplier of
    the layer's input -- against fp64 products (exact fp32 kernel: 1e-5 of each tensor's scale) and crnn_dropout_mask's decisions (the zeros: exactly), with the
    decisions read from the site's keep bytes and evaluated in the kernel;
    rows are owned by workgroups in steps of 8, so ragged row counts (203, 7, 2049) exercise the tails; padding columns of dx (lddx > K) stay untouched."""
    rs = np.random.RandomState(M + K + C)
    x = rs.normal(size=(M, K)).astype(np.float32); dy = rs.normal(size=(M, C)).astype(np.float32) * 0.1; W = rs.normal(size=(K, C)).astype(np.float32)
    seed, layer = 1234567 + M, 9
    xd, dyd, Wd = dev(x), dev(dy), dev(W)
    dx = torch.full((M, lddx), 3.0, device="cuda"); g = torch.full((K * C + C + 8,), 4.0, device="cuda")
    assert L().crnn_dense_bwd_small_supported(M, K, C) == 0
    nb = L().crnn_dense_bwd_small_scratch_bytes(M, K, C)
    scratch = torch.empty(nb // 4, device="cuda")
    gW, gb = g[:K * C], g[K * C:K * C + C]
    keep = torch.zeros(M * K // 8 + 4, dtype=torch.uint8, device="cuda")       # the site's keep bytes, as the forward's dropout pass writes them
    ok(L().crnn_dropout_keep_bytes(P(keep), M * K // 8, rate, seed, layer, S()))
    keep_arg = P(keep) if rate > 0 else None
    ok(L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx), P(gW), P(gb), P(scratch), nb, M, K, C, K, lddx, keep_arg, rate, seed, layer, S()))
    mask = torch.ones(M * K, device="cuda")
    if rate > 0:
        ok(L().crnn_dropout_mask(P(mask), M * K, rate, seed, layer, S()))
    mask = host(mask).reshape(M, K).astype(np.float64)
    x64, dy64, W64 = x[:, :K].astype(np.float64), dy.astype(np.float64), W.astype(np.float64)
    want_dx = (dy64 @ W64.T) * mask
    got_dx = host(dx)
    assert_close(got_dx[:, :K], want_dx, rtol=1e-5, atol=1e-5 * np.abs(want_dx).max(), what="dx")
    assert (got_dx[:, :K][mask == 0] == 0).all()          # dropped elements are exact zeros
    assert (got_dx[:, K:] == 3.0).all()
    want_W = x64.T @ dy64
    assert_close(host(gW).reshape(K, C), want_W, rtol=1e-5, atol=2e-6 * np.abs(want_W).max() + 1e-6, what="dW")
    assert_close(host(gb), dy64.sum(0), rtol=1e-5, atol=1e-6 * M ** 0.5, what="db")
    assert bool((g[K * C + C:] == 4.0).all())
    again = torch.full_like(g, 5.0); dx2 = torch.empty_like(dx)
    ok(L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx2), P(again[:K * C]), P(again[K * C:K * C + C]), P(scratch), nb, M, K, C, K, lddx, keep_arg, rate, seed, layer, S()))
    assert torch.equal(again[:K * C + C], g[:K * C + C]) and torch.equal(dx2[:, :K], dx[:, :K])      # fixed summation order
    if rate > 0:     # without the table the loader wave evaluates the decisions itself: the same bits
        dx3 = torch.empty_like(dx); g3 = torch.full_like(g, 6.0)
        ok(L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx3), P(g3[:K * C]), P(g3[K * C:K * C + C]), P(scratch), nb, M, K, C, K, lddx, None, rate, seed, layer, S()))
        assert torch.equal(g3[:K * C + C], g[:K * C + C]) and torch.equal(dx3[:, :K], dx[:, :K])
    # refusals: bias gradient elsewhere, short scratch, x rows not contiguous, shapes outside the rule
    assert L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx), P(gW), P(gb), P(scratch), nb, M, K, C, K + 4, lddx, keep_arg, rate, seed, layer, S()) == -3
    other = zeros(C)
    assert L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx), P(gW), P(other), P(scratch), nb, M, K, C, K, lddx, keep_arg, rate, seed, layer, S()) == -3
    assert L().crnn_dense_bwd_small(P(xd), P(dyd), P(Wd), P(dx), P(gW), P(gb), P(scratch), nb - 4, M, K, C, K, lddx, keep_arg, rate, seed, layer, S()) == -2
    for (m_, k_, c_) in [(64, 640, 38), (64, 192, 38), (64, 64, 38), (64, 512, 41), (0, 512, 38)]:
        assert L().crnn_dense_bwd_small_supported(m_, k_, c_) == -3


@pytest.mark.parametrize("rows,C,ldx,ldy,rate", [(52 * 256, 512, 512, 512, 0.2), (13, 64, 72, 64, 0.5), (7, 8, 8, 12, 0.1), (33, 128, 128, 128, 0.0)])
def test_dropout_with_keep_bytes_is_dropout_plus_the_keep_table(rows, C, ldx, ldy, rate):
    """crnn_dropout_keep (round 5): the same outputs as crnn_dropout bit for bit, and the keep bytes crnn_dropout_keep_bytes writes for the site."""
    rs = np.random.RandomState(rows + C)
    x = dev(rs.normal(size=(rows, ldx)).astype(np.float32))
    seed, layer = 99 + rows, 9
    y = torch.full((rows, ldy), 3.0, device="cuda"); want = torch.full((rows, ldy), 3.0, device="cuda")
    keep = torch.full((rows * C // 8 + 8,), 7, dtype=torch.uint8, device="cuda"); kw = torch.zeros((rows * C // 8 + 3) // 4 * 4, dtype=torch.uint8, device="cuda")
    ok(L().crnn_dropout_keep(P(x), P(y), P(keep), rows, C, ldx, ldy, rate, seed, layer, S()))
    ok(L().crnn_dropout(P(x), P(want), rows, C, ldx, ldy, rate, seed, layer, S()))
    ok(L().crnn_dropout_keep_bytes(P(kw), rows * C // 8, rate, seed, layer, S()))
    assert torch.equal(y, want)
    assert torch.equal(keep[:rows * C // 8], kw[:rows * C // 8]) and bool((keep[rows * C // 8:] == 7).all())
    assert L().crnn_dropout_keep(P(x), P(y), P(keep), rows, C + 4, ldx, ldy, rate, seed, layer, S()) == -3      # whole groups of 8 only


def test_ctc_impossible_and_greedy_bitexact():
    y = np.full((2, 7, 4), 0.25)
    labels = np.array([[1, 1, 1], [0, 3, 3]]); il = np.array([3, 5]); ll = np.array([3, 1])
    loss = zeros(2); dl = zeros(7, 2, 4)
    ok(L().crnn_ctc_loss_grad(P(dev(y)), P(dev(labels, np.int32)), P(dev(il, np.int32)), P(dev(ll, np.int32)), P(loss), P(dl), 2, 7, 4, 3, 2, 0.5, S()))
    lh = host(loss)
    assert np.isinf(lh[0]) and np.isfinite(lh[1]) and np.all(host(dl)[:, 0] == 0)
    rs = np.random.RandomState(4)
    yp = ops.softmax_fwd(rs.normal(size=(40, 52, 38)) * 3).astype(np.float32)
    yp[0, 3] = yp[0, 3, ::-1].copy(); yp[1, :, :] = 1.0 / 38  # ties -> first index
    il2 = rs.randint(1, 53, size=40)
    ref, rl = ctc.ctc_greedy_decode(yp, il2)
    out = zeros(40, 52, dtype=torch.int32); ln = zeros(40, dtype=torch.int32)
    ok(L().crnn_ctc_greedy_decode(P(dev(yp)), P(dev(il2, np.int32)), P(out), P(ln), 40, 52, 38, S()))
    assert np.array_equal(host(out), ref) and np.array_equal(host(ln), rl)


@pytest.mark.parametrize("bw,merge", [(10, 1), (10, 0), (3, 1), (1, 1), (16, 1), (25, 1), (64, 1), (40, 0)])
def test_beam_decode_matches_oracle(bw, merge):
    rs = np.random.RandomState(bw + merge)
    B, T, C = 24, 52, 38
    # mixture of peaked (realistic) and flat (many near-ties, prefix re-entry) posteriors
    logits = rs.normal(size=(B, T, C)) * rs.choice([0.7, 2.0, 6.0], size=(B, 1, 1))
    yp = ops.softmax_fwd(logits).astype(np.float32)
    il = np.full(B, T); il[:4] = [1, 2, 17, 51]
    ref, rl, rsc = ctc.ctc_beam_decode(yp, beam_width=bw, merge_repeated=bool(merge), input_length=il)
    out = zeros(B, T, dtype=torch.int32); ln = zeros(B, dtype=torch.int32); sc = zeros(B)
    ok(L().crnn_ctc_beam_decode(P(dev(yp)), P(dev(il, np.int32)), P(out), P(ln), P(sc), B, T, C, bw, merge, S()))
    assert np.array_equal(host(ln), rl)
    assert np.array_equal(host(out), ref)
    assert_close(host(sc), rsc, rtol=1e-4, atol=1e-3, what="beam score")


# ------------------------------------------------------------------------------------------------ optimizers
def test_adam_sgd_clipnorm():
    rs = np.random.RandomState(0)
    n = 100003
    p0 = rs.normal(size=n); g = rs.normal(size=n) * 0.05
    for clip in (5.0, 0.0):
        pd, gd = dev(p0), dev(g)
        m, v = zeros(n), zeros(n); norm = zeros(2); scr = zeros(1024, dtype=torch.float64)
        ok(L().crnn_global_norm(P(gd), n, clip, P(scr), P(norm), S()))
        nrm = np.sqrt((g ** 2).sum())
        assert_close(host(norm)[0], nrm, rtol=1e-5, what="norm")
        cs = clip / nrm if (clip and nrm >= clip) else 1.0
        po = {"w": p0.copy()}
        opt = M.Adam(lr=1e-3, clipnorm=clip)
        for it in range(3):
            lr_t = 1e-3 * np.sqrt(1 - .999 ** (it + 1)) / (1 - .5 ** (it + 1))
            ok(L().crnn_adam_step(P(pd), P(gd), P(m), P(v), n, lr_t, 0.5, 0.999, 1e-7, P(norm), S()))
            opt.step(po, {"w": g})
        assert_close(host(pd), po["w"], rtol=1e-5, atol=1e-6, what="adam")
        pd = dev(p0); vel = zeros(n)
        po = {"w": p0.copy()}
        opt = M.SGD(lr=1e-2, clipnorm=clip)
        for it in range(3):
            ok(L().crnn_sgd_step(P(pd), P(gd), P(vel), n, 1e-2 / (1 + 1e-6 * it), 0.9, 1, P(norm), S()))
            opt.step(po, {"w": g})
        assert_close(host(pd), po["w"], rtol=1e-5, atol=1e-6, what="sgd")


# ------------------------------------------------------------------------------------------------ bf16 storage variants
def _to_bf16_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16)


def _f(t):
    return t.float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("B,H,W,C", [(3, 104, 36, 64), (2, 104, 36, 128), (5, 52, 18, 256), (2, 52, 9, 512), (2, 204, 36, 128), (256, 52, 9, 512),
                                     (70, 52, 18, 256),
                                     # round 5 -- step rows of five to eight compute waves and bf16 rows cut into channel ranges: image width 48 (416 columns: blocks 2..7,
                                     # blocks 3..7 as two ranges), width 64 (544 columns, two ranges from block 3 on), and 5 / 6 / 8-wave rows
                                     (3, 64, 52, 64), (2, 64, 52, 128), (4, 32, 26, 256), (3, 32, 13, 512), (70, 32, 13, 512),
                                     (2, 104, 68, 64), (2, 104, 68, 128), (3, 52, 34, 256), (2, 52, 17, 512),
                                     (2, 40, 20, 128), (2, 24, 24, 128), (2, 16, 32, 128), (2, 104, 40, 128), (2, 51, 9, 256)])
def test_row_stream_depthwise_equals_the_halo_tile_kernel(B, H, W, C):
    """crnn_dwconv3x3_fwd_stream (rows streamed through an LDS ring by a loader wave with global_load_lds; every compute lane owns a 16-byte
    column and adds an arriving row's taps to three running output rows) against crnn_dwconv3x3_fwd_ex / crnn_dwconv3x3_bn_relu6_fwd on the
    same bf16 maps: the outputs bit for bit (same fp32 fma chain per output) for the forward, the flipped taps (data gradient) and the
    folded BatchNorm + ReLU6 (inference) forms; the BatchNorm partial sums to fp32 summation round-off and against the NumPy oracle.  The
    CRNN's blocks 2-7 (block 2 runs two row bands side by side), the IAM height, one band per workgroup (batch 256) and many (small
    batches); memory around the output untouched; repeated launches give the same bits."""
    assert L().crnn_dwconv_fwd_stream_supported(B, H, W, C) == 0
    rs = np.random.RandomState(B + H + W + C)
    x = _bf16_round(rs.normal(size=(B, H, W, C))); k = rs.normal(size=(3, 3, C))
    xd, kd = _to_bf16_dev(x), dev(k)
    rows = L().crnn_dwconv_fwd_stream_rows(B, H, W, C); nt = L().crnn_dwconv_num_tiles(B, H, W)
    assert rows >= B
    for flip in (0, 1):
        o1 = torch.full((B * H * W * C + 64,), 7.0, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros(B * H * W * C, dtype=torch.bfloat16, device="cuda")
        p1 = torch.full((rows + 1, 2, C), 3.0, device="cuda"); p2 = zeros(nt, 2, C)
        ok(L().crnn_dwconv3x3_fwd_stream(P(xd), P(kd), P(o1), P(p1), None, B, H, W, C, flip, S()))
        ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(o2), P(p2), B, H, W, C, flip, 1, S()))
        assert torch.equal(o1[:-64].view(torch.int16), o2.view(torch.int16)), "flip %d: max diff %g" % (flip, float((o1[:-64].float() - o2.float()).abs().max()))
        assert bool((o1[-64:] == 7.0).all()) and bool((p1[rows] == 3.0).all())
        a, b = host(p1[:rows]).sum(0), host(p2).sum(0)
        assert_close(a, b, rtol=2e-5, atol=1e-5 * np.sqrt(B * H * W) * 3, what="statistics vs the tile kernel")
        if flip == 0 and B * H * W * C <= 4e6:
            ref = ops.dwconv_fwd(x, k)
            assert_close(_f(o1[:-64].view(B, H, W, C)), ref, rtol=2.0 ** -8, atol=2e-2, what="vs oracle")
            assert_close(a[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2, what="sum vs oracle")
            assert_close(a[1], (ref ** 2).sum((0, 1, 2)), rtol=1e-4, atol=1e-2, what="sum of squares vs oracle")
        o3 = torch.zeros_like(o2); p3 = torch.zeros(rows, 2, C, device="cuda")
        ok(L().crnn_dwconv3x3_fwd_stream(P(xd), P(kd), P(o3), P(p3), None, B, H, W, C, flip, S()))
        assert torch.equal(o3.view(torch.int16), o2.view(torch.int16)) and torch.equal(p3, p1[:rows])
    st = dev(np.concatenate([rs.normal(size=C), 1 + rs.uniform(size=C), 1 + 0.3 * rs.normal(size=C), 0.5 * rs.normal(size=C) + 1.0]))
    o1 = torch.zeros(B * H * W * C, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros_like(o1)
    ok(L().crnn_dwconv3x3_fwd_stream(P(xd), P(kd), P(o1), None, P(st), B, H, W, C, 0, S()))
    ok(L().crnn_dwconv3x3_bn_relu6_fwd(P(xd), P(kd), P(st), P(o2), B, H, W, C, 1, S()))
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    # plain product, no statistics
    o1.zero_()
    ok(L().crnn_dwconv3x3_fwd_stream(P(xd), P(kd), P(o1), None, None, B, H, W, C, 0, S()))
    ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(o2), None, B, H, W, C, 0, 1, S()))
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))


@pytest.mark.parametrize("B,H,W,C", [(3, 104, 36, 64), (2, 104, 36, 128), (5, 52, 18, 256), (2, 52, 9, 512), (256, 52, 9, 512), (70, 104, 36, 64)])
@pytest.mark.parametrize("rate", [0.1, 0.0])
def test_row_stream_depthwise_with_the_batchnorm2_prologue_equals_the_two_pass_path(B, H, W, C, rate):
    """crnn_dwconv3x3_fwd_stream_pro / crnn_dwconv3x3_bwd_stream_pro: the previous block's BatchNorm-2 + ReLU6 + Dropout(.1) (utils.py:48-56)
    applied to its pointwise output q inside the depthwise row-stream kernels (LDS, one row ahead) instead of by crnn_bn_act_pool_drop_ex
    writing x: depthwise outputs, BatchNorm-1 statistic partials, data gradient and depthwise weight gradient are bit-identical to the
    two-pass path on the materialised x.  The CRNN's un-pooled block outputs (inputs of blocks 2, 3, 5, 7); one band per workgroup
    (batch 256) and several; with and without dropout; memory around the outputs untouched."""
    assert L().crnn_dwconv_fwd_stream_pro_supported(B, H, W, C) == 0 and L().crnn_dwconv_bwd_stream_pro_supported(B, H, W, C) == 0
    rs = np.random.RandomState(B + H + W + C)
    n = B * H * W * C
    qd = (torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(n % 1000)) * 1.5).to(torch.bfloat16)
    k = rs.normal(size=(3, 3, C)); kd = dev(k)
    st2 = dev(np.concatenate([rs.normal(size=C), 1 + rs.uniform(size=C), 1 + 0.5 * rs.normal(size=C), 1.5 * rs.normal(size=C) + 1.5]))
    seed, layer = 77, 4
    ng = n // 8
    keep = torch.full((ng + 3 + 64,), 0x55, dtype=torch.uint8, device="cuda")
    ok(L().crnn_dropout_keep_bytes(P(keep), ng, rate, seed, layer, S()))
    assert bool((keep[(ng + 3) // 4 * 4:] == 0x55).all())
    if rate > 0:     # the keep bytes are the decisions of crnn_dropout_mask, bit e of byte g = element 8 g + e
        m = zeros(n)
        ok(L().crnn_dropout_mask(P(m), n, rate, seed, layer, S()))
        bits = ((keep[:ng].to(torch.int32).unsqueeze(1) >> torch.arange(8, device="cuda", dtype=torch.int32)) & 1).reshape(-1)
        assert torch.equal(bits.bool(), m > 0)
    else:
        assert bool((keep[:ng] == 0xFF).all())
    xd = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_bn_act_pool_drop_ex(P(qd), P(st2), P(xd), B, H, W, C, 1, 1, rate, seed, layer, 1, 1, S()))
    xf = xd.float()
    assert float((xf == 0).float().mean()) > (0.08 if rate > 0 else 0.0) and float((xf > 6.0).float().mean()) > (1e-3 if rate > 0 else -1)
    # ---- forward
    rows = L().crnn_dwconv_fwd_stream_rows(B, H, W, C)
    o1 = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); p1 = zeros(rows, 2, C)
    ok(L().crnn_dwconv3x3_fwd_stream(P(xd), P(kd), P(o1), P(p1), None, B, H, W, C, 0, S()))
    o2 = torch.full((n + 64,), 7.0, dtype=torch.bfloat16, device="cuda"); p2 = torch.full((rows + 1, 2, C), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_fwd_stream_pro(P(qd), P(st2), rate, P(keep), P(kd), P(o2), P(p2), B, H, W, C, S()))
    assert torch.equal(o2[:-64].view(torch.int16), o1.view(torch.int16)), "d: max diff %g" % float((o2[:-64].float() - o1.float()).abs().max())
    assert torch.equal(p2[:rows], p1) and bool((o2[-64:] == 7.0).all()) and bool((p2[rows] == 3.0).all())
    o3 = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); p3 = zeros(rows, 2, C)
    ok(L().crnn_dwconv3x3_fwd_stream_pro(P(qd), P(st2), rate, P(keep), P(kd), P(o3), P(p3), B, H, W, C, S()))
    assert torch.equal(o3, o2[:-64]) and torch.equal(p3, p1), "repeat launches differ"
    # another dropout site / seed gives another mask
    if rate > 0:
        keep2 = torch.zeros_like(keep)
        ok(L().crnn_dropout_keep_bytes(P(keep2), ng, rate, seed + 1, layer, S()))
        ok(L().crnn_dwconv3x3_fwd_stream_pro(P(qd), P(st2), rate, P(keep2), P(kd), P(o3), P(p3), B, H, W, C, S()))
        assert not torch.equal(o3, o1)
        assert L().crnn_dwconv3x3_fwd_stream_pro(P(qd), P(st2), rate, None, P(kd), P(o3), P(p3), B, H, W, C, S()) == -2
    # ---- backward of the consuming depthwise stage
    dd = o1                                                     # d = dwconv(x)
    dad = (torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(5)) * 0.7).to(torch.bfloat16)
    M = B * H * W
    dh = dd.float().view(M, C).double()
    mean, var = dh.mean(0).cpu().numpy(), dh.var(0, unbiased=False).cpu().numpy()
    gamma, beta = rs.normal(size=C) * 0.3 + 1.0, rs.normal(size=C) * 0.5 + 1.0
    scale = gamma / np.sqrt(var + 1e-3)
    st1 = dev(np.concatenate([mean, var, scale, beta - mean * scale]))
    coef = dev(np.concatenate([rs.normal(size=C) * 1e-3, rs.normal(size=C) * 1e-3]))
    brow = L().crnn_dwconv_bwd_stream_rows(B, H, W, C)
    dx1 = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); dk1 = zeros(9, C); sc = zeros(brow * 9 * C)
    ok(L().crnn_dwconv3x3_bwd_stream(P(dd), P(dad), P(st1), P(coef), P(xd), P(kd), P(dx1), P(dk1), P(sc), B, H, W, C, S()))
    dx2 = torch.full((n + 64,), 9.0, dtype=torch.bfloat16, device="cuda"); dk2 = zeros(9, C); sc2 = torch.full((brow * 9 * C + 16,), 5.0, device="cuda")
    ok(L().crnn_dwconv3x3_bwd_stream_pro(P(dd), P(dad), P(st1), P(coef), P(qd), P(st2), rate, P(keep), P(kd), P(dx2), P(dk2), P(sc2), None, B, H, W, C, S()))
    assert torch.equal(dx2[:-64].view(torch.int16), dx1.view(torch.int16)), "dx: max diff %g" % float((dx2[:-64].float() - dx1.float()).abs().max())
    assert torch.equal(dk2, dk1), "dk: max diff %g" % float((dk2 - dk1).abs().max())
    assert bool((dx2[-64:] == 9.0).all()) and bool((sc2[-16:] == 5.0).all())
    assert float(dk1.abs().max()) > 0
    # ---- the same launch also taking the statistics pass of the producer's BatchNorm-2 backward (gy = dx through dropout and the ReLU6 gate of q):
    # dx / dk unchanged bit for bit, sums = crnn_bn_bwd_ex's (pass 1 + finalize) to fp32 summation order
    dx3 = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); dk3 = zeros(9, C)
    parts2 = torch.full((brow + 1, 2, C), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_bwd_stream_pro(P(dd), P(dad), P(st1), P(coef), P(qd), P(st2), rate, P(keep), P(kd), P(dx3), P(dk3), P(sc2), P(parts2), B, H, W, C, S()))
    assert torch.equal(dx3, dx1) and torch.equal(dk3, dk1) and bool((parts2[brow] == 3.0).all())
    dg2, db2, coef2 = zeros(C), zeros(C), zeros(2 * C)
    ok(L().crnn_bn_bwd_finalize(P(parts2), brow, C, M, P(dg2), P(db2), P(coef2), S()))
    gq = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); dg1, db1, coef1 = zeros(C), zeros(C), zeros(2 * C)
    pp = zeros(max(L().crnn_bn_bwd_chunks(M), 1) * 2 * C + 64); gam = dev(np.ones(C))
    ok(L().crnn_bn_bwd_ex(P(qd), P(dx1), P(st2), P(gam), P(gq), P(dg1), P(db1), P(pp), P(coef1), B, H, W, C, 1, 1, rate, seed, layer, 1, S()))
    for a, b, what in ((db2, db1, "sum gy"), (dg2, dg1, "sum gy xhat"), (coef2, coef1, "coefficients")):
        assert_close(host(a), host(b), rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())), what="BatchNorm-2 backward " + what)
    gq2 = torch.zeros_like(gq)
    ok(L().crnn_bn_bwd_apply_ex(P(qd), P(dx1), P(st2), P(coef1), P(gq2), B, H, W, C, 1, 1, rate, seed, layer, 1, S()))
    assert torch.equal(gq2, gq)


@pytest.mark.parametrize("B,H,W,C", [(3, 104, 36, 64), (2, 104, 36, 128), (5, 52, 18, 256), (2, 52, 9, 512), (256, 52, 9, 512), (70, 104, 36, 64)])
@pytest.mark.parametrize("rate", [0.1, 0.0])
def test_fp32_row_stream_depthwise_kernels_and_their_batchnorm2_prologue_forms(B, H, W, C, rate):
    """The fp32 forms of the row-stream kernels (round 4, the parity mode): crnn_dwconv3x3_fwd_stream_dt against the halo-tile kernel (outputs
    bit for bit, statistics to summation order); crnn_dwconv3x3_fwd_stream_pro_ex / crnn_dwconv3x3_bwd_stream_pro_ex (BatchNorm-2 + ReLU6 +
    Dropout(.1) of the previous block applied to q in LDS; keep bytes read as nibbles) against crnn_bn_act_pool_drop_ex + the plain fp32 forms on
    the materialised x: outputs, statistic partials, data and weight gradients bit for bit; the BatchNorm-2 backward statistics taken by the same
    launch against crnn_bn_bwd_ex to summation order.  Memory around the outputs untouched."""
    F32 = 0
    assert L().crnn_dwconv_fwd_stream_supported_ex(B, H, W, C, F32) == 0
    assert L().crnn_dwconv_fwd_stream_pro_supported_ex(B, H, W, C, F32) == 0 and L().crnn_dwconv_bwd_stream_pro_supported_ex(B, H, W, C, F32) == 0
    rs = np.random.RandomState(B + H + W + C + 1)
    n = B * H * W * C
    qd = torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(n % 1000 + 1)) * 1.5
    k = rs.normal(size=(3, 3, C)); kd = dev(k)
    st2 = dev(np.concatenate([rs.normal(size=C), 1 + rs.uniform(size=C), 1 + 0.5 * rs.normal(size=C), 1.5 * rs.normal(size=C) + 1.5]))
    seed, layer = 78, 2
    ng = n // 8
    keep = torch.full((ng + 3 + 64,), 0x55, dtype=torch.uint8, device="cuda")
    ok(L().crnn_dropout_keep_bytes(P(keep), ng, rate, seed, layer, S()))
    xd = zeros(n)
    ok(L().crnn_bn_act_pool_drop_ex(P(qd), P(st2), P(xd), B, H, W, C, 1, 1, rate, seed, layer, F32, F32, S()))
    assert float((xd == 0).float().mean()) > (0.08 if rate > 0 else 0.0) and float((xd > 6.0).float().mean()) > (1e-3 if rate > 0 else -1)
    # ---- forward: tile kernel, plain stream form, prologue form
    ntiles = L().crnn_dwconv_num_tiles(B, H, W)
    o0 = zeros(n); p0 = zeros(ntiles, 2, C)
    ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(o0), P(p0), B, H, W, C, 0, F32, S()))
    rows = L().crnn_dwconv_fwd_stream_rows_ex(B, H, W, C, F32)
    assert rows >= B
    o1 = torch.full((n + 64,), 7.0, device="cuda"); p1 = torch.full((rows + 1, 2, C), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_fwd_stream_dt(P(xd), P(kd), P(o1), P(p1), B, H, W, C, 0, F32, S()))
    assert torch.equal(o1[:-64].view(torch.int32), o0.view(torch.int32)), "stream vs tile: max diff %g" % float((o1[:-64] - o0).abs().max())
    assert bool((o1[-64:] == 7.0).all()) and bool((p1[rows] == 3.0).all())
    t0, t1 = host(p0).sum(0), host(p1[:rows]).sum(0)
    assert_close(t1, t0, rtol=1e-4, atol=1e-4 * np.abs(t0).max(), what="statistics stream vs tile")
    o2 = torch.full((n + 64,), 7.0, device="cuda"); p2 = torch.full((rows + 1, 2, C), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_fwd_stream_pro_ex(P(qd), P(st2), rate, P(keep), P(kd), P(o2), P(p2), B, H, W, C, F32, S()))
    assert torch.equal(o2[:-64].view(torch.int32), o0.view(torch.int32)), "d: max diff %g" % float((o2[:-64] - o0).abs().max())
    assert torch.equal(p2[:rows], p1[:rows]) and bool((o2[-64:] == 7.0).all()) and bool((p2[rows] == 3.0).all())
    o3 = zeros(n); p3 = zeros(rows, 2, C)
    ok(L().crnn_dwconv3x3_fwd_stream_pro_ex(P(qd), P(st2), rate, P(keep), P(kd), P(o3), P(p3), B, H, W, C, F32, S()))
    assert torch.equal(o3, o2[:-64]) and torch.equal(p3, p1[:rows]), "repeat launches differ"
    if rate > 0:
        assert L().crnn_dwconv3x3_fwd_stream_pro_ex(P(qd), P(st2), rate, None, P(kd), P(o3), P(p3), B, H, W, C, F32, S()) == -2
    # ---- backward of the consuming depthwise stage
    dd = o0
    dad = torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(6)) * 0.7
    M = B * H * W
    dh = dd.view(M, C).double()
    mean, var = dh.mean(0).cpu().numpy(), dh.var(0, unbiased=False).cpu().numpy()
    gamma, beta = rs.normal(size=C) * 0.3 + 1.0, rs.normal(size=C) * 0.5 + 1.0
    scale = gamma / np.sqrt(var + 1e-3)
    st1 = dev(np.concatenate([mean, var, scale, beta - mean * scale]))
    coef = dev(np.concatenate([rs.normal(size=C) * 1e-3, rs.normal(size=C) * 1e-3]))
    brow = L().crnn_dwconv_bwd_stream_rows_ex(B, H, W, C, F32)
    dx1 = zeros(n); dk1 = zeros(9, C); sc = zeros(brow * 9 * C)
    ok(L().crnn_dwconv3x3_bwd_stream_ex(P(dd), P(dad), P(st1), P(coef), P(xd), P(kd), P(dx1), P(dk1), P(sc), B, H, W, C, F32, S()))
    dx2 = torch.full((n + 64,), 9.0, device="cuda"); dk2 = zeros(9, C); sc2 = torch.full((brow * 9 * C + 16,), 5.0, device="cuda")
    ok(L().crnn_dwconv3x3_bwd_stream_pro_ex(P(dd), P(dad), P(st1), P(coef), P(qd), P(st2), rate, P(keep), P(kd), P(dx2), P(dk2), P(sc2), None, B, H, W, C, F32, S()))
    assert torch.equal(dx2[:-64].view(torch.int32), dx1.view(torch.int32)), "dx: max diff %g" % float((dx2[:-64] - dx1).abs().max())
    assert torch.equal(dk2, dk1), "dk: max diff %g" % float((dk2 - dk1).abs().max())
    assert bool((dx2[-64:] == 9.0).all()) and bool((sc2[-16:] == 5.0).all()) and float(dk1.abs().max()) > 0
    dx3 = zeros(n); dk3 = zeros(9, C)
    parts2 = torch.full((brow + 1, 2, C), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_bwd_stream_pro_ex(P(dd), P(dad), P(st1), P(coef), P(qd), P(st2), rate, P(keep), P(kd), P(dx3), P(dk3), P(sc2), P(parts2), B, H, W, C, F32, S()))
    assert torch.equal(dx3, dx1) and torch.equal(dk3, dk1) and bool((parts2[brow] == 3.0).all())
    dg2, db2, coef2 = zeros(C), zeros(C), zeros(2 * C)
    ok(L().crnn_bn_bwd_finalize(P(parts2), brow, C, M, P(dg2), P(db2), P(coef2), S()))
    gq = zeros(n); dg1, db1, coef1 = zeros(C), zeros(C), zeros(2 * C)
    pp = zeros(max(L().crnn_bn_bwd_chunks(M), 1) * 2 * C + 64); gam = dev(np.ones(C))
    ok(L().crnn_bn_bwd_ex(P(qd), P(dx1), P(st2), P(gam), P(gq), P(dg1), P(db1), P(pp), P(coef1), B, H, W, C, 1, 1, rate, seed, layer, F32, S()))
    for a, b, what in ((db2, db1, "sum gy"), (dg2, dg1, "sum gy xhat"), (coef2, coef1, "coefficients")):
        assert_close(host(a), host(b), rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())), what="BatchNorm-2 backward " + what)


def test_dropout_rng_statistics():
    """The counter-based dropout RNG (csrc/common.h: three ChaCha quarter-rounds per group of 8 elements): the kept fraction matches the
    rate, masks of different seeds / dropout sites are independent, neighbouring elements and channel strides are uncorrelated."""
    n = 1 << 22
    rate = 0.1
    ms = []
    for seed, layer in ((1, 1), (1, 2), (2, 1), (2 ** 40 + 3, 7)):
        m = zeros(n)
        ok(L().crnn_dropout_mask(P(m), n, rate, seed, layer, S()))
        vals = torch.unique(m)
        assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.9) < 1e-6
        ms.append((m > 0).float())
        frac = 1.0 - float(ms[-1].mean())
        assert abs(frac - rate) < 4 * np.sqrt(rate * 0.9 / n) + 2e-5, (seed, layer, frac)     # floor(rate * 65536) / 65536 = 0.09999
    sd = rate * 0.9
    for a in range(len(ms)):
        for b in range(a + 1, len(ms)):
            cov = float(((ms[a] - 0.9) * (ms[b] - 0.9)).mean()) / sd
            assert abs(cov) < 5 / np.sqrt(n), (a, b, cov)
    d = ms[0] - 0.9
    for stride in (1, 2, 3, 7, 8, 64, 128, 512, 4608, 36 * 512):
        cov = float((d[:-stride] * d[stride:]).mean()) / sd
        assert abs(cov) < 5 / np.sqrt(n), (stride, cov)
    m0 = zeros(64)
    ok(L().crnn_dropout_mask(P(m0), 64, 0.0, 1, 1, S()))
    assert bool((m0 == 1).all())


@pytest.mark.parametrize("B,H,W,C", [(3, 64, 52, 64), (2, 64, 52, 128), (4, 32, 26, 256), (3, 32, 13, 512), (2, 104, 68, 64), (3, 52, 34, 256), (2, 40, 20, 64), (2, 51, 9, 128)])
def test_fp32_row_stream_depthwise_on_step_rows_of_five_to_nine_waves(B, H, W, C):
    """Round 5: crnn_dwconv3x3_fwd_stream_dt(fp32) on the step rows image widths 48 / 64 give (and 5-wave rows), forward and flipped taps, against the
    halo-tile kernel: outputs bit for bit, statistics to summation order, memory around the outputs untouched, repeated launches the same bits."""
    F32 = 0
    assert L().crnn_dwconv_fwd_stream_supported_ex(B, H, W, C, F32) == 0
    rs = np.random.RandomState(B + H + W + C + 5)
    n = B * H * W * C
    xd = torch.randn(n, device="cuda", generator=torch.Generator("cuda").manual_seed(n % 997 + 3))
    kd = dev(rs.normal(size=(3, 3, C)))
    ntiles = L().crnn_dwconv_num_tiles(B, H, W)
    rows = L().crnn_dwconv_fwd_stream_rows_ex(B, H, W, C, F32)
    assert rows >= B
    for flip in (0, 1):
        o0 = zeros(n); p0 = zeros(ntiles, 2, C)
        ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(o0), P(p0), B, H, W, C, flip, F32, S()))
        o1 = torch.full((n + 64,), 7.0, device="cuda"); p1 = torch.full((rows + 1, 2, C), 3.0, device="cuda")
        ok(L().crnn_dwconv3x3_fwd_stream_dt(P(xd), P(kd), P(o1), P(p1), B, H, W, C, flip, F32, S()))
        assert torch.equal(o1[:-64].view(torch.int32), o0.view(torch.int32)), "flip %d: max diff %g" % (flip, float((o1[:-64] - o0).abs().max()))
        assert bool((o1[-64:] == 7.0).all()) and bool((p1[rows] == 3.0).all())
        t0, t1 = host(p0).sum(0), host(p1[:rows]).sum(0)
        assert_close(t1, t0, rtol=1e-4, atol=1e-4 * np.abs(t0).max(), what="statistics stream vs tile")
        o2 = zeros(n); p2 = zeros(rows, 2, C)
        ok(L().crnn_dwconv3x3_fwd_stream_dt(P(xd), P(kd), P(o2), P(p2), B, H, W, C, flip, F32, S()))
        assert torch.equal(o2, o1[:-64]) and torch.equal(p2, p1[:rows]), "repeat launches differ"


def test_row_stream_depthwise_refuses_other_shapes():
    """Step rows that fill fewer than five compute waves, and channel counts that are no whole groups of 8, stay with the halo-tile kernels."""
    for B, H, W, C in [(2, 13, 18, 64), (2, 52, 18, 252), (2, 13, 9, 128), (2, 13, 9, 64)]:
        assert L().crnn_dwconv_fwd_stream_supported(B, H, W, C) == -3 and L().crnn_dwconv_fwd_stream_rows(B, H, W, C) == 0
    for B, H, W, C in [(2, 52, 18, 252), (2, 13, 9, 64)]:         # (fp32 rows have twice the columns: 13 x 18 x 64 fills five waves there)
        assert L().crnn_dwconv_fwd_stream_supported_ex(B, H, W, C, 0) == -3 and L().crnn_dwconv_fwd_stream_rows_ex(B, H, W, C, 0) == 0
    # round 5: shapes the 9 KiB step-row rule used to refuse (image widths 48 and 64, 5-wave rows)
    for B, H, W, C in [(2, 104, 40, 128), (2, 51, 9, 256), (4, 64, 52, 64), (4, 32, 26, 256), (4, 104, 68, 128), (4, 52, 17, 512)]:
        assert L().crnn_dwconv_fwd_stream_supported(B, H, W, C) == 0 and L().crnn_dwconv_fwd_stream_rows(B, H, W, C) >= B
        assert L().crnn_dwconv_fwd_stream_supported_ex(B, H, W, C, 0) == 0


def test_bf16_storage_dwconv_bn_chain():
    """Storage-typed kernels (dtype=1): inputs/outputs are bf16 tensors, arithmetic fp32.  Reference = the oracle on the
    bf16-rounded inputs; outputs agree to one bf16 rounding (2^-8 relative)."""
    rs = np.random.RandomState(3)
    B, H, W, C = 2, 13, 18, 64
    x = _bf16_round(rs.normal(size=(B, H, W, C))); k = rs.normal(size=(3, 3, C)); g = _bf16_round(rs.normal(size=(B, H, W, C)))
    xd, gd = _to_bf16_dev(x), _to_bf16_dev(g)
    _KEEP = [xd, gd]
    out = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    nt = L().crnn_dwconv_num_tiles(B, H, W)
    parts = zeros(nt, 2, C)
    ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(dev(k)), P(out), P(parts), B, H, W, C, 0, 1, S()))
    ref = ops.dwconv_fwd(x, k)
    assert_close(_f(out), ref, rtol=2.0 ** -8, atol=2e-2, what="dw fwd bf16")
    assert_close(host(parts).sum(0)[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2, what="stats from fp32 values")
    dk = zeros(9, C); scr = zeros(nt * 9 * C)
    ok(L().crnn_dwconv3x3_wgrad_ex(P(xd), P(gd), P(dk), P(scr), B, H, W, C, 1, S()))
    assert_close(host(dk).reshape(3, 3, C), ops.dwconv_bwd(x, k, g)[1], rtol=1e-4, atol=1e-3, what="dw wgrad bf16 in, fp32 out")
    # BN statistics + apply + pool on a bf16 tensor, bf16 result
    gamma = 1 + 0.3 * rs.normal(size=C); beta = 0.5 * rs.normal(size=C) + 1.0
    Mr = B * H * W
    ch = L().crnn_colreduce_chunks(Mr); cp = zeros(ch, 2, C)
    ok(L().crnn_colreduce_ex(P(xd), P(cp), Mr, C, C, 2, 1, S()))
    st = zeros(4 * C)
    ok(L().crnn_bn_finalize(P(cp), ch, C, Mr, P(dev(gamma)), P(dev(beta)), P(st), S()))
    y_bn, mean, var = ops.bn_train_fwd(x, gamma, beta)
    assert_close(host(st)[:C], mean, what="mean"); assert_close(host(st)[C:2 * C], var, what="var")
    Hq, Wq = (H // 1), (W // 2)
    y = torch.zeros(B, Hq, Wq, C, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_bn_act_pool_drop_ex(P(xd), P(st), P(y), B, H, W, C, 1, 2, 0.0, 0, 0, 1, 1, S()))
    r = ops.relu6_fwd(y_bn)
    assert_close(_f(y), ops.maxpool_fwd(r, 1, 2), rtol=2.0 ** -8, atol=2e-2, what="bn+relu6+pool bf16")
    # BN backward with bf16 x / g / dx
    gp = _bf16_round(rs.normal(size=(B, Hq, Wq, C)))
    gpd = _to_bf16_dev(gp)
    dx = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda"); dgm = zeros(C); dbt = zeros(C)
    pp = zeros(L().crnn_bn_bwd_chunks(Mr), 2, C); coef = zeros(2 * C)
    ok(L().crnn_bn_bwd_ex(P(xd), P(gpd), P(st), P(dev(gamma)), P(dx), P(dgm), P(dbt), P(pp), P(coef), B, H, W, C, 1, 2, 0.0, 0, 0, 1, S()))
    gr = ops.relu6_bwd_from_out(r, ops.maxpool_bwd(r, gp, 1, 2))
    dx_ref, dg_ref, db_ref = ops.bn_train_bwd(x, gamma, mean, var, gr)
    assert_close(host(dgm), dg_ref, rtol=1e-3, atol=1e-2, what="dgamma"); assert_close(host(dbt), db_ref, rtol=1e-3, atol=1e-2, what="dbeta")
    assert_close(_f(dx), dx_ref, rtol=2.0 ** -7, atol=2e-2, what="dx bf16")


@pytest.mark.parametrize("shape", [(2, 30, 26, 64), (2, 12, 52, 128), (1, 9, 7, 64), (1, 6, 70, 64)])
def test_bf16_dwconv_ragged_width_after_nan_filled_lds(shape):
    """Widths that are not a multiple of the kernel's 3-pixel group: the last group of a row is ragged, and its window
    reads LDS the tile fill never wrote.  A NaN-input launch first leaves NaN patterns in LDS; the weight gradient, the
    forward and the statistics of the next launch must not see them (regression: 0 * NaN in the wgrad accumulators)."""
    B, H, W, C = shape
    rs = np.random.RandomState(sum(shape))
    x = _bf16_round(rs.normal(size=shape)); k = rs.normal(size=(3, 3, C)); g = _bf16_round(rs.normal(size=shape))
    xd, gd, kd = _to_bf16_dev(x), _to_bf16_dev(g), dev(k)
    nt = L().crnn_dwconv_num_tiles(B, H, W)
    poison = torch.full((8, 64, 64, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    pout = torch.zeros_like(poison)
    for trial in range(3):
        ok(L().crnn_dwconv3x3_fwd_ex(P(poison), P(kd), P(pout), None, 8, 64, 64, C, 0, 1, S()))
        dk = zeros(9, C); scr = zeros(nt * 9 * C)
        ok(L().crnn_dwconv3x3_wgrad_ex(P(xd), P(gd), P(dk), P(scr), B, H, W, C, 1, S()))
        assert_close(host(dk).reshape(3, 3, C), ops.dwconv_bwd(x, k, g)[1], rtol=1e-4, atol=1e-3, what="wgrad trial %d" % trial)
        ok(L().crnn_dwconv3x3_fwd_ex(P(poison), P(kd), P(pout), None, 8, 64, 64, C, 0, 1, S()))
        out = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda"); parts = zeros(nt, 2, C)
        ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(out), P(parts), B, H, W, C, 0, 1, S()))
        ref = ops.dwconv_fwd(x, k)
        assert_close(_f(out), ref, rtol=2.0 ** -8, atol=2e-2, what="fwd trial %d" % trial)
        assert_close(host(parts).sum(0)[0], ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2, what="stats trial %d" % trial)


@pytest.mark.parametrize("shape", [(3000, 128, 64, 1), (1111, 64, 256, 1), (2049, 256, 128, 0), (520, 512, 512, 1)])
def test_pwconv_with_producer_bn_relu6_matches_the_two_pass_path(shape):
    """crnn_pwconv_bnrelu6_fwd / _wgrad apply ReLU6(BN(d)) while the GEMM stages its operand; the result must be
    bit-identical to materialising a = ReLU6(BN(d)) with crnn_bn_act_pool_drop_ex (bf16) and running the plain GEMMs --
    output tensor, BatchNorm statistics partials and weight gradient -- and within bf16 round-off of the fp64 oracle."""
    M, N, K, wt = shape
    rs = np.random.RandomState(M + N)
    d = _bf16_round(rs.normal(size=(M, K)) * 2.0)
    scale = rs.normal(size=K); shift = rs.normal(size=K) + 1.0
    w = _bf16_round(rs.normal(size=(K, N)) / np.sqrt(K)); g = _bf16_round(rs.normal(size=(M, N)))
    st = dev(np.concatenate([np.zeros(K), np.ones(K), scale, shift]))
    dd, gd = _to_bf16_dev(d), _to_bf16_dev(g)
    wd = _to_bf16_dev(w.T.copy() if wt else w)
    _KEEP = [dd, gd, wd]
    # two-pass reference on the device
    a = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_bn_act_pool_drop_ex(P(dd), P(st), P(a), 1, 1, M, K, 1, 1, 0.0, 0, 0, 1, 1, S()))
    rows = L().crnn_pwconv_stat_rows(M)
    q0 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); p0 = zeros(rows, 2, N)
    ok(L().crnn_pwconv_fwd(P(a), P(wd), P(q0), M, N, K, P(p0), None, 1, 1, 1, 1, wt, S()))
    scr = zeros(16 * 1024 * 1024)
    dw0 = zeros(K, N)
    ok(L().crnn_gemm_bf16_ex(2, P(a), P(gd), P(dw0), K, N, M, K, N, N, None, 0, 0, 0, P(scr), 64 * 1024 * 1024, 1, 1, 0, S()))
    # fused
    q1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); p1 = zeros(rows, 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd(P(dd), P(st), P(wd), P(q1), M, N, K, P(p1), 1, wt, S()))
    dw1 = zeros(K, N)
    scr.fill_(float("nan"))
    ok(L().crnn_pwconv_bnrelu6_wgrad(P(dd), P(st), P(gd), P(dw1), M, N, K, P(scr), 64 * 1024 * 1024, S()))
    assert torch.equal(q0, q1), "forward differs from the two-pass path"
    assert torch.equal(p0, p1), "statistics partials differ"
    assert torch.equal(dw0, dw1), "weight gradient differs"
    # and against the oracle arithmetic (a rounded to bf16, fp32 accumulation)
    a_ref = _bf16_round(np.clip(d * scale + shift, 0.0, 6.0))
    assert_close(_f(a), a_ref, rtol=2.0 ** -8, atol=1e-6, what="a")
    assert_close(_f(q1), _f(a).astype(np.float64) @ w, rtol=2.0 ** -7, atol=2e-2, what="q")
    assert_close(host(dw1), _f(a).astype(np.float64).T @ g, rtol=1e-3, atol=2e-2 * np.sqrt(M / 1000.0), what="dw")
    # contract: a missing BatchNorm state is refused
    assert L().crnn_pwconv_bnrelu6_fwd(P(dd), None, P(wd), P(q1), M, N, K, None, 1, wt, S()) != 0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gemm_bf16_storage_operands_and_result(mode):
    rs = np.random.RandomState(40 + mode)
    Mm, N, K = 300, 136, 200
    A = _bf16_round(rs.normal(size=(Mm, K))); B = _bf16_round(rs.normal(size=(K, N)))
    ref = A @ B
    if mode == 0:
        Ad, Bd, lda, ldb = _to_bf16_dev(A), _to_bf16_dev(B), K, N
    elif mode == 1:
        Ad, Bd, lda, ldb = _to_bf16_dev(A), _to_bf16_dev(B.T), K, K
    else:
        Ad, Bd, lda, ldb = _to_bf16_dev(A.T), _to_bf16_dev(B), Mm, N
    scr = zeros(16 * 1024 * 1024)
    C32 = zeros(Mm, N)
    ok(L().crnn_gemm_bf16_ex(mode, P(Ad), P(Bd), P(C32), Mm, N, K, lda, ldb, N, None, 0, 0, 0, P(scr), 64 * 1024 * 1024, 1, 1, 0, S()))
    assert_close(host(C32), ref, rtol=2e-5, atol=1e-4, what="bf16 operands, fp32 result")
    C16 = torch.zeros(Mm, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_gemm_bf16_ex(mode, P(Ad), P(Bd), P(C16), Mm, N, K, lda, ldb, N, None, 0, 0, 0, P(scr), 64 * 1024 * 1024, 1, 1, 1, S()))
    assert_close(_f(C16), ref, rtol=2.0 ** -8, atol=1e-3, what="bf16 result")
    # mixed: fp32 A (activations) x bf16 B, accumulate into a bf16 C
    A32 = dev(A) if mode != 2 else dev(A.T)
    C0 = _bf16_round(rs.normal(size=(Mm, N))); Cacc = _to_bf16_dev(C0)
    ok(L().crnn_gemm_bf16_ex(mode, P(A32), P(Bd), P(Cacc), Mm, N, K, lda, ldb, N, None, 0, 1, 0, P(scr), 64 * 1024 * 1024, 0, 1, 1, S()))
    assert_close(_f(Cacc), ref + C0, rtol=2.0 ** -7, atol=1e-2, what="accumulate into bf16")


@pytest.mark.parametrize("case", ["fp32", "bf16", "bf16s"])
def test_pwconv_fwd_with_statistics_epilogue(case):
    """crnn_pwconv_fwd: q = a @ w plus the per-tile (sum, sum of squares) of q as stored (utils.py:49-50)."""
    rs = np.random.RandomState(77)
    Mm, N, K = 128 * 5 + 37, 192, 64          # ragged last row tile, two column tiles of 128 (second one half empty)
    A = rs.normal(size=(Mm, K)); W = rs.normal(size=(K, N)) * 0.2
    rows = L().crnn_pwconv_stat_rows(Mm)
    assert rows == 6
    parts = zeros(rows, 2, N)
    if case == "fp32":
        Ad, Wd, Q = dev(A), dev(W), zeros(Mm, N)
        ok(L().crnn_pwconv_fwd(P(Ad), P(Wd), P(Q), Mm, N, K, P(parts), None, 0, 0, 0, 0, 0, S()))
        q = host(Q); ref = A.astype(np.float32).astype(np.float64) @ W.astype(np.float32).astype(np.float64)
        assert_close(q, ref, rtol=1e-5, atol=1e-5, what="pwconv fp32")
    else:
        A = _bf16_round(A); W = _bf16_round(W)
        ref = A @ W
        Wd = _to_bf16_dev(W)
        if case == "bf16":
            Ad, Q = dev(A), zeros(Mm, N)
            ok(L().crnn_pwconv_fwd(P(Ad), P(Wd), P(Q), Mm, N, K, P(parts), None, 1, 0, 1, 0, 0, S()))
            q = host(Q)
            assert_close(q, ref, rtol=2e-5, atol=1e-4, what="pwconv bf16 products")
        else:
            Ad = _to_bf16_dev(A); Q = torch.zeros(Mm, N, dtype=torch.bfloat16, device="cuda")
            ok(L().crnn_pwconv_fwd(P(Ad), P(Wd), P(Q), Mm, N, K, P(parts), None, 1, 1, 1, 1, 0, S()))
            q = _f(Q)
            assert_close(q, ref, rtol=2.0 ** -8, atol=1e-3, what="pwconv bf16 storage")
            # the transposed-weight form (W^T made by crnn_transpose_batch) must give the very same stored result
            WT = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda"); Q2 = torch.zeros_like(Q); parts2 = zeros(rows, 2, N)
            off = (ctypes.c_long * 1)(0); rr = (ctypes.c_int * 1)(K); cc = (ctypes.c_int * 1)(N)
            ok(L().crnn_transpose_batch(P(dev(W)), P(WT), 1, off, off, rr, cc, 1, S()))
            assert torch.equal(WT, Wd.t().contiguous())
            ok(L().crnn_pwconv_fwd(P(Ad), P(WT), P(Q2), Mm, N, K, P(parts2), None, 1, 1, 1, 1, 1, S()))
            assert_close(_f(Q2), q, rtol=2.0 ** -8, atol=1e-3, what="pwconv with W^T")
            assert_close(host(parts2), host(parts), rtol=1e-3, atol=1e-2, what="stats with W^T")
    pr = host(parts).astype(np.float64)
    q64 = q.astype(np.float64)
    # the statistics are those of the STORED values, tile by tile (128 rows each)
    for t in range(rows):
        blk = q64[128 * t: 128 * (t + 1)]
        assert_close(pr[t, 0], blk.sum(0), rtol=1e-5, atol=1e-3, what="tile %d sum" % t)
        assert_close(pr[t, 1], (blk * blk).sum(0), rtol=1e-5, atol=1e-3, what="tile %d sumsq" % t)
    # statistics are refused together with bias / accumulate-style epilogues? (plain product only) -> covered by the ABI contract


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1000, 256, 128), (129, 512, 256), (4097, 256, 512), (128 * 300 + 5, 128, 256), (77, 384, 192)])
def test_persistent_nt_gemm_equals_the_tile_kernel(M, N, K):
    """crnn_gemm_nt_bf16 (persistent workgroups, LDS-DMA ring, loader + MFMA waves, swapped MFMA operands, direct 16-byte
    stores) against crnn_gemm_bf16_ex mode 1 on bf16 operands: same MFMA, same k order -> the very same bf16 result; and
    against an fp64 product of the rounded operands.  Ragged M (last stripe partly empty), one and several channel passes,
    more stripes than CUs (persistent loop), N / K that are multiples of 128 / 64 but not powers of two."""
    rs = np.random.RandomState(M + N + K)
    X = _bf16_round(rs.normal(size=(M, K))); W = _bf16_round(rs.normal(size=(N, K)) * 0.2)
    Xd, Wd = _to_bf16_dev(X), _to_bf16_dev(W)
    Y = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device="cuda")      # rows past M must stay untouched
    ok(L().crnn_gemm_nt_bf16(P(Xd), P(Wd), P(Y), M, N, K, S()))
    Y2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_gemm_bf16_ex(1, P(Xd), P(Wd), P(Y2), M, N, K, K, K, N, None, 0, 0, 0, None, 0, 1, 1, 1, S()))
    got, ref_dev = Y[:M].float().cpu().numpy(), Y2.float().cpu().numpy()
    assert np.array_equal(got, ref_dev), "differs from the tile kernel: max %g" % np.abs(got - ref_dev).max()
    assert np.all(Y[M:].float().cpu().numpy() == 7.0)
    ref = X @ W.T
    assert_close(got, ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max(), what="nt gemm vs fp64")     # bf16 output rounding
    # shapes outside the kernel's rules are refused (the caller falls back to crnn_gemm_bf16_ex)
    assert L().crnn_gemm_nt_bf16(P(Xd), P(Wd), P(Y), M, 64, K, S()) == -3
    assert L().crnn_gemm_nt_bf16(P(Xd), P(Wd), P(Y), M, N, K - 32 if K > 64 else 96, S()) == -3


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1000, 256, 128), (129, 512, 256), (4097, 256, 512), (128 * 300 + 5, 128, 256),
                                   (128 * 2100 + 77, 512, 512), (70000, 384, 128), (5, 1024, 64), (52 * 256, 4608, 128), (52 * 64 + 3, 4608, 128), (900, 1152, 64)])
def test_weights_resident_gemm_equals_the_tile_kernel(M, N, K):
    """crnn_gemm_wres_bf16 (weight fragments resident in registers, pixel rows through an LDS-DMA ring of 8 stages, channel
    slices of a stripe on one XCD, fragments of the next stage read ahead of the barrier) against crnn_gemm_bf16_ex mode 1:
    same MFMA, same k order -> the very same bf16 result; and against an fp64 product.  Ragged M, 1..8 channel slices, K of
    1..8 stages, fewer stripes than workgroups, more stripes than the ring is deep times the grid (long persistent loops); round 5: 36 and 9 slices
    (more than an XCD has CUs: dense1's data gradient at K <= 128)."""
    rs = np.random.RandomState(M + N + K)
    X = _bf16_round(rs.normal(size=(M, K))); W = _bf16_round(rs.normal(size=(N, K)) * 0.2)
    Xd, Wd = _to_bf16_dev(X), _to_bf16_dev(W)
    Y = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device="cuda")      # rows past M must stay untouched
    assert L().crnn_gemm_wres_supported(N, K) == 0
    for rep in range(2):                                                      # a second launch over the same ring
        ok(L().crnn_gemm_wres_bf16(P(Xd), P(Wd), P(Y), M, N, K, S()))
    Y2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_gemm_bf16_ex(1, P(Xd), P(Wd), P(Y2), M, N, K, K, K, N, None, 0, 0, 0, None, 0, 1, 1, 1, S()))
    assert torch.equal(Y[:M], Y2), "differs from the tile kernel: max %g" % float((Y[:M].float() - Y2.float()).abs().max())
    assert bool((Y[M:] == 7.0).all())
    if M <= 70000:
        ref = X @ W.T
        assert_close(Y[:M].float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2 * np.abs(ref).max(), what="wres gemm vs fp64")
    assert L().crnn_gemm_wres_bf16(P(Xd), P(Wd), P(Y), M, 64, K, S()) == -3 and L().crnn_gemm_wres_supported(N, 192) == -3
    assert L().crnn_gemm_wres_supported(1152, 256) == -3 and L().crnn_gemm_wres_supported(8320, 128) == -3 and L().crnn_gemm_wres_supported(1152, 64) == 0


@pytest.mark.parametrize("M,N,K", [(128 * 3, 128, 256), (128 * 41, 256, 256), (128 * 300, 256, 512), (128 * 531, 512, 512), (128 * 936, 512, 512), (128 * 7488, 128, 256),
                                   (128, 1024, 256), (128 * 9, 384, 512)])
def test_weights_resident_data_gradient_with_batchnorm_backward_statistics(M, N, K):
    """crnn_gemm_wres_bf16_bnstats: the data-gradient GEMM da = dq . W^T whose storer waves also take the statistics pass of the depthwise
    BatchNorm's backward.  da must be the very bits of crnn_gemm_wres_bf16; dgamma / dbeta / coef after
    crnn_bn_bwd_finalize must equal crnn_bn_bwd_ex's statistics pass on the same da and d up to the order of the fp32 partial sums, and an
    fp64 evaluation of sum(gy), sum(gy * xhat) with gy = da where 0 < d * scale + shift < 6.  One stripe, fewer stripes than workgroups, odd
    and even numbers of stripes per workgroup, 1..8 channel slices, the CRNN's own shapes (blocks 3, 6/7 at batch 256)."""
    rs = np.random.RandomState(M % 1000 + N + K)
    dq = _bf16_round(rs.normal(size=(M, K))); W = _bf16_round(rs.normal(size=(N, K)) * 0.1)
    d = _bf16_round(rs.normal(size=(M, N)) * 1.5 + 0.3)
    gamma = rs.uniform(0.5, 1.5, N); beta = rs.normal(size=N) * 0.5 + 1.0
    mean = d.mean(0); var = d.var(0)
    inv = 1.0 / np.sqrt(var.astype(np.float32) + np.float32(1e-3))
    scale = (gamma * inv).astype(np.float32); shift = (beta - mean * gamma * inv).astype(np.float32)
    bnstate = dev(np.concatenate([mean, var, scale, shift]).astype(np.float32))
    dqd, Wd, dd = _to_bf16_dev(dq), _to_bf16_dev(W), _to_bf16_dev(d)
    assert L().crnn_gemm_wres_bnstats_supported(M, N, K) == 0
    rows = L().crnn_gemm_wres_bnstats_rows(M, N, K)
    assert rows > 0
    parts = torch.full((rows * 2 * N,), float("nan"), device="cuda")         # every row and column must be written
    da = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    dg, db, coef = zeros(N), zeros(N), zeros(2 * N)
    for rep in range(2):
        ok(L().crnn_gemm_wres_bf16_bnstats(P(dqd), P(Wd), P(da), M, N, K, P(dd), P(bnstate), P(parts), S()))
    ok(L().crnn_bn_bwd_finalize(P(parts), rows, N, M, P(dg), P(db), P(coef), S()))
    assert bool(torch.isfinite(parts).all())
    da0 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_gemm_wres_bf16(P(dqd), P(Wd), P(da0), M, N, K, S()))
    assert torch.equal(da, da0), "da differs from crnn_gemm_wres_bf16"
    # the stand-alone statistics pass on the same tensors (B x H x W = 1 x 1 x M)
    nch = L().crnn_bn_bwd_chunks(M)
    parts2 = zeros(nch * 2 * N); dg2, db2, coef2 = zeros(N), zeros(N), zeros(2 * N)
    ok(L().crnn_bn_bwd_ex(P(dd), P(da0), P(bnstate), P(dev(gamma)), None, P(dg2), P(db2), P(parts2), P(coef2), 1, 1, M, N, 1, 1, 0.0, 0, 0, 1, S()))
    for a, b, what in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (coef, coef2, "coef")):
        a, b = host(a).astype(np.float64), host(b).astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-6, (what, np.abs(a - b).max(), np.abs(b).max())
    # fp64 statistics of the device's own da
    g = da0.float().cpu().numpy().astype(np.float64)
    t = d.astype(np.float32) * scale + shift                                  # fp32 fma on the device; threshold ties are measure-zero here
    live = (t > 0) & (t < 6)
    gy = np.where(live, g, 0.0)
    xhat = (d - mean.astype(np.float32).astype(np.float64)) * inv.astype(np.float64)
    assert_close(host(db), gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="dbeta vs fp64")
    assert_close(host(dg), (gy * xhat).sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy * xhat).sum(0).max(), what="dgamma vs fp64")
    # shapes outside the rules are refused
    assert L().crnn_gemm_wres_bnstats_supported(M + 5, N, K) == -3 and L().crnn_gemm_wres_bnstats_supported(M, N, 128) == -3
    assert L().crnn_gemm_wres_bf16_bnstats(P(dqd), P(Wd), P(da), M, N, K, None, P(bnstate), P(parts), S()) == -2


@pytest.mark.parametrize("M,N,K", [(128 * 3, 128, 64), (128 * 40, 256, 128), (128 * 700 + 0, 128, 64), (128 * 300, 512, 256), (128 * 530, 512, 512), (128, 1024, 64),
                                   (128 * 117, 256, 128), (128 * 117, 128, 64), (128 * 1500, 256, 256)])
def test_weights_resident_forward_pointwise_equals_the_tile_kernel(M, N, K):
    """crnn_pwconv_bnrelu6_fwd_wres (register-resident weights, IO waves applying BatchNorm + ReLU6 on the way into the LDS ring,
    draining the staged stripes and accumulating the BatchNorm-2 statistics) against crnn_pwconv_bnrelu6_fwd (tile kernel, W^T
    operand): q bit for bit; the statistics = the column sums / sums of squares of the stored q (fp64 reference of the device's
    own q, and the tile kernel's partial rows summed) to fp32 summation round-off.  1..8 channel slices, 1..8 stages, stripes fewer
    than / many times the workgroups, workgroups without a stripe (their statistics rows must be zero)."""
    rs = np.random.RandomState(M % 9973 + N + K)
    d = _bf16_round(rs.normal(size=(M, K)) * 2.0); W = _bf16_round(rs.normal(size=(N, K)) * 0.2)
    mean, var = rs.normal(size=K) * 0.3, rs.uniform(0.5, 2.0, size=K)
    scale = rs.normal(size=K) * 0.3 + 1.0; shift = rs.normal(size=K) * 0.5 + 1.0
    st = dev(np.concatenate([mean, var, scale, shift]))
    dd, Wd = _to_bf16_dev(d), _to_bf16_dev(W)
    assert L().crnn_pwconv_fwd_wres_supported(M, N, K) == 0
    rows = L().crnn_pwconv_fwd_wres_rows(M, N, K)
    q1 = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda"); parts1 = torch.full((rows, 2, N), 3.0, device="cuda")
    for rep in range(4):                                      # repeated: a hand-off race between the wave roles shows as run-to-run differences
        ok(L().crnn_pwconv_bnrelu6_fwd_wres(P(dd), P(st), P(Wd), P(q1), M, N, K, P(parts1), S()))
        if rep == 0: q_first, parts_first = q1.clone(), parts1.clone()
        else: assert torch.equal(q_first, q1) and torch.equal(parts_first, parts1), "run %d differs from run 0" % rep
    rows2 = L().crnn_pwconv_stat_rows(M)
    q2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); parts2 = zeros(rows2, 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd(P(dd), P(st), P(Wd), P(q2), M, N, K, P(parts2), 1, 1, S()))
    assert torch.equal(q1[:M], q2), "q differs: max %g" % float((q1[:M].float() - q2.float()).abs().max())
    assert bool((q1[M:] == 7.0).all())
    got = host(parts1).sum(0); tile = host(parts2).sum(0)
    qf = q2.float().cpu().numpy().astype(np.float64)
    ref = np.stack([qf.sum(0), (qf * qf).sum(0)])
    assert_close(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max(), what="statistics vs fp64 sums of the stored q")
    assert_close(got, tile, rtol=2e-5, atol=2e-5 * np.abs(ref).max(), what="statistics vs the tile kernel's")
    # against the oracle product of the transformed operand (bf16 output rounding)
    if M <= 128 * 40:
        a = _bf16_round(np.clip(d * scale + shift, 0.0, 6.0))
        assert_close(q1[:M].float().cpu().numpy(), a @ W.T, rtol=1e-2, atol=1e-2 * np.abs(a @ W.T).max(), what="q vs fp64")
    assert L().crnn_pwconv_fwd_wres_supported(M + 64, N, K) == -3 and L().crnn_pwconv_fwd_wres_supported(M, N + 64, K) == -3


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (64 * 7, 256, 128), (64 * 1000, 128, 256), (64 * 531, 512, 512), (64 * 3000 + 64, 256, 256), (64 * 40, 1024, 128)])
def test_streaming_pointwise_weight_gradient_equals_the_tile_kernel(M, N, K):
    """crnn_pwconv_bnrelu6_wgrad_stream (output tile in the MFMA waves' registers, IO waves streaming BatchNorm + ReLU6-transformed d
    rows and g rows k-major through an LDS ring, ranges of chunks per workgroup, fixed-order second stage) against
    crnn_pwconv_bnrelu6_wgrad (tile GEMM, other range boundaries): fp32 summation round-off; and against the fp64 product of
    the transformed operand.  One chunk in all, ranges shorter than the pipeline, ragged last range, 1..32 output tiles; repeated
    launches give the same bits (deterministic)."""
    rs = np.random.RandomState(M % 9973 + N + K)
    d = _bf16_round(rs.normal(size=(M, K)) * 2.0); g = _bf16_round(rs.normal(size=(M, N)))
    mean, var = rs.normal(size=K) * 0.3, rs.uniform(0.5, 2.0, size=K)
    scale = rs.normal(size=K) * 0.3 + 1.0; shift = rs.normal(size=K) * 0.5 + 1.0
    st = dev(np.concatenate([mean, var, scale, shift]))
    dd, gd = _to_bf16_dev(d), _to_bf16_dev(g)
    assert L().crnn_pwconv_wgrad_stream_supported(M, N, K) == 0
    nb = L().crnn_pwconv_wgrad_stream_scratch_bytes(M, N, K)
    assert 0 < nb <= 64 << 20
    scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    dw1 = torch.full((K, N), 7.0, device="cuda"); dw1b = torch.full((K, N), 5.0, device="cuda")
    ok(L().crnn_pwconv_bnrelu6_wgrad_stream(P(dd), P(st), P(gd), P(dw1), M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()))
    scratch.fill_(float("nan"))
    ok(L().crnn_pwconv_bnrelu6_wgrad_stream(P(dd), P(st), P(gd), P(dw1b), M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()))
    assert torch.equal(dw1, dw1b)
    dw2 = zeros(K, N)
    ok(L().crnn_pwconv_bnrelu6_wgrad(P(dd), P(st), P(gd), P(dw2), M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()))
    a = _bf16_round(np.clip(d * scale + shift, 0.0, 6.0))
    ref = a.T @ g
    # fp32 accumulation of a NON-NEGATIVE operand (ReLU6 output) over up to ~200 k pixels: the running sums grow monotonically, the
    # rounding errors of a chain of n chunks add up to ~n * 2^-24 of the sum (measured 1.9e-4 of the largest entry at M = 192 k)
    tol = (2e-5 + 6e-8 * M / 64) * np.abs(ref).max()
    assert_close(host(dw1), host(dw2), rtol=1e-4, atol=tol, what="stream vs tile kernel")
    # (the fp64 reference rounds ReLU6(BN(d)) to bf16 from fp64, the device from its fp32 fma: a few operands differ by one bf16 ulp)
    assert_close(host(dw1), ref, rtol=1e-4, atol=tol + 1e-3 * np.abs(ref).max(), what="stream vs fp64")
    assert L().crnn_pwconv_wgrad_stream_supported(M + 32, N, K) == -3 and L().crnn_pwconv_wgrad_stream_supported(M, N, 64) == -3


@pytest.mark.parametrize("M,N,K,lda,ldb,ldc", [(128, 1024, 13312, 128, 1024, 1024), (256, 1024, 13056, 512, 1024, 1024), (256, 512, 64 * 5, 256, 1024, 1024),
                                               (128, 128, 64, 128, 128, 128)])
def test_streaming_tn_gemm_on_fp32_operands_equals_the_tile_kernel(M, N, K, lda, ldb, ldc):
    """crnn_gemm_tn_stream (the weight-gradient stream on fp32 operands: rows rounded to bf16 on the way into the LDS ring) against
    crnn_gemm_bf16_ex mode 2 with fp32 A and B (same rounding, other reduction ranges) and against the fp64 product of the
    bf16-rounded operands: fp32 summation round-off.  The recurrent layers' shapes (T*B rows, strided H operand, partial column range of a
    wider result), short reductions; the columns of C beyond N stay untouched; repeated launches give the same bits."""
    rs = np.random.RandomState(M + N + K % 977)
    A = rs.normal(size=(K, lda)); Bm = rs.normal(size=(K, ldb))
    Ad, Bd = dev(A), dev(Bm)
    scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    C1 = torch.full((M, ldc), 7.0, device="cuda"); C2 = torch.full((M, ldc), 7.0, device="cuda"); C3 = torch.full((M, ldc), 7.0, device="cuda")
    ok(L().crnn_gemm_tn_stream(P(Ad), lda, P(Bd), ldb, P(C1), ldc, M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()))
    scratch.fill_(float("nan"))
    ok(L().crnn_gemm_tn_stream(P(Ad), lda, P(Bd), ldb, P(C3), ldc, M, N, K, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()))
    assert torch.equal(C1, C3)
    ok(L().crnn_gemm_bf16_ex(2, P(Ad), P(Bd), P(C2), M, N, K, lda, ldb, ldc, None, 0, 0, 0, P(scratch), ctypes.c_size_t(scratch.numel() * 4), 0, 0, 0, S()))
    ref = _bf16_round(A[:, :M]).T @ _bf16_round(Bm[:, :N])
    tol = (2e-5 + 6e-8 * K / 64) * np.abs(ref).max() + 1e-5 * np.sqrt(K)
    assert_close(host(C1)[:, :N], host(C2)[:, :N], rtol=1e-4, atol=tol, what="stream vs tile kernel")
    assert_close(host(C1)[:, :N], ref, rtol=1e-4, atol=tol, what="stream vs fp64")
    if ldc > N: assert bool((C1[:, N:] == 7.0).all())
    assert L().crnn_gemm_tn_stream(P(Ad), lda, P(Bd), ldb, P(C1), ldc, M, N, K - 32, P(scratch), ctypes.c_size_t(scratch.numel() * 4), S()) == -3


@pytest.mark.parametrize("dt", [0, 1])
def test_transpose_batch_of_eight_matrices_of_very_different_sizes(dt):
    """crnn_transpose_batch: eight transposes in one launch (a one-dimensional grid, every matrix owns its own block range -- round 5: dense1's 4608 x 128
    weight joined the pointwise convolutions' in the forward's batch), fp32 and bf16 outputs, ragged sizes; elements outside the outputs stay untouched."""
    rs = np.random.RandomState(17 + dt)
    shapes = [(64, 128), (128, 256), (4608, 128), (1, 1), (33, 65), (512, 38), (100, 7), (31, 200)]
    src = rs.normal(size=sum(r * c for r, c in shapes) + 5).astype(np.float32)
    in_off, out_off, o = [], [], 0
    for r, c in shapes:
        in_off.append(o); out_off.append(o + 3); o += r * c
    total = o + 16
    dst = torch.full((total,), 9.0, device="cuda", dtype=torch.bfloat16 if dt else torch.float32)
    arr = lambda v, t: (t * len(v))(*v)
    ok(L().crnn_transpose_batch(P(dev(src)), P(dst), len(shapes), arr(in_off, ctypes.c_long), arr(out_off, ctypes.c_long), arr([r for r, _ in shapes], ctypes.c_int),
                                arr([c for _, c in shapes], ctypes.c_int), dt, S()))
    got = dst.float().cpu().numpy()
    want = np.full(total, 9.0, dtype=np.float32)
    for (r, c), i, j in zip(shapes, in_off, out_off):
        m = src[i:i + r * c].reshape(r, c).T
        want[j:j + r * c] = (_bf16_round(m) if dt else m).reshape(-1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("M,N,K,T,rate", [(52 * 64, 128, 4608, 52, 0.4), (52 * 256, 128, 4608, 52, 0.4), (64 * 3, 256, 192, 0, 0.0), (26 * 64, 128, 1152, 26, 0.25)])
def test_dense_forward_stream_with_relu_row_permutation_and_dropout(M, N, K, T, rate):
    """crnn_dense_fwd_stream (round 5: dense1's forward on the 64-row stripe stream, bf16 x7 against a bf16 W^T; bias + ReLU + the rows batch-major ->
    time-major + Dropout in the epilogue) against the fp64 evaluation of the bf16 operands (fp32 accumulation over K: 2e-5 of the scale) with the
    multipliers crnn_dropout_mask gives the site; dropped and clamped elements are exact zeros; repeated launches give the same bits."""
    rs = np.random.RandomState(M + N + K)
    X = _bf16_round(np.maximum(rs.normal(size=(M, K)), 0) * 2.0); WT = _bf16_round(rs.normal(size=(N, K)) * 0.02); bias = rs.normal(size=N).astype(np.float32) * 0.5
    Xd, Wd, bd = _to_bf16_dev(X), _to_bf16_dev(WT), dev(bias)
    seed, layer = 4242 + M, 8
    assert L().crnn_dense_fwd_stream_supported(M, N, K) == 0
    Y = torch.full((M + 2, N), 9.0, device="cuda"); Y2 = torch.full((M + 2, N), 9.0, device="cuda")
    ok(L().crnn_dense_fwd_stream(P(Xd), P(Wd), P(bd), P(Y), M, N, K, K, K, 1, T, rate, seed, layer, S()))
    ok(L().crnn_dense_fwd_stream(P(Xd), P(Wd), P(bd), P(Y2), M, N, K, K, K, 1, T, rate, seed, layer, S()))
    assert torch.equal(Y, Y2) and bool((Y[M:] == 9.0).all())
    want = np.maximum(X @ WT.T + bias.astype(np.float64), 0.0)
    if T:
        m = np.arange(M); orow = (m % T) * (M // T) + m // T
        perm = np.empty_like(want); perm[orow] = want; want = perm
    mask = torch.ones(M * N, device="cuda")
    if rate > 0:
        ok(L().crnn_dropout_mask(P(mask), M * N, rate, seed, layer, S()))
    mask = host(mask).reshape(M, N).astype(np.float64)
    want = want * mask
    got = host(Y[:M])
    assert_close(got, want, rtol=2e-5, atol=2e-5 * np.abs(want).max(), what="dense forward stream")
    assert (got[mask == 0] == 0).all() and (got >= 0).all()
    Y3 = torch.empty(M, N, device="cuda")       # no ReLU, no permutation, no dropout: the plain product + bias
    ok(L().crnn_dense_fwd_stream(P(Xd), P(Wd), P(bd), P(Y3), M, N, K, K, K, 0, 0, 0.0, seed, layer, S()))
    plain = X @ WT.T + bias.astype(np.float64)
    assert_close(host(Y3), plain, rtol=2e-5, atol=2e-5 * np.abs(plain).max(), what="plain product")
    assert L().crnn_dense_fwd_stream_supported(M + 32, N, K) == -3 and L().crnn_dense_fwd_stream_supported(M, 384, K) == -3 and L().crnn_dense_fwd_stream_supported(M, N, K + 32) == -3
    if T:
        assert L().crnn_dense_fwd_stream(P(Xd), P(Wd), P(bd), P(Y), M, N, K, K, K, 1, T + 1 if M % (T + 1) else T + 7, rate, seed, layer, S()) == -3      # rows not a multiple of permP


def test_dense1_stream_inference_rows_do_not_depend_on_their_position():
    """crnn_dense_fwd_stream without dropout (the inference launch: predict.py at any batch size): every workgroup walks the reduction in ascending order, so a
    row's fp32 sum is the same bits wherever the row sits in the batch -- rows permuted on the way in come out permuted, and the first stripe of a long batch
    equals a short batch of the same rows.  (Training launches, drop_rate > 0, rotate each stripe's start -- round 5's memory-channel skew -- and are only
    deterministic run to run: the header says so.)"""
    M, N, K = 64 * 12, 128, 4608
    rs = np.random.RandomState(5)
    X = _bf16_round(np.maximum(rs.normal(size=(M, K)), 0) * 2.0); WT = _bf16_round(rs.normal(size=(N, K)) * 0.02); bias = rs.normal(size=N).astype(np.float32) * 0.5
    perm = rs.permutation(M)
    Wd, bd = _to_bf16_dev(WT), dev(bias)
    Y0, Y1, Y2 = zeros(M, N), zeros(M, N), zeros(64, N)
    ok(L().crnn_dense_fwd_stream(P(_to_bf16_dev(X)), P(Wd), P(bd), P(Y0), M, N, K, K, K, 1, 0, 0.0, 1, 8, S()))
    ok(L().crnn_dense_fwd_stream(P(_to_bf16_dev(X[perm])), P(Wd), P(bd), P(Y1), M, N, K, K, K, 1, 0, 0.0, 1, 8, S()))
    ok(L().crnn_dense_fwd_stream(P(_to_bf16_dev(X[64 * 7:64 * 8])), P(Wd), P(bd), P(Y2), 64, N, K, K, K, 1, 0, 0.0, 1, 8, S()))
    assert np.array_equal(host(Y1), host(Y0)[perm]), "a row's result depends on its position in the batch"
    assert np.array_equal(host(Y2), host(Y0)[64 * 7:64 * 8]), "a row's result depends on the batch size"


@pytest.mark.parametrize("M,N,K,lda,ldb,ldc", [(4608, 128, 52 * 256, 4608, 128, 128), (4608, 128, 52 * 64, 4608, 128, 128), (1152, 128, 52 * 16, 1152, 128, 128),
                                               (256, 256, 640, 264, 256, 260), (2304, 128, 64, 2304, 136, 128)])
def test_streaming_tn_gemm_on_bf16_operands_equals_the_tile_kernel(M, N, K, lda, ldb, ldc):
    """crnn_gemm_tn_bf16_stream (round 5: dense1's weight gradient dW1 = x7^T gbm on the weight-gradient stream, bf16 operands, no transform; 36 / 9 / 18
    feature tiles -- more than an XCD has CUs -- over cus / tiles row ranges, a range's tiles spread over the XCDs) against crnn_gemm_bf16_ex
    mode 2 on the same bf16 operands (other reduction ranges: fp32 summation round-off) and the fp64 product; repeated launches give the same bits;
    columns of C beyond N stay untouched."""
    rs = np.random.RandomState(M + N + K % 977)
    A = _bf16_round(np.maximum(rs.normal(size=(K, lda)), 0) * 2.2); Bm = _bf16_round(rs.normal(size=(K, ldb)) * 0.05)
    Ad, Bd = _to_bf16_dev(A), _to_bf16_dev(Bm)
    assert L().crnn_gemm_tn_bf16_stream_supported(M, N, K) == 0
    nb = L().crnn_gemm_tn_bf16_stream_scratch_bytes(M, N, K)
    assert 0 < nb <= 64 << 20
    scratch = torch.empty(nb // 4, dtype=torch.float32, device="cuda")
    C1 = torch.full((M, ldc), 7.0, device="cuda"); C2 = torch.full((M, ldc), 7.0, device="cuda"); C3 = torch.full((M, ldc), 7.0, device="cuda")
    ok(L().crnn_gemm_tn_bf16_stream(P(Ad), lda, P(Bd), ldb, P(C1), ldc, M, N, K, P(scratch), nb, S()))
    scratch.fill_(float("nan"))
    ok(L().crnn_gemm_tn_bf16_stream(P(Ad), lda, P(Bd), ldb, P(C3), ldc, M, N, K, P(scratch), nb, S()))
    assert torch.equal(C1, C3)
    big = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    ok(L().crnn_gemm_bf16_ex(2, P(Ad), P(dev(Bm)), P(C2), M, N, K, lda, ldb, ldc, None, 0, 0, 0, P(big), ctypes.c_size_t(big.numel() * 4), 1, 0, 0, S()))      # (bf16 A, fp32 B holding bf16 values: dense1's former call)
    ref = A[:, :M].T @ Bm[:, :N]
    tol = (2e-5 + 6e-8 * K / 64) * np.abs(ref).max() + 1e-5 * np.sqrt(K)
    assert_close(host(C1)[:, :N], host(C2)[:, :N], rtol=1e-4, atol=tol, what="stream vs tile kernel")
    assert_close(host(C1)[:, :N], ref, rtol=1e-4, atol=tol, what="stream vs fp64")
    if ldc > N: assert bool((C1[:, N:] == 7.0).all())
    assert L().crnn_gemm_tn_bf16_stream(P(Ad), lda, P(Bd), ldb, P(C1), ldc, M, N, K, P(scratch), nb - 4, S()) == -3      # short scratch
    assert L().crnn_gemm_tn_bf16_stream_supported(M + 64, N, K) == -3 and L().crnn_gemm_tn_bf16_stream_supported(8320, 128, 64) == -3


@pytest.mark.parametrize("M,N,K", [(128 * 5, 128, 64), (128 * 300 + 17, 256, 256), (128 * 700, 512, 512), (100, 128, 128)])
def test_weights_resident_inference_conv_with_folded_batchnorm_equals_the_tile_kernel(M, N, K):
    """crnn_pwconv_fwd_wres_folded (predict path: pointwise conv with the following BatchNorm + ReLU6 applied to the fp32 accumulators
    in the MFMA waves before the one rounding to bf16) against crnn_pwconv_fwd(out_bnstate) on the tile GEMM: bit for bit; and
    against the fp64 evaluation.  Ragged M, 1..4 channel slices, repeated launches."""
    rs = np.random.RandomState(M % 9973 + N + K)
    a = _bf16_round(np.abs(rs.normal(size=(M, K))) * 1.5); W = _bf16_round(rs.normal(size=(N, K)) * 0.2)
    scale, shift = rs.normal(size=N) * 0.3 + 1.0, rs.normal(size=N) * 0.5 + 1.0
    st = dev(np.concatenate([rs.normal(size=N), rs.uniform(0.5, 2.0, size=N), scale, shift]))
    ad, Wd = _to_bf16_dev(a), _to_bf16_dev(W)
    y1 = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")
    for rep in range(3):
        ok(L().crnn_pwconv_fwd_wres_folded(P(ad), P(Wd), P(y1), M, N, K, P(st), S()))
        if rep == 0: first = y1.clone()
        else: assert torch.equal(first, y1)
    y2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_pwconv_fwd(P(ad), P(Wd), P(y2), M, N, K, None, P(st), 1, 1, 1, 1, 1, S()))
    assert torch.equal(y1[:M], y2), "differs from the tile kernel: max %g" % float((y1[:M].float() - y2.float()).abs().max())
    assert bool((y1[M:] == 7.0).all())
    ref = np.clip((a @ W.T) * scale + shift, 0.0, 6.0)
    assert_close(y1[:M].float().cpu().numpy(), ref, rtol=1e-2, atol=2e-2, what="folded conv vs fp64")


@pytest.mark.parametrize("M,N,K", [(128, 64, 128), (128 * 9, 128, 256), (128 * 40, 256, 256), (128 * 21, 512, 512), (128 * 1500, 128, 256), (128 * 7, 64, 128)])
def test_three_plane_data_gradient_with_batchnorm_backward_statistics(M, N, K):
    """crnn_gemm_f32x3_bnstats (parity mode): the pointwise conv's data gradient da = dq . W^T as three-plane products whose epilogue also takes
    the statistics pass of the depthwise BatchNorm's backward.  da must be the very bits of crnn_gemm_f32x3; dgamma / dbeta / coef after
    crnn_bn_bwd_finalize_folded must equal crnn_bn_bwd_ex's statistics pass on the same da and d up to the order of the fp32 partial sums and an
    fp64 evaluation; crnn_bn_bwd_apply_ex (pass 2 alone) must reproduce crnn_bn_bwd_ex's dx from the same coefficients bit for bit.  One and
    many tile rows (beyond 1024: the folded finalize), both tile widths."""
    rs = np.random.RandomState(M % 1000 + N + K + 7)
    dq = rs.normal(size=(M, K)).astype(np.float32); W = (rs.normal(size=(N, K)) * 0.1).astype(np.float32)
    d = (rs.normal(size=(M, N)) * 1.5 + 0.3).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, N); beta = rs.normal(size=N) * 0.5 + 1.0
    mean = d.astype(np.float64).mean(0); var = d.astype(np.float64).var(0)
    inv = 1.0 / np.sqrt(var.astype(np.float32) + np.float32(1e-3))
    scale = (gamma * inv).astype(np.float32); shift = (beta - mean * gamma * inv).astype(np.float32)
    bnstate = dev(np.concatenate([mean, var, scale, shift]).astype(np.float32))
    dqd, Wd, dd = dev(dq), dev(W), dev(d)
    assert L().crnn_gemm_f32x3_bnstats_supported(M, N, K) == 0
    rows = L().crnn_gemm_f32x3_bnstats_rows(M)
    assert rows == M // 128
    parts = torch.full((rows * 2 * N,), float("nan"), device="cuda")         # every row and column must be written
    da = zeros(M, N); dg, db, coef = zeros(N), zeros(N), zeros(2 * N); fold = zeros(32 * 2 * N)
    for rep in range(2):
        ok(L().crnn_gemm_f32x3_bnstats(P(dqd), P(Wd), P(da), M, N, K, P(dd), P(bnstate), P(parts), S()))
    assert bool(torch.isfinite(parts).all())
    ok(L().crnn_bn_bwd_finalize_folded(P(parts), rows, N, M, P(dg), P(db), P(coef), P(fold), S()))
    da0 = zeros(M, N); scr = zeros(16)
    ok(L().crnn_gemm_f32x3(1, P(dqd), P(Wd), P(da0), M, N, K, K, K, N, None, 0, 0, 0, None, 0, S()))
    assert torch.equal(da, da0), "da differs from crnn_gemm_f32x3"
    nch = L().crnn_bn_bwd_chunks(M)
    parts2 = zeros(nch * 2 * N); dg2, db2, coef2 = zeros(N), zeros(N), zeros(2 * N); dx2 = zeros(M, N)
    ok(L().crnn_bn_bwd_ex(P(dd), P(da0), P(bnstate), P(dev(gamma)), P(dx2), P(dg2), P(db2), P(parts2), P(coef2), 1, 1, M, N, 1, 1, 0.0, 0, 0, 0, S()))
    for a, b, what in ((dg, dg2, "dgamma"), (db, db2, "dbeta"), (coef, coef2, "coef")):
        a, b = host(a).astype(np.float64), host(b).astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-6, (what, np.abs(a - b).max(), np.abs(b).max())
    dx1 = torch.full((M, N), 7.0, device="cuda")
    ok(L().crnn_bn_bwd_apply_ex(P(dd), P(da0), P(bnstate), P(coef2), P(dx1), 1, 1, M, N, 1, 1, 0.0, 0, 0, 0, S()))
    assert torch.equal(dx1, dx2), "pass 2 alone differs from crnn_bn_bwd_ex"
    g = host(da0).astype(np.float64)
    t = d * scale + shift
    live = (t > 0) & (t < 6)
    gy = np.where(live, g, 0.0)
    xhat = (d.astype(np.float64) - mean.astype(np.float32).astype(np.float64)) * inv.astype(np.float64)
    assert_close(host(db), gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="dbeta vs fp64")
    assert_close(host(dg), (gy * xhat).sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy * xhat).sum(0).max(), what="dgamma vs fp64")
    assert L().crnn_gemm_f32x3_bnstats_supported(M + 5, N, K) == -3 and L().crnn_gemm_f32x3_bnstats_supported(M, 96, K) == -3
    assert L().crnn_gemm_f32x3_bnstats(P(dqd), P(Wd), P(da), M, N, K, None, P(bnstate), P(parts), S()) == -2


@pytest.mark.parametrize("M,K,N", [(128 * 40, 128, 256), (1000, 64, 128), (128 * 6, 512, 512), (128 * 3 + 5, 256, 64), (128 * 200, 64, 128)])
def test_three_plane_gemms_apply_batchnorm_relu6_while_staging(M, K, N):
    """crnn_pwconv_bnrelu6_fwd_f32x3 / crnn_pwconv_bnrelu6_wgrad_f32x3 (parity mode): the staging waves of the three-plane kernel apply
    ReLU6(d * scale + shift) to the raw fp32 items before the plane split.  Both must equal the unfused sequence -- crnn_bn_act_pool_drop_ex,
    then crnn_pwconv_fwd(bf16_products = 2) with its statistics / crnn_gemm_f32x3 mode 2 -- bit for bit: whole and ragged tiles, both tile
    widths, the split reduction of the weight gradient."""
    rs = np.random.RandomState(M % 1000 + N + K + 3)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); w = dev((rs.normal(size=(K, N)) * 0.1).astype(np.float32))
    g = dev(rs.normal(size=(M, N)).astype(np.float32))
    st = dev(np.concatenate([rs.normal(size=K), rs.uniform(0.5, 2.0, size=K), rs.normal(size=K) * 0.3 + 1.0, rs.normal(size=K) * 0.5 + 0.5]).astype(np.float32))
    a = zeros(M, K)
    ok(L().crnn_bn_act_pool_drop_ex(P(d), P(st), P(a), 1, 1, M, K, 1, 1, 0.0, 0, 0, 0, 0, S()))
    rows = L().crnn_pwconv_stat_rows(M)
    q0, q1 = zeros(M, N), torch.full((M + 2, N), 7.0, device="cuda"); p0, p1 = zeros(rows, 2, N), zeros(rows, 2, N)
    ok(L().crnn_pwconv_fwd(P(a), P(w), P(q0), M, N, K, P(p0), None, 2, 0, 0, 0, 0, S()))
    for rep in range(2):
        ok(L().crnn_pwconv_bnrelu6_fwd_f32x3(P(d), P(st), P(w), P(q1), M, N, K, P(p1), S()))
    assert torch.equal(q1[:M], q0) and bool((q1[M:] == 7.0).all()), "forward differs: max %g" % float((q1[:M] - q0).abs().max())
    assert torch.equal(p1, p0)
    assert float(q0.abs().max()) > 0
    scr = zeros(16 * 1024 * 1024); sb = ctypes.c_size_t(scr.numel() * 4)
    dw0, dw1 = zeros(K, N), zeros(K, N)
    ok(L().crnn_gemm_f32x3(2, P(a), P(g), P(dw0), K, N, M, K, N, N, None, 0, 0, 0, P(scr), sb, S()))
    ok(L().crnn_pwconv_bnrelu6_wgrad_f32x3(P(d), P(st), P(g), P(dw1), M, N, K, P(scr), sb, S()))
    assert torch.equal(dw1, dw0), "weight gradient differs: max %g" % float((dw1 - dw0).abs().max())
    ref = host(a).astype(np.float64).T @ host(g).astype(np.float64)
    assert_close(host(dw1), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max() + 1e-6, what="weight gradient vs fp64")


def _bnstate(rs, C):
    return dev(np.concatenate([rs.normal(size=C), rs.uniform(0.5, 2.0, size=C), rs.normal(size=C) * 0.3 + 1.0, rs.normal(size=C) * 0.5 + 0.5]).astype(np.float32))


@pytest.mark.parametrize("planes", [3, 2])
@pytest.mark.parametrize("M,K,N", [(64 * 5, 64, 128), (64 * 1100, 64, 128), (64 * 37, 128, 256), (64 * 700, 128, 256), (32 * 9, 256, 256), (32 * 333, 256, 512),
                                   (32 * 40, 512, 512), (32 * 210, 512, 512), (32 * 70, 512, 64), (64 * 3, 128, 1024)])
def test_resident_weight_plane_forward_equals_the_tile_kernel(M, K, N, planes):
    """crnn_pwconv_bnrelu6_fwd_wres3 (round 6, gemm_wres3.hip): the parity mode's pointwise forward with the weights' planes resident in registers and the
    pixel rows streamed once.  q must equal crnn_pwconv_bnrelu6_fwd_f32x3 / _f32x2 (same planes, same products) up to the ORDER of the fp32 accumulation
    (four interleaved chains per output, two reduction halves at K = 512) and be as close to fp64 as the tile kernel.  The statistics
    partials must give the tile kernel's column sums / sums of squares up to the order of the fp32 partial sums; rows behind M stay untouched; one to
    many stripes per workgroup (the IO waves' prologue, steady state and tail all run)."""
    rs = np.random.RandomState(M % 1000 + N + K + planes)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); w = dev((rs.normal(size=(K, N)) * 0.1).astype(np.float32))
    st = _bnstate(rs, K)
    assert L().crnn_gemm_wres3_supported(M, N, K) == 0
    rows0 = L().crnn_pwconv_stat_rows(M); rows1 = L().crnn_gemm_wres3_stat_rows(M, N, K)
    assert rows1 > 0
    q0 = zeros(M, N); q1 = torch.full((M + 2, N), 7.0, device="cuda"); p0 = zeros(rows0, 2, N); p1 = torch.full((rows1, 2, N), float("nan"), device="cuda")
    ok((L().crnn_pwconv_bnrelu6_fwd_f32x3 if planes == 3 else L().crnn_pwconv_bnrelu6_fwd_f32x2)(P(d), P(st), P(w), P(q0), M, N, K, P(p0), S()))
    for rep in range(2):
        ok(L().crnn_pwconv_bnrelu6_fwd_wres3(P(d), P(st), P(w), P(q1), M, N, K, planes, P(p1), S()))
    assert bool((q1[M:] == 7.0).all()), "rows behind M written"
    assert bool(torch.isfinite(p1).all()), "a statistics row or column was not written"
    a = np.clip(host(d).astype(np.float64) * host(st)[2 * K:3 * K] + host(st)[3 * K:], 0.0, 6.0)
    ref = a @ host(w).astype(np.float64)
    e0 = np.abs(host(q0) - ref).max(); e1 = np.abs(host(q1[:M]) - ref).max(); sc = np.abs(ref).max()
    # the same planes and products as the tile kernel, summed as four interleaved chains (and two reduction halves at K = 512) instead of one: equal up to
    # the order of the fp32 accumulation
    assert float((q1[:M] - q0).abs().max()) <= 2e-6 * sc, (float((q1[:M] - q0).abs().max()), sc)
    assert e1 <= max(1.5 * e0, 2e-6 * sc), (e0, e1, sc)
    if planes == 3: assert e1 <= 3e-6 * sc
    s0 = host(p0).astype(np.float64).sum(0); s1 = host(p1).astype(np.float64).sum(0)
    qh = host(q1[:M]).astype(np.float64)
    for k, exact in ((0, qh.sum(0)), (1, (qh * qh).sum(0))):
        tol = 2e-6 * np.abs(qh if k == 0 else qh * qh).sum(0).max() + 1e-6
        assert np.abs(s1[k] - exact).max() <= tol, ("statistics vs fp64", k, np.abs(s1[k] - exact).max(), tol)
        assert np.abs(s1[k] - s0[k]).max() <= 2 * tol
    # shape rules
    assert L().crnn_gemm_wres3_supported(M + 8, N, K) == -3 and L().crnn_gemm_wres3_supported(M, N + 32, K) == -3 and L().crnn_gemm_wres3_supported(M, N, 96) == -3
    assert L().crnn_pwconv_bnrelu6_fwd_wres3(P(d), P(st), P(w), P(q1), M, N, K, 4, P(p1), S()) == -2
    assert L().crnn_pwconv_bnrelu6_fwd_wres3(P(d), None, P(w), P(q1), M, N, K, 3, P(p1), S()) == -2


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("M,N,K", [(32 * 9, 128, 256), (32 * 500, 128, 256), (32 * 41, 256, 256), (32 * 300, 256, 512), (32 * 40, 512, 512), (32 * 260, 512, 512), (32 * 8, 64, 512)])
def test_resident_weight_plane_data_gradient_with_batchnorm_backward_statistics(M, N, K, planes):
    """crnn_gemm_wres3_bnstats: da = dq . W^T with W's planes resident in registers, dq streamed once, and BatchNorm-1's backward statistics taken while the
    result is drained.  da: crnn_gemm_f32x3_bnstats / _f32x2_bnstats up to the order of the fp32 accumulation; dgamma / dbeta / coef
    after crnn_bn_bwd_finalize_folded equal the tile kernel's up to the order of the partial sums."""
    rs = np.random.RandomState(M % 1000 + N + K + 11 * planes)
    dq = dev(rs.normal(size=(M, K)).astype(np.float32)); W = dev((rs.normal(size=(N, K)) * 0.1).astype(np.float32))
    dh = (rs.normal(size=(M, N)) * 1.5 + 0.3).astype(np.float32); d = dev(dh)
    gamma = rs.uniform(0.5, 1.5, N); beta = rs.normal(size=N) * 0.5 + 1.0
    mean = dh.astype(np.float64).mean(0); var = dh.astype(np.float64).var(0)
    inv = 1.0 / np.sqrt(var.astype(np.float32) + np.float32(1e-3))
    scale = (gamma * inv).astype(np.float32); shift = (beta - mean * gamma * inv).astype(np.float32)
    bnstate = dev(np.concatenate([mean, var, scale, shift]).astype(np.float32))
    assert L().crnn_gemm_wres3_supported(M, N, K) == 0
    rows1 = L().crnn_gemm_wres3_stat_rows(M, N, K)
    p1 = torch.full((rows1, 2, N), float("nan"), device="cuda"); da1 = torch.full((M + 2, N), 7.0, device="cuda")
    for rep in range(2):
        ok(L().crnn_gemm_wres3_bnstats(P(dq), P(W), P(da1), M, N, K, planes, P(d), P(bnstate), P(p1), S()))
    assert bool((da1[M:] == 7.0).all()) and bool(torch.isfinite(p1).all())
    da0 = zeros(M, N)
    if M % 128 == 0 and L().crnn_gemm_f32x3_bnstats_supported(M, N, K) == 0:
        p0 = zeros(L().crnn_gemm_f32x3_bnstats_rows(M), 2, N)
        ok((L().crnn_gemm_f32x3_bnstats if planes == 3 else L().crnn_gemm_f32x2_bnstats)(P(dq), P(W), P(da0), M, N, K, P(d), P(bnstate), P(p0), S()))
    else:
        ok((L().crnn_gemm_f32x3 if planes == 3 else L().crnn_gemm_f32x2)(1, P(dq), P(W), P(da0), M, N, K, K, K, N, None, 0, 0, 0, None, 0, S()))
    ref = host(dq).astype(np.float64) @ host(W).astype(np.float64).T
    sc = np.abs(ref).max()
    assert float((da1[:M] - da0).abs().max()) <= 2e-6 * sc, "da differs from the tile kernel by more than the accumulation order: %g" % float((da1[:M] - da0).abs().max())
    assert np.abs(host(da1[:M]) - ref).max() <= (3e-6 if planes == 3 else 3e-5) * sc
    g = host(da1[:M]).astype(np.float64)
    t = dh * scale + shift
    gy = np.where((t > 0) & (t < 6), g, 0.0)
    xhat = (dh.astype(np.float64) - mean.astype(np.float32).astype(np.float64)) * inv.astype(np.float64)
    s1 = host(p1).astype(np.float64).sum(0)
    assert_close(s1[0], gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="sum gy vs fp64")
    assert_close(s1[1], (gy * xhat).sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy * xhat).sum(0).max(), what="sum gy xhat vs fp64")
    dg, db, coef = zeros(N), zeros(N), zeros(2 * N); fold = zeros(32 * 2 * N)
    ok(L().crnn_bn_bwd_finalize_folded(P(p1), rows1, N, M, P(dg), P(db), P(coef), P(fold), S()))
    assert_close(host(db), gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="dbeta vs fp64")
    assert L().crnn_gemm_wres3_bnstats(P(dq), P(W), P(da1), M, N, 128, planes, P(d), P(bnstate), P(p1), S()) == -3
    assert L().crnn_gemm_wres3_bnstats(P(dq), P(W), P(da1), M, N, K, planes, None, P(bnstate), P(p1), S()) == -2


@pytest.mark.parametrize("M,K,N,bn", [(32 * 7, 128, 128, True), (32 * 900, 128, 256, True), (32 * 64 * 3, 256, 256, True), (32 * 500, 256, 512, True), (32 * 1200, 512, 512, True),
                                        (32 * 333, 512, 128, False), (32 * 2, 128, 128, False)])
def test_two_plane_weight_gradient_on_the_pixel_stream_equals_the_tile_kernel(M, K, N, bn):
    """crnn_pwconv_bnrelu6_wgrad_planes_stream (round 6, gemm_wgrad3.hip): the parity mode's pointwise weight gradient with two bf16 planes per operand on a pixel
    stream -- against crnn_pwconv_bnrelu6_wgrad_f32x2 (same planes and products, other reduction ranges: fp32 summation round-off) and the fp64 product of the
    activated operand (16 significant bits per factor: 3e-5 of the scale); ranges of one chunk to hundreds; without the BatchNorm transform too; repeated launches
    give the same bits."""
    rs = np.random.RandomState(M % 1000 + N + K)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); g = dev(rs.normal(size=(M, N)).astype(np.float32))
    st = _bnstate(rs, K)
    assert L().crnn_pwconv_wgrad_planes_stream_supported(M, N, K) == 0
    nb = L().crnn_pwconv_wgrad_planes_stream_scratch_bytes(M, N, K)
    assert 0 < nb <= 64 << 20
    scr = torch.empty(nb // 4, device="cuda"); scr2 = zeros(16 * 1024 * 1024); sb2 = ctypes.c_size_t(scr2.numel() * 4)
    dw1 = torch.full((K + 1, N), 7.0, device="cuda"); dw2 = torch.full((K + 1, N), 7.0, device="cuda"); dw0 = zeros(K, N)
    ok(L().crnn_pwconv_bnrelu6_wgrad_planes_stream(P(d), P(st) if bn else None, P(g), P(dw1), M, N, K, P(scr), ctypes.c_size_t(nb), S()))
    ok(L().crnn_pwconv_bnrelu6_wgrad_planes_stream(P(d), P(st) if bn else None, P(g), P(dw2), M, N, K, P(scr), ctypes.c_size_t(nb), S()))
    assert torch.equal(dw1, dw2) and bool((dw1[K:] == 7.0).all())
    if bn:
        a = zeros(M, K)
        ok(L().crnn_bn_act_pool_drop_ex(P(d), P(st), P(a), 1, 1, M, K, 1, 1, 0.0, 0, 0, 0, 0, S()))
        ok(L().crnn_pwconv_bnrelu6_wgrad_f32x2(P(d), P(st), P(g), P(dw0), M, N, K, P(scr2), sb2, S()))
    else:
        a = d
        ok(L().crnn_gemm_f32x2(2, P(d), P(g), P(dw0), K, N, M, K, N, N, None, 0, 0, 0, P(scr2), sb2, S()))
    ref = host(a).astype(np.float64).T @ host(g).astype(np.float64)
    sc = np.abs(ref).max()
    assert float((dw1[:K] - dw0).abs().max()) <= 2e-6 * sc + 1e-6, (float((dw1[:K] - dw0).abs().max()), sc)
    assert np.abs(host(dw1[:K]) - ref).max() <= 3e-5 * sc + 1e-6
    assert L().crnn_pwconv_wgrad_planes_stream_supported(M + 8, N, K) == -3 and L().crnn_pwconv_wgrad_planes_stream_supported(M, N, 64) == -3
    assert L().crnn_pwconv_wgrad_planes_stream_supported(M, 1024, 1152) == -3      # 72 tiles
    assert L().crnn_pwconv_bnrelu6_wgrad_planes_stream(P(d), P(st), P(g), P(dw1), M, N, K, P(scr), ctypes.c_size_t(1024), S()) == -3


@pytest.mark.parametrize("rows,M,N,lda,ldb,ldc", [(52 * 256, 128, 1024, 128, 1024, 1024), (51 * 256, 256, 1024, 512, 1024, 1024), (51 * 64, 256, 1024, 256, 1024, 1028),
                                                    (32 * 3, 128, 128, 132, 136, 128), (52 * 256, 4608, 128, 4608, 128, 128), (52 * 16, 1152, 256, 1152, 256, 256)])
def test_two_plane_transposed_product_with_leading_dimensions_on_the_pixel_stream(rows, M, N, lda, ldb, ldc):
    """crnn_gemm_tn_planes_stream (round 6): C = A^T B over the rows of A [rows][lda >= M] and B [rows][ldb >= N] -- the recurrent layers' weight gradients of the
    parity mode (x^T dz, h_prev^T dz; h_prev a column half of the concatenated hidden states: lda = 2u) -- against crnn_gemm_f32x2 mode 2 (same planes and
    products, other reduction ranges) and the fp64 product; columns beyond M / N of the operands and beyond N of C's rows are not touched; repeats give the same bits."""
    rs = np.random.RandomState(rows % 1000 + M + N)
    A = dev(rs.normal(size=(rows, lda)).astype(np.float32)); B = dev((rs.normal(size=(rows, ldb)) * 0.3).astype(np.float32))
    assert L().crnn_pwconv_wgrad_planes_stream_supported(rows, N, M) == 0
    nb = L().crnn_pwconv_wgrad_planes_stream_scratch_bytes(rows, N, M)
    scr = torch.empty(nb // 4, device="cuda"); scr2 = zeros(16 * 1024 * 1024); sb2 = ctypes.c_size_t(scr2.numel() * 4)
    c1 = torch.full((M + 1, ldc), 7.0, device="cuda"); c2 = torch.full((M + 1, ldc), 7.0, device="cuda"); c0 = zeros(M, ldc)
    ok(L().crnn_gemm_tn_planes_stream(P(A), lda, P(B), ldb, P(c1), ldc, M, N, rows, P(scr), ctypes.c_size_t(nb), S()))
    ok(L().crnn_gemm_tn_planes_stream(P(A), lda, P(B), ldb, P(c2), ldc, M, N, rows, P(scr), ctypes.c_size_t(nb), S()))
    assert torch.equal(c1, c2) and bool((c1[M:] == 7.0).all()) and bool((c1[:, N:] == 7.0).all())
    ok(L().crnn_gemm_f32x2(2, P(A), P(B), P(c0), M, N, rows, lda, ldb, ldc, None, 0, 0, 0, P(scr2), sb2, S()))
    ref = host(A)[:, :M].astype(np.float64).T @ host(B)[:, :N].astype(np.float64)
    sc = np.abs(ref).max()
    assert float((c1[:M, :N] - c0[:, :N]).abs().max()) <= 2e-6 * sc + 1e-6, (float((c1[:M, :N] - c0[:, :N]).abs().max()), sc)
    assert np.abs(host(c1[:M, :N]) - ref).max() <= 3e-5 * sc + 1e-6
    assert L().crnn_gemm_tn_planes_stream(P(A), M - 4, P(B), ldb, P(c1), ldc, M, N, rows, P(scr), ctypes.c_size_t(nb), S()) == -3
    assert L().crnn_gemm_tn_planes_stream(P(A), lda, P(B), ldb, P(c1), ldc, M, N, rows, P(scr), ctypes.c_size_t(1024), S()) == -3
    assert L().crnn_gemm_tn_planes_stream(P(A), lda, P(B), ldb, P(c1), ldc, 64, N, rows, P(scr), ctypes.c_size_t(nb), S()) == -3


@pytest.mark.parametrize("M,N,K,pairs,lda,ldw,ldy", [(52 * 256, 128, 1024, 2, 1024, 1024, 128), (52 * 256, 256, 1024, 2, 1024, 1024, 256), (64 * 5, 128, 768, 2, 772, 768, 132),
                                                      (64, 384, 64, 1, 64, 68, 384), (102 * 64, 256, 1024, 2, 1024, 1024, 256)])
def test_two_plane_stripe_stream_of_the_recurrent_input_gradients(M, N, K, pairs, lda, ldw, ldy):
    """crnn_gemm_nt_f32x2_stream (round 6): dX = dZf Wf^T + dZb Wb^T with two bf16 planes per operand in one stripe-stream launch -- against two accumulating
    crnn_gemm_f32x2 launches (same planes and products, other summation order) and the fp64 sum; one pair alone; padded leading dimensions (columns beyond N
    of Y's rows untouched); repeats give the same bits; shapes outside the rules are refused."""
    rs = np.random.RandomState(M % 1000 + N + K)
    A = [dev((rs.normal(size=(M, lda)) * 0.2).astype(np.float32)) for _ in range(2)]
    W = [dev((rs.normal(size=(N, ldw)) * 0.1).astype(np.float32)) for _ in range(2)]
    y1 = torch.full((M + 1, ldy), 7.0, device="cuda"); y2 = torch.full((M + 1, ldy), 7.0, device="cuda"); y0 = zeros(M, ldy)
    a1 = P(A[1]) if pairs == 2 else None; w1 = P(W[1]) if pairs == 2 else None
    ok(L().crnn_gemm_nt_f32x2_stream(P(A[0]), P(W[0]), a1, w1, P(y1), M, N, K, lda, ldw, ldy, S()))
    ok(L().crnn_gemm_nt_f32x2_stream(P(A[0]), P(W[0]), a1, w1, P(y2), M, N, K, lda, ldw, ldy, S()))
    assert torch.equal(y1, y2) and bool((y1[M:] == 7.0).all()) and bool((y1[:, N:] == 7.0).all())
    scr = zeros(16 * 1024 * 1024); sb = ctypes.c_size_t(scr.numel() * 4)
    ok(L().crnn_gemm_f32x2(1, P(A[0]), P(W[0]), P(y0), M, N, K, lda, ldw, ldy, None, 0, 0, 0, P(scr), sb, S()))
    if pairs == 2: ok(L().crnn_gemm_f32x2(1, P(A[1]), P(W[1]), P(y0), M, N, K, lda, ldw, ldy, None, 0, 1, 0, P(scr), sb, S()))
    ref = sum(host(A[i])[:, :K].astype(np.float64) @ host(W[i])[:, :K].astype(np.float64).T for i in range(pairs))
    sc = np.abs(ref).max()
    assert float((y1[:M, :N] - y0[:, :N]).abs().max()) <= 2e-6 * sc + 1e-6, (float((y1[:M, :N] - y0[:, :N]).abs().max()), sc)
    assert np.abs(host(y1[:M, :N]) - ref).max() <= 3e-5 * sc + 1e-6
    assert L().crnn_gemm_nt_f32x2_stream(P(A[0]), P(W[0]), a1, w1, P(y1), M + 32, N, K, lda, ldw, ldy, S()) == -3
    assert L().crnn_gemm_nt_f32x2_stream(P(A[0]), P(W[0]), a1, w1, P(y1), M, 64, K, lda, ldw, ldy, S()) == -3
    assert L().crnn_gemm_nt_f32x2_stream(P(A[0]), P(W[0]), P(A[1]), None, P(y1), M, N, K, lda, ldw, ldy, S()) == -2


def _planes_of(x, stride=None):
    """crnn_split3_planes of a device fp32 tensor -> (planes tensor [3 * stride] of bf16 words, stride)."""
    n = x.numel(); stride = stride or n
    pl = torch.full((3 * stride + 8,), -1, dtype=torch.int16, device="cuda")
    ok(L().crnn_split3_planes(P(x), P(pl), n, stride, S()))
    return pl, stride


@pytest.mark.parametrize("n,extra", [(4, 0), (1028, 0), (128 * 512, 64), (1 << 20, 4)])
def test_split3_planes_are_the_three_bf16_terms_of_each_value(n, extra):
    """crnn_split3_planes: plane 0 = bf16(x) (round to nearest even), plane 1 = bf16(x - plane 0), plane 2 = bf16(x - plane 0 - plane 1); the three
    terms add up to x to 2^-24 relative; words behind a plane's n values (the stride's slack) stay untouched; bad arguments refused."""
    rs = np.random.RandomState(n % 977 + extra)
    x = dev((rs.normal(size=n) * np.exp(rs.uniform(-6, 6, size=n))).astype(np.float32))
    pl, st = _planes_of(x, n + extra)
    w = pl[:3 * st].view(3, st)
    if extra: assert bool((w[:, n:] == -1).all())
    assert bool((pl[3 * st:] == -1).all())
    planes = [(w[k, :n].int() << 16).view(torch.float32) for k in range(3)]
    assert torch.equal(planes[0], x.bfloat16().float())
    r1 = x - planes[0]
    assert torch.equal(planes[1], r1.bfloat16().float())
    assert torch.equal(planes[2], (r1 - planes[1]).bfloat16().float())
    tot = planes[0].double() + planes[1].double() + planes[2].double()
    assert float(((tot - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -23
    assert L().crnn_split3_planes(P(x), P(pl), n + 1, st + 4, S()) == -2 and L().crnn_split3_planes(P(x), P(pl), n, n - 4, S()) == -2
    assert L().crnn_split3_planes(P(x), P(pl), n, st + 2, S()) == -2


@pytest.mark.parametrize("M,K,N", [(128 * 40, 128, 256), (128 * 6, 512, 512), (128 * 3, 256, 64), (128 * 200, 64, 128), (128 * 9, 64, 64)])
def test_three_plane_gemms_read_operands_split_beforehand(M, K, N):
    """The *_pl entry points (weights -- and the incoming gradient -- handed over as bf16 planes from crnn_split3_planes instead of being split by
    every tile that stages them) must return the very bits of their fp32-operand forms: forward product + BatchNorm statistics
    (crnn_pwconv_bnrelu6_fwd_f32x3_pl), data gradient + BatchNorm-backward statistics (crnn_gemm_f32x3_bnstats_pl with the weight planes alone and
    with both operands as planes).  Plane strides with slack, both tile widths; ragged shapes are refused (-3: the caller falls back)."""
    rs = np.random.RandomState(M % 1000 + N + K + 11)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); w = dev((rs.normal(size=(K, N)) * 0.1).astype(np.float32))
    st = dev(np.concatenate([rs.normal(size=K), rs.uniform(0.5, 2.0, size=K), rs.normal(size=K) * 0.3 + 1.0, rs.normal(size=K) * 0.5 + 0.5]).astype(np.float32))
    rows = L().crnn_pwconv_stat_rows(M)
    q0, q1 = zeros(M, N), torch.full((M + 2, N), 7.0, device="cuda"); p0, p1 = zeros(rows, 2, N), zeros(rows, 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd_f32x3(P(d), P(st), P(w), P(q0), M, N, K, P(p0), S()))
    wpl, ws = _planes_of(w, K * N + 64)
    for rep in range(2):
        ok(L().crnn_pwconv_bnrelu6_fwd_f32x3_pl(P(d), P(st), P(w), P(wpl), ws, P(q1), M, N, K, P(p1), S()))
    assert torch.equal(q1[:M], q0) and bool((q1[M:] == 7.0).all()), "forward differs: max %g" % float((q1[:M] - q0).abs().max())
    assert torch.equal(p1, p0) and float(q0.abs().max()) > 0
    if K % 64 == 0:
        # data gradient of the same conv: da [M][K] = dq [M][N] . W [K][N]^T  (the entry point's N = conv K, its K = conv N)
        dq = dev(rs.normal(size=(M, N)).astype(np.float32))
        bnstate = st                                               # [mean | var | scale | shift] of the conv's K input channels
        assert L().crnn_gemm_f32x3_bnstats_supported(M, K, N) == 0
        r = L().crnn_gemm_f32x3_bnstats_rows(M)
        da0, da1, da2 = zeros(M, K), zeros(M, K), zeros(M, K); s0, s1, s2 = zeros(r * 2 * K), zeros(r * 2 * K), zeros(r * 2 * K)
        ok(L().crnn_gemm_f32x3_bnstats(P(dq), P(w), P(da0), M, K, N, P(d), P(bnstate), P(s0), S()))
        ok(L().crnn_gemm_f32x3_bnstats_pl(P(dq), None, 0, P(w), P(wpl), ws, P(da1), M, K, N, P(d), P(bnstate), P(s1), S()))
        assert torch.equal(da1, da0) and torch.equal(s1, s0), "weight planes: data gradient differs: max %g" % float((da1 - da0).abs().max())
        qpl, qs = _planes_of(dq)
        ok(L().crnn_gemm_f32x3_bnstats_pl(P(dq), P(qpl), qs, P(w), P(wpl), ws, P(da2), M, K, N, P(d), P(bnstate), P(s2), S()))
        assert torch.equal(da2, da0) and torch.equal(s2, s0), "both operands as planes: data gradient differs: max %g" % float((da2 - da0).abs().max())
        assert L().crnn_gemm_f32x3_bnstats_pl(P(dq), P(qpl), qs, P(w), None, 0, P(da2), M, K, N, P(d), P(bnstate), P(s2), S()) == -3
    # ragged tiles: refused
    assert L().crnn_pwconv_bnrelu6_fwd_f32x3_pl(P(d), P(st), P(w), P(wpl), ws, P(q1), M - 3, N, K, P(p1), S()) == -3


@pytest.mark.parametrize("M,K,N", [(128 * 40, 128, 256), (128 * 6, 512, 512), (128 * 3 + 5, 256, 64), (128 * 200, 64, 128), (1000, 64, 128)])
def test_two_plane_gemms_carry_sixteen_bits_per_factor(M, K, N):
    """The f32x2 entry points (two bf16 planes per operand, products hi*hi + hi*mid + mid*hi, fp32 accumulation: the parity mode's backward GEMMs) against
    their three-plane forms and an fp64 reference: every result within 2e-5 of the largest magnitude (a product's relative error is <= 3 * 2^-18;
    measured 5e-6), i.e. two orders of magnitude tighter than single-plane bf16 products, with the three-plane result within 2e-6.  Forward with
    statistics, data gradient with the BatchNorm-backward statistics (whole tiles), weight gradient (split reduction), the plain GEMM entry in all
    three modes; whole and ragged tiles; run-to-run identical."""
    rs = np.random.RandomState(M % 1000 + N + K + 5)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); w = dev((rs.normal(size=(K, N)) * 0.1).astype(np.float32))
    g = dev(rs.normal(size=(M, N)).astype(np.float32))
    st = dev(np.concatenate([rs.normal(size=K), rs.uniform(0.5, 2.0, size=K), rs.normal(size=K) * 0.3 + 1.0, rs.normal(size=K) * 0.5 + 0.5]).astype(np.float32))
    a = zeros(M, K)
    ok(L().crnn_bn_act_pool_drop_ex(P(d), P(st), P(a), 1, 1, M, K, 1, 1, 0.0, 0, 0, 0, 0, S()))
    a64, w64, g64 = host(a).astype(np.float64), host(w).astype(np.float64), host(g).astype(np.float64)
    rows = L().crnn_pwconv_stat_rows(M)
    scr = zeros(16 * 1024 * 1024); sb = ctypes.c_size_t(scr.numel() * 4)

    def close(x2, x3, ref, what):
        x2, x3 = host(x2).astype(np.float64), host(x3).astype(np.float64)
        m = np.abs(ref).max()
        e2, e3 = np.abs(x2 - ref).max() / m, np.abs(x3 - ref).max() / m
        assert e3 <= 2e-6 and e2 <= 2e-5, (what, e2, e3)
        return e2
    # forward + statistics
    q3, q2 = zeros(M, N), torch.full((M + 2, N), 7.0, device="cuda"); p3, p2 = zeros(rows, 2, N), zeros(rows, 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd_f32x3(P(d), P(st), P(w), P(q3), M, N, K, P(p3), S()))
    for rep in range(2):
        ok(L().crnn_pwconv_bnrelu6_fwd_f32x2(P(d), P(st), P(w), P(q2), M, N, K, P(p2), S()))
        if rep == 0: first = q2.clone()
        else: assert torch.equal(first, q2)
    assert bool((q2[M:] == 7.0).all())
    e = close(q2[:M], q3, a64 @ w64, "forward")
    s2, s3 = host(p2).astype(np.float64).sum(0), host(p3).astype(np.float64).sum(0)
    assert np.abs(s2 - s3).max() <= 1e-4 * np.abs(s3).max(), "statistics"
    # weight gradient (producer prologue, split reduction)
    dw3, dw2 = zeros(K, N), zeros(K, N)
    ok(L().crnn_pwconv_bnrelu6_wgrad_f32x3(P(d), P(st), P(g), P(dw3), M, N, K, P(scr), sb, S()))
    ok(L().crnn_pwconv_bnrelu6_wgrad_f32x2(P(d), P(st), P(g), P(dw2), M, N, K, P(scr), sb, S()))
    close(dw2, dw3, a64.T @ g64, "weight gradient")
    # plain entry: the unfused forms of the same three products
    for mode, (A, Bm, Cs, m_, n_, k_, lda, ldb) in {0: (a, w, (M, N), M, N, K, K, N), 1: (g, w, (M, K), M, K, N, N, N), 2: (a, g, (K, N), K, N, M, K, N)}.items():
        c3, c2 = zeros(*Cs), zeros(*Cs)
        ok(L().crnn_gemm_f32x3(mode, P(A), P(Bm), P(c3), m_, n_, k_, lda, ldb, n_, None, 0, 0, 0, P(scr), sb, S()))
        ok(L().crnn_gemm_f32x2(mode, P(A), P(Bm), P(c2), m_, n_, k_, lda, ldb, n_, None, 0, 0, 0, P(scr), sb, S()))
        ref = a64 @ w64 if mode == 0 else g64 @ w64.T if mode == 1 else a64.T @ g64
        close(c2, c3, ref, "plain mode %d" % mode)
    # data gradient with the BatchNorm-backward statistics (whole tiles only)
    if L().crnn_gemm_f32x3_bnstats_supported(M, K, N) == 0:
        r = L().crnn_gemm_f32x3_bnstats_rows(M)
        da3, da2 = zeros(M, K), zeros(M, K); t3, t2 = zeros(r * 2 * K), zeros(r * 2 * K)
        ok(L().crnn_gemm_f32x3_bnstats(P(g), P(w), P(da3), M, K, N, P(d), P(st), P(t3), S()))
        ok(L().crnn_gemm_f32x2_bnstats(P(g), P(w), P(da2), M, K, N, P(d), P(st), P(t2), S()))
        close(da2, da3, g64 @ w64.T, "data gradient")
        b2, b3 = host(t2).astype(np.float64).reshape(r, 2, K).sum(0), host(t3).astype(np.float64).reshape(r, 2, K).sum(0)
        assert np.abs(b2 - b3).max() <= 1e-4 * np.abs(b3).max() + 1e-6, "BatchNorm-backward statistics"
    else:
        assert L().crnn_gemm_f32x2_bnstats(P(g), P(w), P(zeros(M, K)), M, K, N, P(d), P(st), P(zeros(8)), S()) == -3
    assert e > 1e-8, "the two-plane form returned the three-plane bits: not exercised"


def _window_major(t):
    """[B][H][W][C] -> rows in 2x2-window-major order: pixel (y, x) is row ((y/2)(W/2) + x/2) 4 + (y&1) 2 + (x&1) of its image."""
    B, H, W, C = t.shape
    return t.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C).contiguous()


@pytest.mark.parametrize("B,H,W,K,N,pool", [(2, 8, 36, 128, 256, 4), (5, 104, 36, 128, 256, 4), (3, 52, 18, 256, 512, 2), (1, 2, 18, 256, 512, 2),
                                            (7, 6, 36, 128, 128, 4)])
def test_weights_resident_inference_conv_pools_in_its_epilogue(B, H, W, K, N, pool):
    """crnn_pwconv_fwd_wres_folded_pool (predict path, the CRNN's two pooled blocks: MaxPooling2D after the block's ReLU6 taken over groups of
    2 | 4 consecutive rows in the MFMA waves' epilogue, the un-pooled map never written) equals max-pooling the output of
    crnn_pwconv_fwd_wres_folded, value for value (rounding to bf16 is monotonic) -- for (2,2) on rows in 2x2-window-major order.  Ragged last
    stripes, one and several channel slices, memory behind the output untouched, repeated launches, unsupported shapes refused."""
    rs = np.random.RandomState(B * 131 + H + W + K + N)
    M = B * H * W
    a = torch.from_numpy(_bf16_round(np.abs(rs.normal(size=(B, H, W, K))) * 1.2).astype(np.float32)).cuda().bfloat16()
    Wd = _to_bf16_dev(_bf16_round(rs.normal(size=(N, K)) * 0.2))
    st = dev(np.concatenate([rs.normal(size=N), rs.uniform(0.5, 2.0, size=N), rs.normal(size=N) * 0.3 + 1.0, rs.normal(size=N) * 0.5 + 0.5]))
    assert L().crnn_pwconv_fwd_wres_folded_pool_supported(M, N, K, pool) == 0
    full = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_pwconv_fwd_wres_folded(P(a), P(Wd), P(full), M, N, K, P(st), S()))
    full = full.view(B, H, W, N).float()
    if pool == 4:
        ref = full.view(B, H // 2, 2, W // 2, 2, N).amax(dim=(2, 4)); ain = _window_major(a)
    else:
        ref = full.view(B, H, W // 2, 2, N).amax(dim=3); ain = a
    y = torch.full((M // pool + 3, N), 7.0, dtype=torch.bfloat16, device="cuda")
    for rep in range(2):
        ok(L().crnn_pwconv_fwd_wres_folded_pool(P(ain), P(Wd), P(y), M, N, K, P(st), pool, S()))
        got = y[:M // pool].float().view(ref.shape)
        assert torch.equal(got, ref), "pooled epilogue differs: max %g" % float((got - ref).abs().max())
        assert bool((y[M // pool:] == 7.0).all())
    assert float(ref.max()) > 0.5 and float((ref == 0).float().mean()) < 0.9          # the comparison saw real values
    assert L().crnn_pwconv_fwd_wres_folded_pool_supported(M, N, K, 3) == -3 and L().crnn_pwconv_fwd_wres_folded_pool_supported(M, N, 512, 2) == -3
    assert L().crnn_pwconv_fwd_wres_folded_pool(P(ain), P(Wd), P(y), M, N, K, P(st), 6 - pool, S()) == -3


@pytest.mark.parametrize("B,H,W,C", [(2, 104, 36, 128), (3, 52, 18, 256), (2, 104, 36, 64)])
def test_row_stream_depthwise_writes_window_major_rows(B, H, W, C):
    """crnn_dwconv3x3_fwd_stream_ex(out_order = 1): the folded inference form with its rows in 2x2-window-major order is the NHWC result,
    permuted -- bit for bit; the training forms refuse the order."""
    rs = np.random.RandomState(B + H + W + C + 1)
    xd, kd = _to_bf16_dev(_bf16_round(rs.normal(size=(B, H, W, C)))), dev(rs.normal(size=(3, 3, C)))
    st = dev(np.concatenate([rs.normal(size=C), 1 + rs.uniform(size=C), 1 + 0.3 * rs.normal(size=C), 0.5 * rs.normal(size=C) + 1.0]))
    o0 = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda"); o1 = torch.full((B * H * W * C + 64,), 7.0, dtype=torch.bfloat16, device="cuda")
    ok(L().crnn_dwconv3x3_fwd_stream_ex(P(xd), P(kd), P(o0), None, P(st), B, H, W, C, 0, 0, S()))
    ok(L().crnn_dwconv3x3_fwd_stream_ex(P(xd), P(kd), P(o1), None, P(st), B, H, W, C, 0, 1, S()))
    assert torch.equal(o1[:-64].view(B, H, W, C).view(torch.int16), _window_major(o0).view(torch.int16)) and bool((o1[-64:] == 7.0).all())
    assert float(o0.float().abs().max()) > 0
    parts = zeros(L().crnn_dwconv_fwd_stream_rows(B, H, W, C), 2, C)
    assert L().crnn_dwconv3x3_fwd_stream_ex(P(xd), P(kd), P(o1), P(parts), None, B, H, W, C, 0, 1, S()) != 0


@pytest.mark.parametrize("M,N,K,two", [(13312, 128, 1024, True), (13312, 256, 1024, True), (64 * 3, 256, 768, True), (64, 128, 64, False), (64 * 9, 128, 128, False)])
def test_streaming_nt_gemm_of_the_recurrent_input_gradient_equals_the_tile_kernel(M, N, K, two):
    """crnn_gemm_nt_f32_stream (dX = dZf Wf^T + dZb Wb^T in one launch: result of a 64-row stripe in the MFMA waves' registers over both
    reductions, fp32 dZ rounded to bf16 and bf16 weight rows staged through registers into an LDS ring) against two crnn_gemm_bf16_ex
    mode-1 launches (the second accumulating) and the fp64 product of the rounded operands: fp32 summation round-off.  The layers'
    shapes, GRU's 3u columns, one pair, one chunk; repeated launches give the same bits."""
    rs = np.random.RandomState(M % 997 + N + K)
    A0, A1 = rs.normal(size=(M, K)), rs.normal(size=(M, K)); W0, W1 = _bf16_round(rs.normal(size=(N, K)) * 0.2), _bf16_round(rs.normal(size=(N, K)) * 0.2)
    A0d, A1d, W0d, W1d = dev(A0), dev(A1), _to_bf16_dev(W0), _to_bf16_dev(W1)
    Y1 = torch.full((M, N), 7.0, device="cuda"); Y3 = torch.full((M, N), 5.0, device="cuda")
    ok(L().crnn_gemm_nt_f32_stream(P(A0d), P(W0d), P(A1d) if two else None, P(W1d) if two else None, P(Y1), M, N, K, K, K, N, S()))
    ok(L().crnn_gemm_nt_f32_stream(P(A0d), P(W0d), P(A1d) if two else None, P(W1d) if two else None, P(Y3), M, N, K, K, K, N, S()))
    assert torch.equal(Y1, Y3)
    Y2 = zeros(M, N); scr = zeros(16 << 20)
    ok(L().crnn_gemm_bf16_ex(1, P(A0d), P(W0d), P(Y2), M, N, K, K, K, N, None, 0, 0, 0, P(scr), ctypes.c_size_t(scr.numel() * 4), 0, 1, 0, S()))
    if two: ok(L().crnn_gemm_bf16_ex(1, P(A1d), P(W1d), P(Y2), M, N, K, K, K, N, None, 0, 1, 0, P(scr), ctypes.c_size_t(scr.numel() * 4), 0, 1, 0, S()))
    ref = _bf16_round(A0) @ W0.T + (_bf16_round(A1) @ W1.T if two else 0.0)
    tol = 2e-5 * np.abs(ref).max() + 1e-6 * np.sqrt(K)
    assert_close(host(Y1), host(Y2), rtol=1e-4, atol=tol, what="stream vs tile kernel")
    assert_close(host(Y1), ref, rtol=1e-4, atol=tol, what="stream vs fp64")
    assert L().crnn_gemm_nt_f32_stream(P(A0d), P(W0d), None, None, P(Y1), M, 192, K, K, K, 192, S()) == -3


@pytest.mark.parametrize("M,N,K,bias", [(8192, 1024, 128, True), (8192, 1024, 256, True), (64 * 5, 384, 64, True), (64 * 3, 768, 192, False), (64, 128, 64, True)])
def test_streaming_input_projection_equals_the_tile_kernel(M, N, K, bias):
    """crnn_gemm_nt_f32_stream_bias (xw = X W + b of the recurrent layers from the bf16 W^T copy: column slabs of 256, or 128 when
    N % 256 != 0, bias added to the finished fp32 sums) against crnn_gemm_bf16_ex mode 0 on the fp32 [K][N] weights (same bf16 rounding
    of both operands, bias in the epilogue) and the fp64 product of the rounded operands.  The layers' shapes (G = 4u = 1024, din = 128 /
    256), the GRU's 3u columns, a single stripe; the surrounding memory stays untouched; repeated launches give the same bits."""
    rs = np.random.RandomState(M % 997 + N + K)
    A = rs.normal(size=(M, K)); W = _bf16_round(rs.normal(size=(K, N)) * 0.2); b = rs.normal(size=(N,)) if bias else None
    Ad, Wd, WTd = dev(A), dev(W), _to_bf16_dev(np.ascontiguousarray(W.T)); bd = dev(b) if bias else None
    Y1 = torch.full((M + 1, N), 7.0, device="cuda"); Y3 = torch.full((M, N), 5.0, device="cuda")
    ok(L().crnn_gemm_nt_f32_stream_bias(P(Ad), P(WTd), None, None, P(Y1), P(bd) if bias else None, M, N, K, K, K, N, S()))
    ok(L().crnn_gemm_nt_f32_stream_bias(P(Ad), P(WTd), None, None, P(Y3), P(bd) if bias else None, M, N, K, K, K, N, S()))
    assert torch.equal(Y1[:M], Y3) and bool((Y1[M:] == 7.0).all())
    Y2 = zeros(M, N); scr = zeros(16 << 20)
    ok(L().crnn_gemm_bf16_ex(0, P(Ad), P(Wd), P(Y2), M, N, K, K, N, N, P(bd) if bias else None, 0, 0, 0, P(scr), ctypes.c_size_t(scr.numel() * 4), 0, 0, 0, S()))
    ref = _bf16_round(A) @ W + (b if bias else 0.0)
    tol = 2e-5 * np.abs(ref).max() + 1e-6 * np.sqrt(K)
    assert_close(host(Y3), host(Y2), rtol=1e-4, atol=tol, what="stream vs tile kernel")
    assert_close(host(Y3), ref, rtol=1e-4, atol=tol, what="stream vs fp64")
    assert L().crnn_gemm_nt_f32_stream_bias(P(Ad), P(WTd), None, None, P(Y3), None, M, 192, K, K, K, 192, S()) == -3


@pytest.mark.parametrize("M,N,K,bias", [(13312, 1024, 128, True), (13312, 1024, 256, True), (64 * 7, 768, 256, True), (64, 256, 64, False), (64 * 300, 512, 192, True),
                                         (64 * 33, 1024, 128, False)])
def test_both_directions_input_projection_equals_the_two_stripe_launches(M, N, K, bias):
    """crnn_rnn_input_proj (round 5): x W + b of both directions of a Bidirectional layer in ONE launch of persistent workgroups (256-column weight slab resident in
    LDS, 64-row stripes walked per workgroup) against two crnn_gemm_nt_f32_stream_bias launches: the same bf16 products in the same order -- bit for bit --, and
    against the fp64 oracle at the bf16 operands' tolerance; memory around the outputs untouched; shapes outside the rule refused."""
    rs = np.random.RandomState(M % 97 + N + K)
    x = rs.normal(size=(M, K)).astype(np.float32)
    wf, wb = (rs.normal(size=(N, K)) * 0.1).astype(np.float32), (rs.normal(size=(N, K)) * 0.1).astype(np.float32)
    bf, bb = rs.normal(size=N).astype(np.float32), rs.normal(size=N).astype(np.float32)
    xd = dev(x); wfd, wbd = _to_bf16_dev(wf), _to_bf16_dev(wb); bfd, bbd = dev(bf), dev(bb)
    assert L().crnn_rnn_input_proj_supported(M, N, K) == 0
    y1f, y1b = zeros(M, N), zeros(M, N)
    ok(L().crnn_gemm_nt_f32_stream_bias(P(xd), P(wfd), None, None, P(y1f), P(bfd) if bias else None, M, N, K, K, K, N, S()))
    ok(L().crnn_gemm_nt_f32_stream_bias(P(xd), P(wbd), None, None, P(y1b), P(bbd) if bias else None, M, N, K, K, K, N, S()))
    y2f = torch.full((M * N + 64,), 7.0, device="cuda"); y2b = torch.full((M * N + 64,), 9.0, device="cuda")
    ok(L().crnn_rnn_input_proj(P(xd), P(wfd), P(wbd), P(bfd) if bias else None, P(bbd) if bias else None, P(y2f), P(y2b), M, N, K, K, K, N, S()))
    assert torch.equal(y2f[:-64].view(M, N), y1f) and torch.equal(y2b[:-64].view(M, N), y1b)
    assert bool((y2f[-64:] == 7.0).all()) and bool((y2b[-64:] == 9.0).all())
    ref = _bf16_round(x) @ _bf16_round(wf).T + (bf if bias else 0.0)
    assert_close(host(y2f[:-64]).reshape(M, N), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max(), what="forward direction vs oracle on the bf16-rounded operands")
    y3f = torch.zeros_like(y2f); y3b = torch.zeros_like(y2b)
    ok(L().crnn_rnn_input_proj(P(xd), P(wfd), P(wbd), P(bfd) if bias else None, P(bbd) if bias else None, P(y3f), P(y3b), M, N, K, K, K, N, S()))
    assert torch.equal(y3f[:-64], y2f[:-64]) and torch.equal(y3b[:-64], y2b[:-64]), "repeat launches differ"
    assert L().crnn_rnn_input_proj_supported(M, N, 320) == -3 and L().crnn_rnn_input_proj_supported(M, 384, K) == -3 and L().crnn_rnn_input_proj_supported(M + 1, N, K) == -3
    assert L().crnn_rnn_input_proj(P(xd), P(wfd), P(wbd), P(bfd), None, P(y3f), P(y3b), M, N, K, K, K, N, S()) == -2      # one bias without the other


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_streaming_pointwise_kernels_on_random_shapes(seed):
    """The three streaming pointwise-conv kernels on drawn shapes (stripe counts around the grid size, 1..8 channel slices, every K,
    ragged last ranges): forward (q bit for bit, statistics to round-off), data gradient (bit for bit), weight gradient (round-off),
    each against the tile GEMM entry points it replaces; repeated launches bit-identical."""
    rs = np.random.RandomState(1000 + seed)
    K = int(rs.choice([64, 128, 256, 512])); N = 128 * int(rs.randint(1, 5)); M = 128 * int(rs.randint(1, 900))
    d = _bf16_round(rs.normal(size=(M, K)) * 2.0); W = _bf16_round(rs.normal(size=(N, K)) * 0.2); g = _bf16_round(rs.normal(size=(M, N)))
    st = dev(np.concatenate([rs.normal(size=K) * 0.3, rs.uniform(0.5, 2.0, size=K), rs.normal(size=K) * 0.3 + 1.0, rs.normal(size=K) * 0.5 + 1.0]))
    dd, Wd, gd = _to_bf16_dev(d), _to_bf16_dev(W), _to_bf16_dev(g)
    scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda"); nb = ctypes.c_size_t(scratch.numel() * 4)
    # forward
    rows = L().crnn_pwconv_fwd_wres_rows(M, N, K)
    q1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); q1b = torch.zeros_like(q1); p1 = zeros(rows, 2, N); p1b = zeros(rows, 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd_wres(P(dd), P(st), P(Wd), P(q1), M, N, K, P(p1), S()))
    ok(L().crnn_pwconv_bnrelu6_fwd_wres(P(dd), P(st), P(Wd), P(q1b), M, N, K, P(p1b), S()))
    q2 = torch.zeros_like(q1); p2 = zeros(L().crnn_pwconv_stat_rows(M), 2, N)
    ok(L().crnn_pwconv_bnrelu6_fwd(P(dd), P(st), P(Wd), P(q2), M, N, K, P(p2), 1, 1, S()))
    assert torch.equal(q1, q2) and torch.equal(q1, q1b) and torch.equal(p1, p1b), (M, N, K)
    t1, t2 = host(p1).sum(0), host(p2).sum(0)
    assert_close(t1, t2, rtol=3e-5, atol=3e-5 * np.abs(t2).max(), what="forward statistics %r" % ((M, N, K),))
    # data gradient: da[M][K] = g[M][N] . Wt[K][N]^T
    Wt = _to_bf16_dev(np.ascontiguousarray(W.T))
    if L().crnn_gemm_wres_supported(K, N) == 0:
        a1 = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"); a2 = torch.zeros_like(a1)
        ok(L().crnn_gemm_wres_bf16(P(gd), P(Wt), P(a1), M, K, N, S()))
        ok(L().crnn_gemm_bf16_ex(1, P(gd), P(Wt), P(a2), M, K, N, N, N, K, None, 0, 0, 0, None, 0, 1, 1, 1, S()))
        assert torch.equal(a1, a2), (M, N, K)
    # weight gradient
    if L().crnn_pwconv_wgrad_stream_supported(M, N, K) == 0:
        w1 = zeros(K, N); w1b = zeros(K, N); w2 = zeros(K, N)
        ok(L().crnn_pwconv_bnrelu6_wgrad_stream(P(dd), P(st), P(gd), P(w1), M, N, K, P(scratch), nb, S()))
        ok(L().crnn_pwconv_bnrelu6_wgrad_stream(P(dd), P(st), P(gd), P(w1b), M, N, K, P(scratch), nb, S()))
        ok(L().crnn_pwconv_bnrelu6_wgrad(P(dd), P(st), P(gd), P(w2), M, N, K, P(scratch), nb, S()))
        assert torch.equal(w1, w1b)
        assert_close(host(w1), host(w2), rtol=1e-4, atol=(2e-5 + 6e-8 * M / 64) * np.abs(host(w2)).max(), what="weight gradient %r" % ((M, N, K),))


def test_bilstm_bf16_recurrent_weights_track_the_fp64_cell():
    """crnn_lstm_fwd_ex / crnn_lstm_bwd_ex with dt_u = bf16: recurrent products on the bf16 MFMA (weights stored bf16, the
    state rounded to bf16 as it is packed).  Against the fp64 cell evaluated with the SAME bf16-rounded weights the only
    difference is the per-step rounding of h / dz to bf16 (2^-9 relative), which the recurrence keeps bounded."""
    B, T, u, din = 20, 9, 128, 40
    rs = np.random.RandomState(31)
    x = rs.normal(size=(B, T, din)); G = 4 * u
    Wt = [rs.normal(size=(din, G)) * 0.3 for _ in range(2)]
    U = [_bf16_round(rs.normal(size=(u, G)) * 0.15) for _ in range(2)]
    bb = [rs.normal(size=G) * 0.2 for _ in range(2)]
    Hs, caches = [], []
    for d in range(2):
        h, c = ops.lstm_fwd(x, Wt[d], U[d], bb[d], reverse=(d == 1))
        Hs.append(h); caches.append(c)
    tm = lambda a: np.ascontiguousarray(np.swapaxes(a, 0, 1))
    xw = [dev(tm(x @ Wt[d] + bb[d])) for d in range(2)]
    ut = [_to_bf16_dev(U[d].T) for d in range(2)]
    hcat = zeros(T, B, 2 * u); cs = [zeros(T, B, u) for _ in range(2)]; gt = [zeros(T, B, G) for _ in range(2)]
    ok(L().crnn_lstm_fwd_ex(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), ctypes.c_void_p(hcat.data_ptr() + 4 * u), 2 * u, P(cs[0]), P(cs[1]),
                            P(gt[0]), P(gt[1]), T, B, u, 1, S()))
    hh = host(hcat)
    for d in range(2):
        assert_close(hh[:, :, d * u:(d + 1) * u], tm(Hs[d]), rtol=0, atol=1e-2, what=f"h dir{d} (bf16 recurrent products)")
        assert_close(host(cs[d]), tm(caches[d][4]), rtol=0, atol=2e-2, what=f"c dir{d}")
    gH = rs.normal(size=(B, T, 2 * u))
    gd = dev(tm(gH))
    dz = [zeros(T, B, G) for _ in range(2)]; dc = [zeros(B, u) for _ in range(2)]
    Ud = [_to_bf16_dev(U[d]) for d in range(2)]
    ok(L().crnn_lstm_bwd_ex(P(Ud[0]), P(Ud[1]), P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), P(gd), ctypes.c_void_p(gd.data_ptr() + 4 * u), 2 * u,
                            P(dz[0]), P(dz[1]), P(dc[0]), P(dc[1]), T, B, u, 1, S()))
    for d in range(2):
        dzh = np.swapaxes(host(dz[d]).astype(np.float64), 0, 1)
        dx, dW, dU, db = ops.lstm_bwd(caches[d], gH[..., d * u:(d + 1) * u])
        ref = dx; got = dzh @ Wt[d].T
        cos = float((ref.ravel() @ got.ravel()) / (np.linalg.norm(ref) * np.linalg.norm(got)))
        assert cos > 0.999, (d, cos)
        # a bf16-sized nudge flips a few hard-sigmoid saturation decisions between the device forward (whose gates the
        # device backward uses) and the fp64 forward, so single elements may differ by more: bound the bulk, not the max
        err = np.abs(got - ref)
        assert np.percentile(err, 99) < 1e-2 * np.abs(ref).max(), (d, np.percentile(err, 99), np.abs(ref).max())
    # unsupported widths are refused, not silently computed in another precision
    assert L().crnn_lstm_fwd_ex(P(xw[0]), P(xw[1]), P(ut[0]), P(ut[1]), P(hcat), P(hcat), 2 * u, P(cs[0]), P(cs[1]), P(gt[0]), P(gt[1]), T, B, 64, 1, S()) == -3


def test_ctc_and_decoders_at_the_size_limits_and_with_empty_labels():
    """Edge cases of the decode / loss kernels: the largest alphabet (64 classes) and label (31) the library accepts, a long
    IAM-like sequence (T = 102), empty labels (label_length = 0), a one-step input, and out-of-range sizes refused."""
    B, T, C, Lmax = 7, 102, 64, 31
    rs = np.random.RandomState(123)
    logits = rs.normal(size=(B, T, C)) * 2.5
    y = ops.softmax_fwd(logits)
    ll = np.array([31, 0, 1, 17, 0, 31, 5])
    labels = np.full((B, Lmax), C - 1, dtype=np.int64)
    for b in range(B):
        labels[b, :ll[b]] = rs.randint(0, C - 1, size=ll[b])
    labels[5, :31] = 7                                    # 31 repeats need 61 frames: feasible only for T-2 >= 61
    il = np.array([T - 2, T - 2, 3, 60, 1, T - 2, T - 2], dtype=np.int64)
    loss_ref, gy = ctc.ctc_loss_and_grad(y, labels, il, ll)
    gl_ref = ops.softmax_bwd(y, gy / B)
    yd = dev(y)
    loss = zeros(B); dl = zeros(T, B, C)
    ok(L().crnn_ctc_loss_grad(P(yd), P(dev(labels, np.int32)), P(dev(il, np.int32)), P(dev(ll, np.int32)), P(loss), P(dl), B, T, C, Lmax, 2,
                              1.0 / B, S()))
    lh = host(loss)
    assert np.array_equal(np.isfinite(lh), np.isfinite(loss_ref))
    fin = np.isfinite(loss_ref)
    assert_close(lh[fin], loss_ref[fin], rtol=1e-4, atol=1e-3, what="ctc loss at C=64, L=31, T=102")
    assert_close(np.swapaxes(host(dl), 0, 1), gl_ref, rtol=1e-3, atol=2e-6, what="dlogits")
    # decoders on the same posteriors (float32 as the model emits them)
    yp = y.astype(np.float32)
    ilf = np.array([T, 1, 2, 60, T, 33, T])
    ref, rl = ctc.ctc_greedy_decode(yp, ilf)
    out = zeros(B, T, dtype=torch.int32); ln = zeros(B, dtype=torch.int32)
    ok(L().crnn_ctc_greedy_decode(P(dev(yp)), P(dev(ilf, np.int32)), P(out), P(ln), B, T, C, S()))
    assert np.array_equal(host(out), ref) and np.array_equal(host(ln), rl)
    bref, brl, bsc = ctc.ctc_beam_decode(yp, beam_width=16, merge_repeated=True, input_length=ilf)
    out = zeros(B, T, dtype=torch.int32); ln = zeros(B, dtype=torch.int32); sc = zeros(B)
    ok(L().crnn_ctc_beam_decode(P(dev(yp)), P(dev(ilf, np.int32)), P(out), P(ln), P(sc), B, T, C, 16, 1, S()))
    assert np.array_equal(host(ln), brl) and np.array_equal(host(out), bref)
    assert_close(host(sc), bsc, rtol=1e-4, atol=1e-3, what="beam score")
    # limits are enforced, not overrun
    assert L().crnn_ctc_beam_decode(P(dev(yp)), None, P(out), P(ln), P(sc), B, T, C, 65, 1, S()) != 0      # beam wider than a wavefront
    assert L().crnn_ctc_loss_grad(P(yd), P(dev(labels, np.int32)), P(dev(il, np.int32)), P(dev(ll, np.int32)), P(loss), P(dl), B, T, C, 32, 2,
                                  1.0 / B, S()) != 0                                                          # 2*32+1 > 64 lanes


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_inference_epilogues_fold_batchnorm_and_relu6(storage):
    """Inference forms (BN scale/shift known up front): crnn_dwconv3x3_bn_relu6_fwd and crnn_pwconv_fwd(out_bnstate)
    must equal conv -> BatchNorm(inference) -> ReLU6 done as separate oracle steps (utils.py:44-51)."""
    rs = np.random.RandomState(5)
    B, H, W, C, N = 3, 13, 9, 64, 128
    bf = storage == "bf16"
    x = rs.normal(size=(B, H, W, C)); k = rs.normal(size=(3, 3, C)) * 0.4
    if bf:
        x = _bf16_round(x)
    gam, bet = rs.uniform(0.5, 1.5, C), rs.normal(size=C) * 0.3
    mm, mv = rs.normal(size=C) * 0.2, rs.uniform(0.5, 2.0, C)
    st = zeros(4 * C)
    ok(L().crnn_bn_infer_state(P(dev(mm)), P(dev(mv)), P(dev(gam)), P(dev(bet)), C, P(st), S()))
    ref_d = ops.dwconv_fwd(x, k)
    ref_a = ops.relu6_fwd(ops.bn_infer_fwd(ref_d, gam, bet, mm, mv))
    xd = _to_bf16_dev(x) if bf else dev(x)
    out = torch.zeros(B, H, W, C, dtype=torch.bfloat16 if bf else torch.float32, device="cuda")
    ok(L().crnn_dwconv3x3_bn_relu6_fwd(P(xd), P(dev(k.reshape(9, C))), P(st), P(out), B, H, W, C, int(bf), S()))
    got = _f(out) if bf else host(out)
    assert_close(got, ref_a, rtol=2.0 ** -8 if bf else 1e-5, atol=2e-2 if bf else 1e-5, what="dw + folded BN + ReLU6")
    # pointwise conv with the folded BatchNorm + ReLU6 of its output
    a = got.reshape(-1, C).astype(np.float64)                 # what the next layer really reads
    Wp = rs.normal(size=(C, N)) * 0.2
    if bf:
        Wp = _bf16_round(Wp)
    g2, b2 = rs.uniform(0.5, 1.5, N), rs.normal(size=N) * 0.3
    m2, v2 = rs.normal(size=N) * 0.2, rs.uniform(0.5, 2.0, N)
    st2 = zeros(4 * N)
    ok(L().crnn_bn_infer_state(P(dev(m2)), P(dev(v2)), P(dev(g2)), P(dev(b2)), N, P(st2), S()))
    ref_x = ops.relu6_fwd(ops.bn_infer_fwd((a @ Wp).reshape(B, H, W, N), g2, b2, m2, v2)).reshape(-1, N)
    M_ = a.shape[0]
    if bf:
        q = torch.zeros(M_, N, dtype=torch.bfloat16, device="cuda")
        ok(L().crnn_pwconv_fwd(P(out), P(_to_bf16_dev(Wp)), P(q), M_, N, C, None, P(st2), 1, 1, 1, 1, 0, S()))
        assert_close(_f(q), ref_x, rtol=2.0 ** -7, atol=3e-2, what="pw + folded BN + ReLU6 (bf16)")
    else:
        q = zeros(M_, N)
        ok(L().crnn_pwconv_fwd(P(out), P(dev(Wp)), P(q), M_, N, C, None, P(st2), 0, 0, 0, 0, 0, S()))
        assert_close(host(q), ref_x, rtol=1e-4, atol=1e-4, what="pw + folded BN + ReLU6 (fp32)")
    # statistics and the folded BatchNorm are mutually exclusive
    assert L().crnn_pwconv_fwd(P(out), P(dev(Wp)), P(q), M_, N, C, P(zeros(8, 2, N)), P(st2), 0, 0, 0, 0, 0, S()) != 0


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_single_channel_pointwise_conv_outer_product_kernels(storage):
    """crnn_pw1_fwd / crnn_pw1_bwd (block 1: one input channel) against the plain matrix form."""
    rs = np.random.RandomState(8)
    Mm, N = 128 * 3 + 45, 64
    bf = storage == "bf16"
    a = rs.normal(size=Mm); w = rs.normal(size=N) * 0.5
    ref = np.outer(a.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64))
    rows = L().crnn_pwconv_stat_rows(Mm)
    parts = zeros(rows, 2, N)
    q = torch.zeros(Mm, N, dtype=torch.bfloat16 if bf else torch.float32, device="cuda")
    ok(L().crnn_pw1_fwd(P(dev(a)), P(dev(w)), P(q), Mm, N, P(parts), int(bf), S()))
    got = _f(q) if bf else host(q)
    assert_close(got, ref, rtol=2.0 ** -8 if bf else 1e-6, atol=1e-6, what="outer product")
    pr = host(parts).astype(np.float64); g64 = got.astype(np.float64)
    for t in range(rows):
        blk = g64[128 * t: 128 * (t + 1)]
        assert_close(pr[t, 0], blk.sum(0), rtol=1e-5, atol=1e-4, what="tile sum"); assert_close(pr[t, 1], (blk * blk).sum(0), rtol=1e-5, atol=1e-4, what="tile sumsq")
    dq = rs.normal(size=(Mm, N))
    if bf:
        dq = _bf16_round(dq)
    dqd = _to_bf16_dev(dq) if bf else dev(dq)
    da = zeros(Mm); dw = zeros(N); scr = zeros(L().crnn_colreduce_chunks(Mm) * N)
    ok(L().crnn_pw1_bwd(P(dev(a)), P(dev(w)), P(dqd), P(da), P(dw), P(scr), Mm, N, int(bf), S()))
    assert_close(host(da), dq @ w.astype(np.float32).astype(np.float64), rtol=1e-5, atol=1e-5, what="da")
    assert_close(host(dw), a.astype(np.float32).astype(np.float64) @ dq, rtol=1e-5, atol=1e-4, what="dw")
    assert L().crnn_pw1_fwd(P(dev(a)), P(dev(w)), P(q), Mm, 48, None, int(bf), S()) == -3        # N must be a power of two
    # inference form: the BatchNorm + ReLU6 after the convolution folded in
    scale, shift = rs.normal(size=N) * 0.3 + 1.0, rs.normal(size=N) * 0.5 + 1.0
    st = dev(np.concatenate([rs.normal(size=N), rs.uniform(0.5, 2.0, size=N), scale, shift]))
    y = torch.full((Mm, N), 9.0, dtype=torch.bfloat16 if bf else torch.float32, device="cuda")
    ok(L().crnn_pw1_fwd_folded(P(dev(a)), P(dev(w)), P(y), Mm, N, P(st), int(bf), S()))
    yref = np.clip(ref * scale.astype(np.float32) + shift.astype(np.float32), 0.0, 6.0)
    assert_close(_f(y) if bf else host(y), yref, rtol=2.0 ** -8 if bf else 1e-6, atol=1e-5, what="folded outer product")


@pytest.mark.parametrize("cin,H,W", [(1, 50, 16), (20, 23, 6), (20, 9, 7)])
def test_localisation_net_direct_conv_kernels(cin, H, W):
    """crnn_loc_conv_fwd / _wgrad / _dgrad: the 5x5 'valid' convolutions of the spatial transformer's localisation net
    (utils.py:248-252) against the im2col restatement."""
    rs = np.random.RandomState(cin + H)
    B = 7
    x = rs.normal(size=(B, H, W, cin)); k = rs.normal(size=(5, 5, cin, 20)) * 0.2; b = rs.normal(size=20) * 0.1
    ref = ops.conv_valid_fwd(x, k, b)
    y = zeros(B, H - 4, W - 4, 20)
    ok(L().crnn_loc_conv_fwd(P(dev(x)), P(dev(k)), P(dev(b)), P(y), B, H, W, cin, S()))
    assert_close(host(y), ref, rtol=1e-5, atol=1e-5, what="conv fwd")
    gy = rs.normal(size=ref.shape)
    dx_ref, dk_ref, db_ref = ops.conv_valid_bwd(x, k, gy)
    chunks = L().crnn_loc_conv_wgrad_chunks(B, H, W)
    grads = zeros(25 * cin * 20 + 20)                       # [dk | db] adjacent, as in the parameter layout
    scr = zeros((chunks + 1) * (25 * cin * 20 + 20))
    ok(L().crnn_loc_conv_wgrad(P(dev(x)), P(dev(gy)), P(grads), ctypes.c_void_p(grads.data_ptr() + 4 * 25 * cin * 20), P(scr), B, H, W, cin, S()))
    g = host(grads)
    assert_close(g[:25 * cin * 20].reshape(k.shape), dk_ref, rtol=1e-4, atol=1e-4, what="dk")
    assert_close(g[25 * cin * 20:], db_ref, rtol=1e-4, atol=1e-4, what="db")
    # separate dk / db buffers take the copy path
    dk2 = zeros(25 * cin * 20); db2 = zeros(20)
    ok(L().crnn_loc_conv_wgrad(P(dev(x)), P(dev(gy)), P(dk2), P(db2), P(scr), B, H, W, cin, S()))
    assert_close(host(dk2).reshape(k.shape), dk_ref, rtol=1e-4, atol=1e-4, what="dk (split)"); assert_close(host(db2), db_ref, rtol=1e-4, atol=1e-4, what="db (split)")
    if cin == 20:
        dx = zeros(B, H, W, cin)
        ok(L().crnn_loc_conv_dgrad(P(dev(gy)), P(dev(k)), P(dx), B, H, W, S()))
        assert_close(host(dx), dx_ref, rtol=1e-4, atol=1e-5, what="dx")
    assert L().crnn_loc_conv_fwd(P(dev(x)), P(dev(k)), P(dev(b)), P(y), B, H, W, 3, S()) == -3


@pytest.mark.parametrize("B,H,W,N,bf", [(3, 104, 36, 64, True), (256, 104, 36, 64, True), (5, 44, 36, 64, False), (2, 204, 36, 64, False), (1, 7, 5, 64, True)])
def test_block1_kernels_with_batchnorm1_folded_in(B, H, W, N, bf):
    """crnn_dwconv3x3_c1_fwd, crnn_pw1_bn_fwd, crnn_pw1_bn_bwd (round 4): block 1's single-channel depthwise conv with its own statistics, the outer
    product applying BatchNorm-1 + ReLU6 to d on the way in, and ONE backward pass over dq for weight gradient, data gradient and BatchNorm-1's backward
    statistics -- against the stand-alone kernels (outputs bit for bit; the statistics the same sums in another order) and the fp64 oracle."""
    rs = np.random.RandomState(B + H + W)
    x = rs.normal(size=(B, H, W, 1)); k = rs.normal(size=(3, 3, 1)); w = rs.normal(size=N) * 0.3
    xd, kd, wd = dev(x), dev(k), dev(w)
    M = B * H * W
    # depthwise conv + statistics
    d_ref = zeros(B, H, W, 1)
    ok(L().crnn_dwconv3x3_fwd_ex(P(xd), P(kd), P(d_ref), None, B, H, W, 1, 0, 0, S()))
    rows = L().crnn_dwconv_c1_stat_rows(B, H, W)
    d = torch.full((M + 8,), 7.0, device="cuda"); parts = torch.full((rows + 1, 2), 3.0, device="cuda")
    ok(L().crnn_dwconv3x3_c1_fwd(P(xd), P(kd), P(d), P(parts), B, H, W, S()))
    assert torch.equal(d[:-8], d_ref.reshape(-1)) and bool((d[-8:] == 7.0).all()) and bool((parts[rows] == 3.0).all())
    dn = host(d_ref).reshape(-1)
    assert_close(host(parts[:rows]).sum(0), np.array([dn.sum(), (dn ** 2).sum()]), rtol=1e-4, atol=1e-3, what="statistics")
    assert_close(dn.reshape(B, H, W, 1), ops.dwconv_fwd(x, k), what="d vs oracle")
    # outer product of relu6(BN(d))
    mean, var = dn.mean(), dn.var(); gamma, beta = 1.3, 1.1
    scale = gamma / np.sqrt(var + 1e-3); st = dev(np.array([mean, var, scale, beta - mean * scale]))
    a = zeros(M)
    ok(L().crnn_bn_act_pool_drop_ex(P(d_ref), P(st), P(a), 1, 1, M, 1, 1, 1, 0.0, 0, 0, 0, 0, S()))
    dtq = 1 if bf else 0
    mk = (lambda *sh: torch.zeros(*sh, dtype=torch.bfloat16, device="cuda")) if bf else zeros
    srows = L().crnn_pwconv_stat_rows(M)
    q1 = mk(M, N); p1 = zeros(srows, 2, N); q2 = mk(M, N); p2 = zeros(srows, 2, N)
    ok(L().crnn_pw1_fwd(P(a), P(wd), P(q1), M, N, P(p1), dtq, S()))
    ok(L().crnn_pw1_bn_fwd(P(d_ref), P(st), P(wd), P(q2), M, N, P(p2), dtq, S()))
    assert torch.equal(q1.view(torch.int16 if bf else torch.int32), q2.view(torch.int16 if bf else torch.int32))
    if bf:
        assert torch.equal(p1, p2)                      # statistics of the values as stored (re-rounded): the same sums
    else:                                               # fp32: the compiler may contract product + sum differently in the two kernels (last-bit differences)
        assert_close(host(p2), host(p1), rtol=1e-5, atol=1e-5 * float(p1.abs().max()), what="BatchNorm-2 statistic partials")
    # backward: one pass over dq
    dq = (torch.randn(M, N, device="cuda", generator=torch.Generator("cuda").manual_seed(3)) * 0.5)
    dq = dq.to(torch.bfloat16) if bf else dq
    chunks = L().crnn_colreduce_chunks(M)
    da1 = zeros(M); dw1 = zeros(N); sc1 = zeros(chunks * N)
    ok(L().crnn_pw1_bwd(P(a), P(wd), P(dq), P(da1), P(dw1), P(sc1), M, N, dtq, S()))
    brow = L().crnn_pw1_bn_bwd_rows(M)
    assert brow == chunks
    da2 = torch.full((M + 8,), 9.0, device="cuda"); dw2 = zeros(N); sc2 = zeros(brow * N); bp = torch.full((brow + 1, 2), 5.0, device="cuda")
    ok(L().crnn_pw1_bn_bwd(P(d_ref), P(st), P(wd), P(dq), P(da2), P(dw2), P(sc2), P(bp), M, N, dtq, S()))
    assert torch.equal(da2[:-8], da1) and torch.equal(dw2, dw1) and bool((da2[-8:] == 9.0).all()) and bool((bp[brow] == 5.0).all())
    # ... and the BatchNorm-1 backward statistics against the stand-alone statistics pass
    dg1, db1, coef1 = zeros(1), zeros(1), zeros(2); gin = zeros(M); pp = zeros(max(L().crnn_bn_bwd_chunks(M), 1) * 2 + 64); gam = dev(np.array([gamma]))
    ok(L().crnn_bn_bwd_ex(P(d_ref), P(da1), P(st), P(gam), P(gin), P(dg1), P(db1), P(pp), P(coef1), B, H, W, 1, 1, 1, 0.0, 0, 0, 0, S()))
    dg2, db2, coef2 = zeros(1), zeros(1), zeros(2)
    ok(L().crnn_bn_bwd_finalize(P(bp), brow, 1, M, P(dg2), P(db2), P(coef2), S()))
    for u, v, nm in ((db2, db1, "sum gy"), (dg2, dg1, "sum gy xhat"), (coef2, coef1, "coefficients")):
        assert_close(host(u), host(v), rtol=2e-4, atol=2e-5 * max(1.0, float(v.abs().max())), what=nm)
    # the depthwise stage backwards in one kernel: BatchNorm-1 backward pass 2 (gin above) + weight gradient + data gradient
    dk1 = zeros(9); scw = zeros(max(L().crnn_dwconv_num_tiles(B, H, W) * 9, 1024 * 9) + 64)
    ok(L().crnn_dwconv3x3_wgrad_ex(P(xd), P(gin), P(dk1), P(scw), B, H, W, 1, 0, S()))
    dx1 = zeros(M)
    ok(L().crnn_dwconv3x3_fwd_ex(P(gin), P(kd), P(dx1), None, B, H, W, 1, 1, 0, S()))
    crow = L().crnn_dwconv_c1_bwd_rows(B, H, W)
    dx2 = torch.full((M + 8,), 9.0, device="cuda"); dk2 = zeros(9); sc3 = torch.full((crow * 9 + 8,), 5.0, device="cuda")
    ok(L().crnn_dwconv3x3_c1_bwd(P(d_ref), P(da1), P(st), P(coef1), P(xd), P(kd), P(dx2), P(dk2), P(sc3), B, H, W, S()))
    assert torch.equal(dx2[:-8], dx1), "dx: max diff %g" % float((dx2[:-8] - dx1).abs().max())
    assert bool((dx2[-8:] == 9.0).all()) and bool((sc3[-8:] == 5.0).all())
    assert_close(host(dk2), host(dk1), rtol=1e-4, atol=1e-5 * max(1.0, float(dk1.abs().max())), what="dk")
    dk3 = zeros(9)
    ok(L().crnn_dwconv3x3_c1_bwd(P(d_ref), P(da1), P(st), P(coef1), P(xd), P(kd), None, P(dk3), P(sc3), B, H, W, S()))
    assert torch.equal(dk3, dk2)


@pytest.mark.parametrize("B,H0,W0", [(7, 100, 32), (256, 100, 32), (3, 200, 32), (5, 40, 32), (2, 60, 48)])
def test_localisation_net_in_one_workgroup_per_sample(B, H0, W0):
    """crnn_loc_net_fwd / crnn_loc_net_bwd (round 4): the spatial transformer's localisation net (utils.py:248-256) with one workgroup per sample.
    Forward: pool1, c1, pool2, flat, fc1, theta bit-identical to the five stand-alone kernels.  Backward from dtheta: all eight parameter gradients
    against the stand-alone sequence (crnn_loc_fc_bwd, crnn_loc_conv_wgrad, crnn_loc_conv_dgrad, crnn_maxpool_bwd, crnn_loc_conv_wgrad) to summation
    order, repeat launches bit-identical; shapes the kernels refuse say so."""
    rs = np.random.RandomState(B + H0 + W0)
    Hs1, Ws1 = H0 // 2, W0 // 2; Ho1, Wo1 = Hs1 - 4, Ws1 - 4; Hs2, Ws2 = Ho1 // 2, Wo1 // 2; Ho2, Wo2 = Hs2 - 4, Ws2 - 4; F = Ho2 * Wo2 * 20
    sup = L().crnn_loc_net_fused_supported(H0, W0)
    assert sup == (0 if (Ho1 % 2 == 0 and Wo1 % 2 == 0 and Ho2 >= 1 and Wo2 >= 1) else -3)
    x = dev(rs.uniform(size=(B, H0, W0)))
    k1 = dev(rs.normal(size=(5, 5, 1, 20)) * 0.3); bc1 = dev(rs.normal(size=20) * 0.1)
    k2 = dev(rs.normal(size=(5, 5, 20, 20)) * 0.1); bc2 = dev(rs.normal(size=20) * 0.1)
    w1 = dev(rs.normal(size=(max(F, 1), 50)) * 0.1); b1 = dev(rs.normal(size=50) * 0.1); w2 = dev(rs.normal(size=(50, 6)) * 0.3); b2 = dev(rs.normal(size=6))
    if sup != 0:
        z = zeros(16)
        assert L().crnn_loc_net_fwd(P(x), P(k1), P(bc1), P(k2), P(bc2), P(w1), P(b1), P(w2), P(b2), P(z), P(z), P(z), P(z), P(z), P(z), B, H0, W0, S()) == -3
        return
    # stand-alone sequence
    p1 = zeros(B, Hs1, Ws1); c1 = zeros(B, Ho1, Wo1, 20); p2 = zeros(B, Hs2, Ws2, 20); fl = zeros(B, F); f1 = zeros(B, 50); th = zeros(B, 6)
    ok(L().crnn_maxpool_fwd(P(x), P(p1), B, H0, W0, 1, 2, 2, S()))
    ok(L().crnn_loc_conv_fwd(P(p1), P(k1), P(bc1), P(c1), B, Hs1, Ws1, 1, S()))
    ok(L().crnn_maxpool_fwd(P(c1), P(p2), B, Ho1, Wo1, 20, 2, 2, S()))
    ok(L().crnn_loc_conv_fwd(P(p2), P(k2), P(bc2), P(fl), B, Hs2, Ws2, 20, S()))
    ok(L().crnn_loc_fc_fwd(P(fl), P(w1), P(b1), P(w2), P(b2), P(f1), P(th), B, F, S()))
    q = [torch.full((t.numel() + 8,), 7.0, device="cuda") for t in (p1, c1, p2, fl, f1, th)]
    ok(L().crnn_loc_net_fwd(P(x), P(k1), P(bc1), P(k2), P(bc2), P(w1), P(b1), P(w2), P(b2), *[P(t) for t in q], B, H0, W0, S()))
    for t, ref, nm in zip(q, (p1, c1, p2, fl, f1, th), ("pool1", "c1", "pool2", "flat", "fc1", "theta")):
        assert torch.equal(t[:-8], ref.reshape(-1)), "%s: max diff %g" % (nm, float((t[:-8] - ref.reshape(-1)).abs().max()))
        assert bool((t[-8:] == 7.0).all()), nm
    assert float(f1.abs().max()) > 0 and float((f1 == 0).float().mean()) > 0.05
    # backward from dtheta
    dth = dev(rs.normal(size=(B, 6)))
    g1 = dict(dfc1=zeros(B, 50), dflat=zeros(B, F), dw1=zeros(F, 50), db1=zeros(50), dw2=zeros(50, 6), db2=zeros(6), dk2=zeros(5, 5, 20, 20), dbc2=zeros(20),
              dp2=zeros(B, Hs2, Ws2, 20), dc1=zeros(B, Ho1, Wo1, 20), dk1=zeros(5, 5, 1, 20), dbc1=zeros(20))
    ok(L().crnn_loc_fc_bwd(P(fl), P(f1), P(dth), P(w1), P(w2), P(g1["dfc1"]), P(g1["dflat"]), P(g1["dw1"]), P(g1["db1"]), P(g1["dw2"]), P(g1["db2"]), B, F, S()))
    scr = zeros((max(L().crnn_loc_conv_wgrad_chunks(B, Hs2, Ws2), L().crnn_loc_conv_wgrad_chunks(B, Hs1, Ws1)) + 2) * (25 * 20 * 20 + 20))
    ok(L().crnn_loc_conv_wgrad(P(p2), P(g1["dflat"]), P(g1["dk2"]), P(g1["dbc2"]), P(scr), B, Hs2, Ws2, 20, S()))
    ok(L().crnn_loc_conv_dgrad(P(g1["dflat"]), P(k2), P(g1["dp2"]), B, Hs2, Ws2, S()))
    ok(L().crnn_maxpool_bwd(P(c1), P(g1["dp2"]), P(g1["dc1"]), B, Ho1, Wo1, 20, 2, 2, S()))
    ok(L().crnn_loc_conv_wgrad(P(p1), P(g1["dc1"]), P(g1["dk1"]), P(g1["dbc1"]), P(scr), B, Hs1, Ws1, 1, S()))
    names = ("dk1", "dbc1", "dk2", "dbc2", "dw1", "db1", "dw2", "db2")
    def fused():
        g = {n: torch.full((g1[n].numel() + 4,), 5.0, device="cuda") for n in names}
        dfc1 = zeros(B, 50); terms = torch.full((L().crnn_loc_net_bwd_scratch(B) + 8,), 3.0, device="cuda")
        ok(L().crnn_loc_net_bwd(P(dth), P(fl), P(f1), P(p1), P(c1), P(p2), P(w1), P(w2), P(k2), P(dfc1), P(terms), *[P(g[n]) for n in names], B, H0, W0, S()))
        assert bool((terms[-8:] == 3.0).all()) and all(bool((g[n][-4:] == 5.0).all()) for n in names)
        return g, dfc1
    g2, dfc1 = fused()
    assert torch.equal(dfc1, g1["dfc1"])
    for n in names:
        ref = host(g1[n]).reshape(-1)
        assert_close(host(g2[n][:-4]), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())), what=n)
    g3, _ = fused()
    assert all(torch.equal(g2[n], g3[n]) for n in names), "repeat launches differ"


@pytest.mark.parametrize("B,F", [(12, 760), (3, 80)])
def test_localisation_net_dense_kernels(B, F):
    """crnn_loc_fc_fwd / crnn_loc_fc_bwd: Dense(50, relu) -> Dense(6) (utils.py:254-255) and their gradients."""
    rs = np.random.RandomState(B)
    flat = rs.normal(size=(B, F)); w1 = rs.normal(size=(F, 50)) * 0.1; b1 = rs.normal(size=50) * 0.1
    w2 = rs.normal(size=(50, 6)) * 0.3; b2 = rs.normal(size=6)
    pre = ops.dense_fwd(flat, w1, b1); h = np.maximum(pre, 0); th = ops.dense_fwd(h, w2, b2)
    fc1 = zeros(B, 50); theta = zeros(B, 6)
    ok(L().crnn_loc_fc_fwd(P(dev(flat)), P(dev(w1)), P(dev(b1)), P(dev(w2)), P(dev(b2)), P(fc1), P(theta), B, F, S()))
    assert_close(host(fc1), h, rtol=1e-5, atol=1e-5, what="fc1"); assert_close(host(theta), th, rtol=1e-5, atol=1e-5, what="theta")
    dth = rs.normal(size=(B, 6))
    dh, dw2_ref, db2_ref = ops.dense_bwd(h, w2, dth)
    dpre = dh * (pre > 0)
    dflat_ref, dw1_ref, db1_ref = ops.dense_bwd(flat, w1, dpre)
    dfc1 = zeros(B, 50); dflat = zeros(B, F); dw1 = zeros(F, 50); db1 = zeros(50); dw2 = zeros(50, 6); db2 = zeros(6)
    ok(L().crnn_loc_fc_bwd(P(dev(flat)), P(fc1), P(dev(dth)), P(dev(w1)), P(dev(w2)), P(dfc1), P(dflat), P(dw1), P(db1), P(dw2), P(db2), B, F, S()))
    for got, ref, nm in ((dfc1, dpre, "dfc1"), (dflat, dflat_ref, "dflat"), (dw1, dw1_ref, "dW1"), (db1, db1_ref, "db1"), (dw2, dw2_ref, "dW2"), (db2, db2_ref, "db2")):
        assert_close(host(got), ref, rtol=1e-4, atol=1e-5, what=nm)


# ------------------------------------------------------------------------------------------------ round 6: gradients as planes
def _split_planes(x, n_planes=3):
    """crnn_split3_planes of a device fp32 tensor -> int16 tensor [3 * n] (plane stride n)."""
    n = x.numel()
    pl = torch.empty(3 * n, dtype=torch.int16, device="cuda")
    ok(L().crnn_split3_planes(P(x), P(pl), n, n, S()))
    return pl


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("M,N,K", [(64 * 9, 128, 256), (64 * 250, 128, 256), (64 * 41, 256, 256), (64 * 150, 256, 512), (64 * 20, 512, 512), (64 * 700, 512, 512),
                                   (64 * 1, 128, 512), (64 * 66, 128, 512), (256 * 52 * 9, 512, 512), (256 * 52 * 18, 256, 256)])   # (the last two: blocks 6 / 7 and 4 at batch 256)
def test_data_gradient_from_planes_equals_the_tile_kernel_bit_for_bit(M, N, K, planes):
    """crnn_gemm_pres_bnstats (round 6, gemm_pres.hip): da = dq . W^T with dq given as pre-split bf16 planes (LDS-DMA, no split in the GEMM), the planes of W
    resident in up to 512 registers per wave, one wave per SIMD.  The same planes, products and single fp32 accumulation chain per result as the tile kernel:
    da EQUALS crnn_gemm_f32x2 / _f32x3 (mode 1) bit for bit; the BatchNorm-1 backward statistics are the tile kernel's sums in another order (vs fp64);
    the planes of a = ReLU6(d * scale + shift) the drain can write equal crnn_split3_planes of crnn_bn_act_pool_drop_ex's output word for word.  Workgroups
    with no stripe, one stripe (prologue + final drain only), stripes below and above the ring's warm-up; repeated launches give the same bits; the memory
    behind the outputs is untouched."""
    if planes == 3 and K == 512:
        assert L().crnn_gemm_pres_supported(M, N, K, 3) == -3
        return
    rs = np.random.RandomState(M % 1000 + N + K + 13 * planes)
    dq = dev((rs.normal(size=(M, K)) * 1e-2).astype(np.float32)); W = dev((rs.normal(size=(N, K)) * 0.1).astype(np.float32))
    dh = (rs.normal(size=(M, N)) * 1.5 + 0.3).astype(np.float32); d = dev(dh)
    gamma = rs.uniform(0.5, 1.5, N); beta = rs.normal(size=N) * 0.5 + 1.0
    mean = dh.astype(np.float64).mean(0); var = dh.astype(np.float64).var(0)
    inv = 1.0 / np.sqrt(var.astype(np.float32) + np.float32(1e-3))
    scale = (gamma * inv).astype(np.float32); shift = (beta - mean * gamma * inv).astype(np.float32)
    bnstate = dev(np.concatenate([mean, var, scale, shift]).astype(np.float32))
    assert L().crnn_gemm_pres_supported(M, N, K, planes) == 0
    rows = L().crnn_gemm_pres_stat_rows(M, N, K, planes)
    assert 0 < rows <= 512
    pl = _split_planes(dq)
    da0 = zeros(M, N)
    ok((L().crnn_gemm_f32x3 if planes == 3 else L().crnn_gemm_f32x2)(1, P(dq), P(W), P(da0), M, N, K, K, K, N, None, 0, 0, 0, None, 0, S()))
    for emit in (0, planes):
        p1 = torch.full((rows + 1, 2, N), float("nan"), device="cuda"); da1 = torch.full((M + 2, N), 7.0, device="cuda")
        ap = torch.full((emit * M * N + 8,), 0x1234, dtype=torch.int16, device="cuda") if emit else None
        for rep in range(2):
            ok(L().crnn_gemm_pres_bnstats(P(pl), M * K, P(W), P(da1), M, N, K, planes, P(d), P(bnstate), P(p1), P(ap), M * N, emit, S()))
            if rep == 0: keep = (da1.clone(), p1.clone())
        assert torch.equal(da1, keep[0]) and torch.equal(p1[:rows], keep[1][:rows])
        assert bool((da1[M:] == 7.0).all()) and bool(torch.isfinite(p1[:rows]).all()) and bool(torch.isnan(p1[rows:]).all())
        assert torch.equal(da1[:M], da0), "da differs from the tile kernel: %g" % float((da1[:M] - da0).abs().max())
        g = host(da1[:M]).astype(np.float64)
        t = dh * scale + shift
        gy = np.where((t > 0) & (t < 6), g, 0.0)
        xhat = (dh.astype(np.float64) - mean.astype(np.float32).astype(np.float64)) * inv.astype(np.float64)
        s1 = host(p1[:rows]).astype(np.float64).sum(0)
        assert_close(s1[0], gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="sum gy vs fp64")
        assert_close(s1[1], (gy * xhat).sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy * xhat).sum(0).max(), what="sum gy xhat vs fp64")
        if emit:
            a = zeros(M, N)
            ok(L().crnn_bn_act_pool_drop_ex(P(d), P(bnstate), P(a), 1, 1, M, N, 1, 1, 0.0, 0, 0, 0, 0, S()))
            ref = _split_planes(a)
            assert torch.equal(ap[:emit * M * N], ref[:emit * M * N]) and bool((ap[emit * M * N:] == 0x1234).all())
    dg, db, coef = zeros(N), zeros(N), zeros(2 * N); fold = zeros(32 * 2 * N)
    ok(L().crnn_bn_bwd_finalize_folded(P(p1), rows, N, M, P(dg), P(db), P(coef), P(fold), S()))
    assert_close(host(db), gy.sum(0), rtol=1e-4, atol=1e-4 * np.abs(gy).sum(0).max(), what="dbeta vs fp64")
    assert L().crnn_gemm_pres_supported(M, N, 128, planes) == -3 and L().crnn_gemm_pres_supported(M + 16, N, K, planes) == -3 and L().crnn_gemm_pres_supported(M, 64, K, planes) == -3
    assert L().crnn_gemm_pres_bnstats(P(pl), M * K, P(W), P(da1), M, N, K, planes, None, P(bnstate), P(p1), None, 0, 0, S()) == -2
    assert L().crnn_gemm_pres_bnstats(P(pl), M * K, P(W), P(da1), M, N, K, 4, P(d), P(bnstate), P(p1), None, 0, 0, S()) == -2


@pytest.mark.parametrize("B,H,W,C,ph,pw,rate", [(2, 8, 12, 64, 1, 1, 0.0), (3, 8, 12, 128, 2, 2, 0.2), (2, 6, 20, 256, 1, 2, 0.2), (1, 9, 7, 32, 1, 1, 0.2), (2, 18, 52, 512, 1, 1, 0.0)])
@pytest.mark.parametrize("planes", [2, 3])
def test_batchnorm_backward_writes_its_input_gradient_as_planes(B, H, W, C, ph, pw, rate, planes):
    """crnn_bn_bwd_planes_ex / crnn_bn_bwd_apply_planes_ex (round 6): the fp32 BatchNorm backward (through dropout, max-pool and ReLU6) with dx written as bf16
    planes -- the words crnn_split3_planes forms from the dx crnn_bn_bwd_ex writes, bit for bit; dgamma, dbeta and coef unchanged; nothing written behind the
    planes; bad arguments rejected."""
    rs = np.random.RandomState(B * 131 + H * 17 + C + ph + 3 * pw)
    Ho, Wo = H // ph, W // pw
    x = dev((rs.normal(size=(B, H, W, C)) * 1.2 + 0.5).astype(np.float32)); g = dev(rs.normal(size=(B, Ho, Wo, C)).astype(np.float32))
    st = _bnstate(rs, C); gamma = dev(rs.uniform(0.5, 1.5, C).astype(np.float32))
    n = B * H * W * C
    chunks = L().crnn_bn_bwd_chunks(B * H * W)
    def run(planes_out):
        dgm, dbt, coef, pp = zeros(C), zeros(C), zeros(2 * C), zeros(chunks * 2 * C + 64)
        if planes_out is None:
            dx = zeros(B, H, W, C)
            ok(L().crnn_bn_bwd_ex(P(x), P(g), P(st), P(gamma), P(dx), P(dgm), P(dbt), P(pp), P(coef), B, H, W, C, ph, pw, rate, 77, 3, 0, S()))
            return dx, dgm, dbt, coef
        ok(L().crnn_bn_bwd_planes_ex(P(x), P(g), P(st), P(gamma), P(planes_out), n, planes, P(dgm), P(dbt), P(pp), P(coef), B, H, W, C, ph, pw, rate, 77, 3, S()))
        return planes_out, dgm, dbt, coef
    dx, dg0, db0, coef0 = run(None)
    ref = _split_planes(dx)
    out = torch.full((planes * n + 16,), 0x4321, dtype=torch.int16, device="cuda")
    _, dg1, db1, coef1 = run(out)
    assert torch.equal(out[:planes * n], ref[:planes * n]) and bool((out[planes * n:] == 0x4321).all())
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(coef0, coef1)
    out2 = torch.full((planes * n + 16,), 0x4321, dtype=torch.int16, device="cuda")
    ok(L().crnn_bn_bwd_apply_planes_ex(P(x), P(g), P(st), P(coef0), P(out2), n, planes, B, H, W, C, ph, pw, rate, 77, 3, S()))
    assert torch.equal(out2, out)
    assert L().crnn_bn_bwd_apply_planes_ex(P(x), P(g), P(st), P(coef0), P(out2), n, 4, B, H, W, C, ph, pw, rate, 77, 3, S()) == -2
    assert L().crnn_bn_bwd_apply_planes_ex(P(x), P(g), P(st), P(coef0), P(out2), n - 4, planes, B, H, W, C, ph, pw, rate, 77, 3, S()) == -2
    assert L().crnn_bn_bwd_apply_planes_ex(P(x), P(g), P(st), P(coef0), None, n, planes, B, H, W, C, ph, pw, rate, 77, 3, S()) == -2


@pytest.mark.parametrize("M,K,N,bn", [(32 * 7, 128, 128, True), (32 * 900, 128, 256, True), (32 * 500, 256, 512, True), (32 * 1200, 512, 512, True), (32 * 333, 512, 128, False)])
def test_weight_gradient_stream_reads_the_planes_of_g(M, K, N, bn):
    """crnn_pwconv_bnrelu6_wgrad_planes_stream_gp: g given as the two planes crnn_bn_bwd_planes_ex writes -- the IO waves copy the words the fp32 form splits:
    the same result bit for bit."""
    rs = np.random.RandomState(M % 1000 + N + K + 5)
    d = dev((rs.normal(size=(M, K)) * 1.5 + 0.4).astype(np.float32)); g = dev((rs.normal(size=(M, N)) * 1e-2).astype(np.float32))
    st = _bnstate(rs, K)
    nb = L().crnn_pwconv_wgrad_planes_stream_scratch_bytes(M, N, K)
    scr = torch.empty(nb // 4, device="cuda")
    dw0 = torch.full((K + 1, N), 7.0, device="cuda"); dw1 = torch.full((K + 1, N), 7.0, device="cuda")
    ok(L().crnn_pwconv_bnrelu6_wgrad_planes_stream(P(d), P(st) if bn else None, P(g), P(dw0), M, N, K, P(scr), ctypes.c_size_t(nb), S()))
    gp = _split_planes(g)
    ok(L().crnn_pwconv_bnrelu6_wgrad_planes_stream_gp(P(d), P(st) if bn else None, P(gp), M * N, P(dw1), M, N, K, P(scr), ctypes.c_size_t(nb), S()))
    assert torch.equal(dw0, dw1) and bool((dw1[K:] == 7.0).all())
    assert L().crnn_pwconv_bnrelu6_wgrad_planes_stream_gp(P(d), None, None, M * N, P(dw1), M, N, K, P(scr), ctypes.c_size_t(nb), S()) == -2


@pytest.mark.parametrize("B,H,W,C,ph,pw,rate", [(3, 8, 12, 128, 2, 2, 0.2), (2, 6, 20, 256, 1, 2, 0.2), (2, 36, 52, 64, 2, 2, 0.0), (5, 18, 26, 24, 1, 2, 0.3)])
@pytest.mark.parametrize("dtype", [0, 1])
def test_pooled_batchnorm_backward_statistics_from_the_saved_window_maxima(B, H, W, C, ph, pw, rate, dtype):
    """Pooled blocks, round 6: crnn_bn_act_pool_drop_qmax_ex writes the same y as crnn_bn_act_pool_drop_ex plus qmax = x at the first maximum of x * scale + shift
    over each pool window (checked against a numpy scan of the window in the kernel's order); crnn_bn_bwd_qmax_ex's statistics pass then reads one value per
    window instead of the window -- dgamma, dbeta, coef and dx equal crnn_bn_bwd_ex's BIT FOR BIT (fp32 and bf16 tensors, with the planes output too); other
    windows are refused."""
    rs = np.random.RandomState(B * 131 + H * 17 + C + ph + 3 * pw + dtype)
    Ho, Wo = H // ph, W // pw
    xh = (rs.normal(size=(B, H, W, C)) * 1.2 + 0.5).astype(np.float32)
    if dtype == 1: xh = host(dev(xh).to(torch.bfloat16).float())           # bf16-representable values
    st = _bnstate(rs, C); gamma = dev(rs.uniform(0.5, 1.5, C).astype(np.float32))
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    x = dev(xh).to(tdt); g = dev(rs.normal(size=(B, Ho, Wo, C)).astype(np.float32)).to(tdt)
    y0 = torch.zeros(B, Ho, Wo, C, device="cuda", dtype=tdt); y1 = torch.zeros_like(y0); qm = torch.full((B * Ho * Wo * C + 16,), 9.0, device="cuda", dtype=tdt)
    ok(L().crnn_bn_act_pool_drop_ex(P(x), P(st), P(y0), B, H, W, C, ph, pw, rate, 77, 3, dtype, dtype, S()))
    ok(L().crnn_bn_act_pool_drop_qmax_ex(P(x), P(st), P(y1), P(qm), B, H, W, C, ph, pw, rate, 77, 3, dtype, dtype, S()))
    assert torch.equal(y0, y1) and bool((qm[B * Ho * Wo * C:] == 9.0).all())
    sth = host(st).reshape(4, C)
    t = (xh.astype(np.float64) * sth[2].astype(np.float64) + sth[3].astype(np.float64)).astype(np.float32)   # (the comparison below only needs the arg-max)
    tw = t.reshape(B, Ho, ph, Wo, pw, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, ph * pw, C)
    xw = xh.reshape(B, Ho, ph, Wo, pw, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, ph * pw, C)
    am = tw.argmax(axis=3)                                                   # first maximum
    ref = np.take_along_axis(xw, am[:, :, :, None, :], axis=3)[:, :, :, 0, :]
    got = host(qm[:B * Ho * Wo * C].float()).reshape(B, Ho, Wo, C)
    srt = np.sort(tw, axis=3)
    clear = (srt[..., -1, :] - srt[..., -2, :]) > 1e-5                        # windows whose maximum is not a near-tie of the float64 restatement
    assert np.array_equal(got[clear], ref[clear]) and clear.mean() > 0.95
    chunks = L().crnn_bn_bwd_chunks(B * H * W)
    def run(qmax, planes):
        dgm, dbt, coef, pp = zeros(C), zeros(C), zeros(2 * C), zeros(chunks * 2 * C + 64)
        n = B * H * W * C
        dx = torch.zeros(B, H, W, C, device="cuda", dtype=tdt) if not planes else torch.zeros(planes * n, dtype=torch.int16, device="cuda")
        ok(L().crnn_bn_bwd_qmax_ex(P(x), P(qmax), P(g), P(st), P(gamma), None if planes else P(dx), P(dx) if planes else None, n, planes, P(dgm), P(dbt), P(pp), P(coef),
                                   B, H, W, C, ph, pw, rate, 77, 3, dtype, S()))
        return dx, dgm, dbt, coef
    for planes in ((0, 2) if dtype == 0 else (0,)):
        a = run(None, planes); b = run(qm, planes)
        for u, v in zip(a, b): assert torch.equal(u, v)
    dgm, dbt, coef, pp = zeros(C), zeros(C), zeros(2 * C), zeros(chunks * 2 * C + 64); dx0 = torch.zeros(B, H, W, C, device="cuda", dtype=tdt)
    ok(L().crnn_bn_bwd_ex(P(x), P(g), P(st), P(gamma), P(dx0), P(dgm), P(dbt), P(pp), P(coef), B, H, W, C, ph, pw, rate, 77, 3, dtype, S()))
    assert torch.equal(dx0, a[0] if dtype == 1 else run(qm, 0)[0]) and torch.equal(dgm, b[1]) and torch.equal(dbt, b[2])
    assert L().crnn_bn_act_pool_drop_qmax_ex(P(x), P(st), P(y1), P(qm), B, H, W, C, 1, 1, rate, 77, 3, dtype, dtype, S()) == -3
    assert L().crnn_bn_act_pool_drop_qmax_ex(P(x), P(st), P(y1), None, B, H, W, C, ph, pw, rate, 77, 3, dtype, dtype, S()) == -2
