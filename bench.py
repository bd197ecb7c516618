#!/usr/bin/env python3
"""bench.py -- text-line images/sec of the full CRNN-OCR train step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank/GPU)

A "step" = forward (train mode: batch statistics, dropout) + CTC loss + backward + [RCCL all-reduce of the
flat fp32 gradient buffer] + global-norm clip + Adam + BatchNorm moving-statistics update on one synthetic
batch that is already resident in HBM.  Workload = BASELINE.json configs[1]: 100x32x1 images, batch 256 per
GPU, max_len 23, time_dense_size 128, n_units 256 (LSTM), Adam(1e-4, beta1 .5, clipnorm 5).  Weak scaling
(global batch 256*N).  Rank 0 prints ONE JSON line.

The timed path is the product only (crnn_mi355x over libcrnn_mi355x.so: weights from crnn_mi355x.init, synthetic batch
generated here); `oracle/` is imported by the `cpu_baseline` leg alone, where it IS the thing being timed.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
PMC_ROUND = "r06"               # the committed counter passes the *_roofline objects quote (profiles/<PMC_ROUND>_pmc_step_<mode>.json, visit r06bo)


def pmc_step_traffic(eng, prefixes, grids=None):
    """Memory-side bytes per train step of the kernels whose (shortened) names start with one of `prefixes`, from the committed counter passes over this very
    command (profiles/r06_pmc_step_<mode>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate counter-only runs, FETCH_SIZE doubled per
    MI355X_MICROARCH.md; scripts/gpu_visit.sh pmc: steps + scripts/pmc_step_summary.py).  Counted in the step's own order and cache state (the requests the
    L2s send to the fabric: last-level-cache hits included).  None when the workload is not the one the passes ran (batch 256, 100x32, LSTM).
    grids: optional {prefix: set of grid sizes in threads} to tell launches of one kernel apart.  -> (bytes per step, launches per step) | None"""
    if eng.B != 256 or (eng.cfg.imgh, eng.cfg.imgw) != (100, 32) or eng.cfg.gru or eng.cfg.flags:
        return None
    path = os.path.join(ROOT, "profiles", "%s_pmc_step_%s.json" % (PMC_ROUND, eng.precision))
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    tot, n = 0.0, 0.0
    for key, v in ks.items():
        name, grid = key.rsplit("|", 1)
        for pfx in prefixes:
            if name.startswith(pfx) and (not grids or pfx not in grids or int(grid) in grids[pfx]):
                tot += v["bytes_per_launch"] * v["calls_per_step"]; n += v["calls_per_step"]
                break
    return (tot, n) if n else None


def pmc_mfma_util(eng, prefixes):
    """MFMA-pipe utilisation by the counters (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average kernel time x 2.4 GHz)) of the step's launches of each prefix."""
    if eng.B != 256 or (eng.cfg.imgh, eng.cfg.imgw) != (100, 32) or eng.cfg.gru or eng.cfg.flags:
        return None
    path = os.path.join(ROOT, "profiles", "%s_pmc_step_%s.json" % (PMC_ROUND, eng.precision))
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    out = []
    for pfx in prefixes:
        v = [x for k, x in ks.items() if k.startswith(pfx) and x.get("mfma_util") is not None and x["calls_per_step"] >= 0.9]
        out.append(round(sum(x["mfma_util"] * x["calls_per_step"] for x in v) / max(1e-9, sum(x["calls_per_step"] for x in v)), 4) if v else None)
    return out


PMC_NOTE = ("memory-side bytes of the same kernels in the train step, per launch set: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE passes over `bench.py "
            "--steps 3 --warmup 2` (profiles/r06_pmc_step_%s.json, profiles/r06_pmc_sq_step_%s.txt: committed constants of visit r06bo, not measured in this run); counted in the step's order and cache state -- the L2s' "
            "requests to the fabric, last-level-cache hits included -- while `achieved` is timed on the re-issued launches")


def depthwise_roofline(eng, iters=15):
    """The HBM-bound kernel north_star singles out: the depthwise 3x3 of blocks 2..7.  bf16s (the headline mode): the six forward
    launches of a step (each with the BatchNorm-statistics epilogue) exactly as the step issues them -- dw_fwd_stream_kernel where its
    shape rule holds (every block of the CRNN), dwconv_tile_kernel<0> otherwise or under CRNN_FLAG_DW_TILE_KERNEL; the data gradient
    lives in the fused depthwise-stage backward (dw_bwd_roofline).  fp32 / bf16 modes: dwconv_tile_kernel<0> forward + data gradient
    (flipped taps) = 12 launches.  Re-issued on the live buffers between events on the launch stream.  Algorithmic bytes per launch
    = read H*W*C + write H*W*C elements per image in the storage type, weights negligible."""
    from crnn_mi355x.engine import _ptr, _stream
    lib = eng.lib
    B = eng.B
    blocks = [(64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)]
    h, w, cin = eng.cfg.imgh + 4, eng.cfg.imgw + 4, 1
    launches, nbytes = [], 0.0
    bf16s = eng.precision == "bf16s"
    esz = 2 if bf16s else 4          # storage bytes per element of the conv-stack tensors
    dtype = 1 if bf16s else 0
    parts = eng.ws_tensor("partials")
    nstream = npro = 0
    rate = 0.1 if eng.cfg.dropout else 0.0
    producers = []       # bf16s: the launch that writes each forward launch's input in the step (previous block's BN + ReLU6 + pool + dropout)
    prev = None          # (h, w, c, ph, pw) of the previous block's pointwise output q
    for i, (co, ph, pw) in enumerate(blocks, 1):
        if i >= 2:
            k = eng.params[eng.layout["b%d_dw" % i][0]:]
            if bf16s:
                qh, qw, qc, qph, qpw = prev
                producers.append((eng.ws_tensor("q%d" % (i - 1)), eng.ws_tensor("bn2s%d" % (i - 1)), eng.ws_tensor("x%d" % (i - 1)), qh, qw, qc, qph, qpw, i - 1))
            st = bf16s and not (eng.cfg.flags & 32) and lib.crnn_dwconv_fwd_stream_supported(B, h, w, cin) == 0
            nstream += int(st)
            # the step's own form: the previous block's BatchNorm-2 + ReLU6 + dropout applied inside the kernel (prologue form, reads q) where the
            # forward does not materialise x (crnn_block_output_fused)
            pro = (eng.ws_tensor("q%d" % (i - 1)), eng.ws_tensor("bn2s%d" % (i - 1)), i - 1) if (st and lib.crnn_block_output_fused(eng._c, i - 1)) else None
            npro += int(pro is not None)
            launches.append((eng.ws_tensor("x%d" % (i - 1)), k, eng.ws_tensor("d%d" % i), parts, h, w, cin, 0, st, pro))   # forward
            nbytes += 2.0 * B * h * w * cin * esz
            if not bf16s:
                launches.append((eng.ws_tensor("gB"), k, eng.ws_tensor("gA"), None, h, w, cin, 1, False, None))            # data gradient
                nbytes += 2.0 * B * h * w * cin * esz
        prev = (h, w, co, ph, pw)
        h, w, cin = h // ph, w // pw, co

    def issue(x, k, o, pt, hh, ww, cc, flip, st, pro):
        if pro is not None:
            lib.crnn_dwconv3x3_fwd_stream_pro(_ptr(pro[0]), _ptr(pro[1]), rate, _ptr(eng.ws_tensor("dm%d" % pro[2])) if rate > 0 else None, _ptr(k), _ptr(o), _ptr(pt), B, hh, ww, cc, _stream())
        elif st:
            lib.crnn_dwconv3x3_fwd_stream(_ptr(x), _ptr(k), _ptr(o), _ptr(pt), None, B, hh, ww, cc, flip, _stream())
        else:
            lib.crnn_dwconv3x3_fwd_ex(_ptr(x), _ptr(k), _ptr(o), _ptr(pt), B, hh, ww, cc, flip, dtype, _stream())
    times = []
    for it in range(iters + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for L in launches:
            issue(*L)
        e1.record()
        torch.cuda.synchronize()
        if it:
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(times))
    ach = nbytes / t / 1e9
    # The same launches in the state the step runs them in: each right after the kernel that writes its input (the 256 MB
    # last-level cache still holds part of it), one event pair per launch (their ~2 us of event latency counted against the kernel)
    in_step = None
    if bf16s and len(producers) == len(launches):
        tot = []
        for it in range(iters + 1):
            evs = []
            for (q, st2, xo, qh, qw, qc, qph, qpw, layer), L in zip(producers, launches):
                if L[9] is None:       # (prologue form: the input q is the pointwise GEMM's output, not re-produced here)
                    lib.crnn_bn_act_pool_drop_ex(_ptr(q), _ptr(st2), _ptr(xo), B, qh, qw, qc, qph, qpw, rate, 1234, layer, 1, 1, _stream())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); issue(*L); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if it:
                tot.append(sum(a.elapsed_time(b) for a, b in evs) * 1e-3)
        ti = float(np.median(tot))
        in_step = {"achieved": round(nbytes / ti / 1e9, 1), "frac": round(nbytes / ti / 1e9 / PEAK_HBM_GBS, 4), "avg_launch_ms": round(1e3 * ti / len(launches), 4)}
    # HBM bytes of the same launch set from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE as separate
    # counter-only runs over scripts/dw_bench.py at batch 256 -- scripts/gpu_visit.sh pmcdw -- folded by scripts/pmc_summary.py
    # and committed under profiles/; FETCH_SIZE x2 per the gfx950 note in MI355X_MICROARCH.md)
    traffic, pmc_file = None, None
    try:
        pmc_file = [f for f in ("r05_pmc_dwconv.json", "r04_pmc_dwconv.json", "r03_pmc_dwconv.json", "r02_pmc_dwconv.json", "r01_pmc_dwconv.json") if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        if B == pmc["batch"] and (eng.cfg.imgh, eng.cfg.imgw) == (100, 32):
            if nstream == len(launches):
                sh = pmc["stream"]["shapes"]
                traffic = sh["104x36x64"]["hbm_bytes_per_launch"] + sh["104x36x128"]["hbm_bytes_per_launch"] \
                    + 2 * sh["52x18x256"]["hbm_bytes_per_launch"] + 2 * sh["52x9x512"]["hbm_bytes_per_launch"]
            elif nstream == 0:
                sh = pmc["modes"]["bf16" if esz == 2 else "fp32"]["shapes"]
                per = len(launches) // 6
                traffic = per * (sh["104x36x64"]["hbm_bytes_per_launch"] + sh["104x36x128"]["hbm_bytes_per_launch"]
                                 + 2 * sh["52x18x256"]["hbm_bytes_per_launch"] + 2 * sh["52x9x512"]["hbm_bytes_per_launch"])
    except Exception:
        pass
    kname = ("dw_fwd_stream_kernel (depthwise 3x3 forward + BatchNorm statistics, blocks 2-7: rows streamed through an LDS ring by a loader wave; %d of the "
             "%d launches in the prologue form: the previous block's BatchNorm-2 + ReLU6 + dropout applied to q in LDS by two transform waves)" % (npro, len(launches))
             if nstream == len(launches) else
             "dwconv_tile_kernel<0> (depthwise 3x3 fwd%s, blocks 2-7, LDS halo tiles)" % ("" if bf16s else " + data-gradient")
             if nstream == 0 else "dw_fwd_stream_kernel + dwconv_tile_kernel<0> (depthwise 3x3 forward, blocks 2-7)")
    # HBM-roofline fraction = the COLD back-to-back figure: the six launches re-issued in a row on the live buffers, every input last
    # touched five launches earlier (1.7 GB working set against the 256 MB last-level cache), launch gaps inside the event pair.  The
    # in-step figure (each launch right after its producer, part of its input still in the last-level cache) is an EFFECTIVE
    # bandwidth, reported beside it -- it is what the timed steps see, not an HBM fraction.
    res = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
           "launches": len(launches), "avg_launch_ms": round(1e3 * t / len(launches), 4), "algorithmic_bytes_per_launch_set": nbytes,
           "traffic": traffic,
           # (round 6) `traffic` is NOT measured by this run: it is a constant read from a committed counter pass
           "traffic_source": None if traffic is None else {"kind": "committed constant, not measured in this run", "file": "profiles/%s" % pmc_file,
                                                           "visit": pmc_file.split("_")[0], "counters": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) / WRITE_SIZE, separate counter-only passes"},
           "traffic_note": None if traffic is None else "HBM bytes per launch set from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the cold "
                           "micro-benchmark (scripts/dw_bench.py; profiles/%s), i.e. the same cache state as `achieved`" % pmc_file,
           "measurement": "HIP events on the launch stream around the launch set re-issued back to back on the live buffers (cold: every input "
                          "last touched five launches earlier), median of %d repetitions" % iters}
    if in_step is not None:
        in_step["note"] = ("effective bandwidth in the step's order and cache state: one event pair per launch, each launch issued right after the "
                           "kernel that writes its input (previous block's BN + ReLU6 + pool + dropout); not an HBM-roofline fraction")
        res["in_step_effective"] = in_step
    # What a plain COPY of the same bytes achieves here: the six (input, output) pairs copied back to back, same cold protocol and events.
    # "sweep" = grid-stride copy (the resident workgroups read one window of the buffer together); "banded" = every workgroup its own
    # contiguous band, the row-stream kernel's pattern (one image per workgroup).  Last: the copies overwrite the forward's outputs.
    try:
        from crnn_mi355x import native as _native
        hooks = _native.hooks()       # measurement hooks live in their own library (libcrnn_testhooks.so), not in the product's
    except Exception:
        hooks = None
    if bf16s and hooks is not None:
        ref = {}
        for name, pattern in (("sweep", 0), ("banded", 1)):
            ts = []
            for it in range(iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for (x, k, o, pt, hh, ww, cc, flip, st, pro) in launches:
                    hooks.crnn_debug_copy(_ptr(x), _ptr(o), B * hh * ww * cc * esz, pattern, 256, _stream())
                e1.record()
                torch.cuda.synchronize()
                if it:
                    ts.append(e0.elapsed_time(e1) * 1e-3)
            tc = float(np.median(ts))
            ref[name] = {"achieved": round(nbytes / tc / 1e9, 1), "frac": round(nbytes / tc / 1e9 / PEAK_HBM_GBS, 4), "kernel_vs_copy": round(tc / t, 3)}
        ref["note"] = ("crnn_debug_copy of the same six tensor pairs (256 workgroups x 256 threads, 16-byte accesses), timed like `achieved`: "
                       "`sweep` = grid-stride, plain loads / stores (nontemporal ones reach 0.74-0.78: profiles/r03_copy_probe.txt), `banded` = one contiguous band per workgroup, the "
                       "access pattern of the row-stream kernel; kernel_vs_copy = copy time / kernel time")
        res["copy_reference"] = ref
    return res


def depthwise_fp32_cold(B, iters=7):
    """The depthwise row-stream pair in the form the PARITY mode runs (fp32 tensors: crnn_dwconv3x3_fwd_stream_dt / crnn_dwconv3x3_bwd_stream_ex), blocks 2..7 at batch B,
    cold protocol of `roofline`: the six launches re-issued back to back on six separate buffer sets (3.4 GB), HIP events on the launch stream, median.
    Algorithmic bytes: forward read + write of H W C fp32 per image, backward three reads (d, da, x) + one write (dx).  -> {fwd: {...}, bwd: {...}} | None"""
    from crnn_mi355x import native
    from crnn_mi355x.engine import _ptr, _stream
    lib = native.lib()
    shapes = [(104, 36, 64), (104, 36, 128), (52, 18, 256), (52, 18, 256), (52, 9, 512), (52, 9, 512)]
    if any(lib.crnn_dwconv_fwd_stream_supported_ex(B, h, w, c, 0) != 0 or lib.crnn_dwconv_bwd_stream_supported_ex(B, h, w, c, 0) != 0 for h, w, c in shapes):
        return None
    bufs, nel = [], 0
    for (h, w, c) in shapes:
        n = B * h * w * c
        x = torch.randn(n, device="cuda"); d = torch.empty_like(x); da = torch.randn(n, device="cuda"); dx = torch.empty_like(x)
        k = torch.randn(9, c, device="cuda"); dk = torch.zeros(9, c, device="cuda")
        st1 = torch.cat([torch.randn(c) * 0.1, 1 + torch.rand(c), 1 + 0.3 * torch.randn(c), 1.0 + 0.5 * torch.randn(c)]).cuda()
        coef = (torch.randn(2 * c) * 1e-3).cuda()
        rows = max(lib.crnn_dwconv_fwd_stream_rows_ex(B, h, w, c, 0) * 2, lib.crnn_dwconv_bwd_stream_rows_ex(B, h, w, c, 0) * 9)
        bufs.append((x, d, da, dx, k, dk, st1, coef, torch.empty(rows * c + 64, device="cuda")))
        nel += n

    def timed(fn):
        ts = []
        for it in range(iters + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for sh, bf in zip(shapes, bufs):
                rc = fn(sh, bf)
                assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            if it:
                ts.append(e0.elapsed_time(e1) * 1e-3)
        return float(np.median(ts))
    tf = timed(lambda sh, bf: lib.crnn_dwconv3x3_fwd_stream_dt(_ptr(bf[0]), _ptr(bf[4]), _ptr(bf[1]), _ptr(bf[8]), B, sh[0], sh[1], sh[2], 0, 0, _stream()))
    tb = timed(lambda sh, bf: lib.crnn_dwconv3x3_bwd_stream_ex(_ptr(bf[1]), _ptr(bf[2]), _ptr(bf[6]), _ptr(bf[7]), _ptr(bf[0]), _ptr(bf[4]), _ptr(bf[3]), _ptr(bf[5]),
                                                                _ptr(bf[8]), B, sh[0], sh[1], sh[2], 0, _stream()))
    out = {}
    for name, t, passes in (("fwd", tf, 2), ("bwd", tb, 4)):
        nb = float(passes) * nel * 4
        out[name] = {"achieved": round(nb / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(nb / t / 1e9 / PEAK_HBM_GBS, 4),
                     "avg_launch_ms": round(1e3 * t / len(shapes), 4), "algorithmic_bytes_per_launch_set": nb}
    out["note"] = ("dw_fwd_stream_kernel / dw_bwd_stream_kernel in their fp32 form (what the parity mode's step launches), six launches of blocks 2-7 back to back on separate "
                   "buffers, cold; the bf16 form is `roofline` / `dw_bwd_roofline`")
    del bufs
    torch.cuda.empty_cache()
    return out


def depthwise_bwd_roofline(eng, iters=5):
    """Secondary: the fused depthwise-stage backward (blocks 2..7, bf16s training; csrc/dwconv_bwd_stream.hip, or csrc/conv_bwd_fused.hip
    where the stream kernel's shape rule fails / under CRNN_FLAG_DW_TILE_KERNEL) on the live buffers, as the step issues it.
    Algorithmic bytes per launch = read d, da, x + write dx = 4 x B*H*W*C elements in the storage type."""
    from crnn_mi355x.engine import _ptr, _stream
    if eng.precision != "bf16s" or (eng.cfg.flags & 16):
        return None
    lib = eng.lib; B = eng.B
    blocks = [(64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)]
    h, w, cin = eng.cfg.imgh + 4, eng.cfg.imgw + 4, 1
    launches, nbytes, nstream, npro = [], 0.0, 0, 0
    rate = 0.1 if eng.cfg.dropout else 0.0
    parts, coef = eng.ws_tensor("partials"), eng.ws_tensor("coef")
    for i, (co, ph, pw) in enumerate(blocks, 1):
        if i >= 2:
            if lib.crnn_dwconv_bwd_fused_supported(h, w, cin) != 0:
                return None
            k = eng.params[eng.layout["b%d_dw" % i][0]:]; gk = eng.grads[eng.layout["b%d_dw" % i][0]:]
            st = not (eng.cfg.flags & 32) and lib.crnn_dwconv_bwd_stream_supported(B, h, w, cin) == 0    # the step's own choice of kernel
            nstream += int(st)
            pro = (eng.ws_tensor("q%d" % (i - 1)), eng.ws_tensor("bn2s%d" % (i - 1)), i - 1) if (st and lib.crnn_block_output_fused(eng._c, i - 1)) else None
            npro += int(pro is not None)
            launches.append((eng.ws_tensor("d%d" % i), eng.ws_tensor("gA"), eng.ws_tensor("bn1s%d" % i), coef, eng.ws_tensor("x%d" % (i - 1)), k,
                             eng.ws_tensor("gB"), gk, parts, h, w, cin, st, pro))
            nbytes += 4 * (2.0 * B * h * w * cin)
        h, w, cin = h // ph, w // pw, co
    times = []
    for it in range(iters + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for d, da, st, cf, x, k, dx, dk, pt, hh, ww, cc, strm, pro in launches:
            if pro is not None:     # the step's own form: x re-formed from the previous block's q in LDS
                lib.crnn_dwconv3x3_bwd_stream_pro(_ptr(d), _ptr(da), _ptr(st), _ptr(cf), _ptr(pro[0]), _ptr(pro[1]), rate,
                                                  _ptr(eng.ws_tensor("dm%d" % pro[2])) if rate > 0 else None, _ptr(k), _ptr(dx), _ptr(dk), _ptr(pt),
                                                  _ptr(eng.ws_tensor("bn2parts")) if (eng.cfg.flags & 2048) else None, B, hh, ww, cc, _stream())
                continue
            fn = lib.crnn_dwconv3x3_bwd_stream if strm else lib.crnn_dwconv3x3_bwd_fused
            fn(_ptr(d), _ptr(da), _ptr(st), _ptr(cf), _ptr(x), _ptr(k), _ptr(dx), _ptr(dk), _ptr(pt), B, hh, ww, cc, _stream())
        e1.record(); torch.cuda.synchronize()
        if it:
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(times)); ach = nbytes / t / 1e9
    kname = ("dw_bwd_stream_kernel (BatchNorm-backward pass 2 + depthwise weight and data gradients, blocks 2-7: rows of d, da, x streamed through an LDS ring, "
             "weight-gradient and data-gradient wave groups; incl. the second-stage sum of the weight-gradient partials; %d of the %d launches re-form x from "
             "the previous block's q in LDS)" % (npro, len(launches)) if nstream == len(launches) else
             "dw_bwd_fused_kernel (BatchNorm-backward pass 2 + depthwise weight and data gradients, blocks 2-7; VALU-issue-bound, DESIGN.md section 4)")
    tr = pmc_step_traffic(eng, ["dw_bwd_stream_kernel"]) if nstream == len(launches) else None
    return {"bound": "hbm", "kernel": kname,
            "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "launches": len(launches),
            "avg_launch_ms": round(1e3 * t / len(launches), 4), "algorithmic_bytes_per_launch_set": nbytes,
            "traffic": None if tr is None or tr[1] != len(launches) else tr[0],
            "traffic_note": None if tr is None else PMC_NOTE % (eng.precision, eng.precision)}


def batchnorm_roofline(eng, iters=5):
    """Secondary: the BatchNorm streaming passes of the conv stack that are kernels of their own in the bf16s step -- the block outputs' BatchNorm-2 +
    ReLU6 + pool + dropout (forward; the un-pooled blocks 1, 2, 4, 6 have none where the next depthwise kernel applies it in LDS) and BatchNorm-2's backward (statistics pass, finalize, apply pass: blocks 7..1) -- re-issued on the
    live buffers.  Algorithmic bytes in the storage type: apply reads q and writes x; the backward's two passes read g and q twice and write dq."""
    from crnn_mi355x.engine import _ptr, _stream
    if eng.precision != "bf16s":
        return None
    lib = eng.lib; B = eng.B
    blocks = [(64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)]
    h, w = eng.cfg.imgh + 4, eng.cfg.imgw + 4
    parts, coef = eng.ws_tensor("partials"), eng.ws_tensor("coef")
    rate = 0.1 if eng.cfg.dropout else 0.0
    fwd, bwd, fb, bb = [], [], 0.0, 0.0
    for i, (co, ph, pw) in enumerate(blocks, 1):
        q, st, x = eng.ws_tensor("q%d" % i), eng.ws_tensor("bn2s%d" % i), eng.ws_tensor("x%d" % i)
        M, Mo = B * h * w, B * (h // ph) * (w // pw)
        gam = eng.params[eng.layout["b%d_bn2_g" % i][0]:]; dg = eng.grads[eng.layout["b%d_bn2_g" % i][0]:]; db = eng.grads[eng.layout["b%d_bn2_b" % i][0]:]
        if not lib.crnn_block_output_fused(eng._c, i):      # (else the next block's depthwise kernels form x from q in LDS: no pass of its own)
            fwd.append((q, st, x, h, w, co, ph, pw, i)); fb += 2.0 * (M + Mo) * co
        bwd.append((q, st, gam, dg, db, h, w, co, ph, pw, i)); bb += 2.0 * (2 * (M + Mo) + M) * co
        h, w = h // ph, w // pw
    gA, gB = eng.ws_tensor("gA"), eng.ws_tensor("gB")
    res = {}
    for name, nb, nl in (("apply", fb, len(fwd)), ("backward", bb, 2 * len(bwd))):
        ts = []
        for it in range(iters + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "apply":
                for q, st, x, hh, ww, cc, ph, pw, layer in fwd:
                    lib.crnn_bn_act_pool_drop_ex(_ptr(q), _ptr(st), _ptr(x), B, hh, ww, cc, ph, pw, rate, 1234, layer, 1, 1, _stream())
            else:
                for q, st, gam, dg, db, hh, ww, cc, ph, pw, layer in reversed(bwd):
                    lib.crnn_bn_bwd_ex(_ptr(q), _ptr(gA), _ptr(st), _ptr(gam), _ptr(gB), _ptr(dg), _ptr(db), _ptr(parts), _ptr(coef), B, hh, ww, cc, ph, pw,
                                       rate, 1234, layer, 1, _stream())
            e1.record(); torch.cuda.synchronize()
            if it:
                ts.append(e0.elapsed_time(e1) * 1e-3)
        t = float(np.median(ts))
        tr = pmc_step_traffic(eng, ["bn_act_pool_drop_kernel"] if name == "apply" else ["bn_bwd_kernel", "bn_bwd_pool_kernel"])
        # (the step's BatchNorm-1 backward statistics of block 2 run bn_bwd_kernel<1> once more than this launch set: same kernel family)
        res[name] = {"achieved": round(nb / t / 1e9, 1), "frac": round(nb / t / 1e9 / PEAK_HBM_GBS, 4), "launches": nl, "ms": round(1e3 * t, 4),
                     "algorithmic_bytes_per_launch_set": nb, "traffic": None if tr is None else tr[0], "traffic_launches_in_step": None if tr is None else tr[1]}
    return {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "kernel": "bn_act_pool_drop_kernel (BatchNorm-2 + ReLU6 + MaxPooling + Dropout of the seven block outputs) and bn_bwd_kernel / bn_bwd_pool_kernel "
                      "(its backward: statistics pass, finalize, apply pass); the gradient buffers are overwritten: run after the step's timing",
            "traffic_note": PMC_NOTE % (eng.precision, eng.precision), **res}


def pointwise_gemm_roofline(eng, iters=5):
    """Secondary: the NN GEMM family (pointwise 1x1 convs b2..b7 fwd, dense1, RNN input projections) of one step on the
    live buffers: flops / time against the MFMA peak of the active mode."""
    from crnn_mi355x.engine import _ptr, _stream
    lib = eng.lib
    bf = eng.precision != "fp32"
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
    sdt = 1 if eng.precision == "bf16s" else 0
    wsrc = eng.ws_tensor("pbf") if bf else eng.params      # bf16 modes read the weights from their bf16 shadow
    W = lambda name: wsrc[eng.layout[name][0]:]
    B, T = eng.B, eng.T
    cfgs = []
    h, w, cin = eng.cfg.imgh + 4, eng.cfg.imgw + 4, 1
    blocks = [(64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)]
    pwT = eng.ws_tensor("pwT") if bf else None         # bf16 W^T copies the forward reads in the bf16 modes (NT GEMM)
    toff = 0
    for i, (co, ph, pw) in enumerate(blocks, 1):
        M = B * h * w
        if co > 64:
            if bf:
                # bf16s: the step's own forward entry (BatchNorm + ReLU6 of the depthwise output applied on the way in, statistics of q
                # taken on the way out; weights-resident kernel where its shape rules hold, else the tile GEMM) on d, not on `a`
                fused = eng.precision == "bf16s" and not (eng.cfg.flags & 8)
                cfgs.append((eng.ws_tensor(("d%d" if fused else "a%d") % i), pwT[toff:], eng.ws_tensor("q%d" % i), M, co, cin, sdt, sdt,
                             ("pw", eng.ws_tensor("bn1s%d" % i)) if fused else 1))
                toff += cin * co
            else:
                cfgs.append((eng.ws_tensor("a%d" % i), W("b%d_pw" % i), eng.ws_tensor("q%d" % i), M, co, cin, sdt, sdt, 0))
        h, w, cin = h // ph, w // pw, co
    feat = w * cin
    TB = T * B
    scratch = eng.ws_tensor("gemm_scratch"); parts = eng.ws_tensor("partials")
    u, G = eng.cfg.units, (3 if eng.cfg.gru else 4) * eng.cfg.units
    # bf16s: dense1 on the step's own kernel (round 5: the 64-row stripe stream over the bf16 W1^T the forward keeps behind the pointwise copies and dense2's
    # padded W^T in "pwT"; bias + ReLU + row permutation + Dropout(.4) in the epilogue), else the tile GEMM
    d1s = eng.precision == "bf16s" and not (eng.cfg.flags & 2) and feat % 8 == 0 and lib.crnn_dense_fwd_stream_supported(TB, eng.cfg.tds, feat) == 0
    if d1s:
        cfgs.append((eng.ws_tensor("x7"), None, eng.ws_tensor("gA"), TB, eng.cfg.tds, feat, sdt, 0, ("d1", pwT[toff + 128 * 2 * u:])))
    else:
        cfgs.append((eng.ws_tensor("x7"), W("dense1_w"), eng.ws_tensor("gA"), TB, eng.cfg.tds, feat, sdt, 0, 0))
    for l, src, k in (("1", "dn1", eng.cfg.tds), ("2", "r1", u)):
        # bf16 modes: the step's own kernel -- both directions of a layer in one launch of persistent workgroups on the bf16 W^T copies the forward keeps
        # (model.hip xw2 / crnn_rnn_input_proj; round 5) --, else one tile GEMM per direction
        xs = bf and u % 128 == 0 and not (eng.cfg.flags & 2) and TB % 64 == 0 and G % 128 == 0 and k % 64 == 0 and lib.crnn_rnn_input_proj_supported(TB, G, k) == 0
        if xs:
            cfgs.append((eng.ws_tensor(src), None, eng.ws_tensor("gB"), TB, 2 * G, k, 0, 0, ("xw2", eng.ws_tensor("wt" + l + "f"), eng.ws_tensor("wt" + l + "b"))))
        else:
            for dr in ("f", "b"):
                cfgs.append((eng.ws_tensor(src), W("rnn" + l + dr + "_w"), eng.ws_tensor("gB"), TB, G, k, 0, 0, 0))
    flops = sum(2.0 * M * N * K for _, _, _, M, N, K, _, _, _ in cfgs)
    # algorithmic HBM bytes of the same launches: A read + C written once, in their storage types (weights negligible)
    hbm_bytes = sum(M * K * (2.0 if dta else 4.0) + M * N * (2.0 if dtc else 4.0) for _, _, _, M, N, K, dta, dtc, _ in cfgs)
    times = []
    for it in range(iters + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for A, Bm, C, M, N, K, dta, dtc, wt in cfgs:
            if isinstance(wt, tuple) and wt[0] == "xw2":      # (both outputs into the scratch tensor: G columns each, the second half behind the first's rows)
                lib.crnn_rnn_input_proj(_ptr(A), _ptr(wt[1]), _ptr(wt[2]), None, None, _ptr(C), ctypes.c_void_p(C.data_ptr() + M * (N // 2) * 4), M, N // 2, K, K, K, N // 2, _stream())
            elif isinstance(wt, tuple) and wt[0] == "d1":
                lib.crnn_dense_fwd_stream(_ptr(A), _ptr(wt[1]), None, _ptr(C), M, N, K, K, K, 1, T, 0.4, 0, 8, _stream())
            elif isinstance(wt, tuple):
                if not (eng.cfg.flags & 2) and lib.crnn_pwconv_fwd_wres_supported(M, N, K) == 0:
                    lib.crnn_pwconv_bnrelu6_fwd_wres(_ptr(A), _ptr(wt[1]), _ptr(Bm), _ptr(C), M, N, K, _ptr(parts), _stream())
                else:
                    lib.crnn_pwconv_bnrelu6_fwd(_ptr(A), _ptr(wt[1]), _ptr(Bm), _ptr(C), M, N, K, _ptr(parts), 1, 1, _stream())
            elif bf:
                lib.crnn_gemm_bf16_ex(1 if wt else 0, _ptr(A), _ptr(Bm), _ptr(C), M, N, K, K, K if wt else N, N, None, 0, 0, 0, _ptr(scratch),
                                      64 * 1024 * 1024, dta, 1, dtc, _stream())
            else:
                lib.crnn_gemm_f32(0, _ptr(A), _ptr(Bm), _ptr(C), M, N, K, K, N, N, None, 0, 0, 0, _ptr(scratch), 64 * 1024 * 1024, _stream())
        e1.record()
        torch.cuda.synchronize()
        if it:
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(times))
    ach = flops / t / 1e12
    kname = ("gemm_wres_fwd_kernel (pointwise 1x1 convs fwd incl. BN+ReLU6 prologue and statistics) + gemm_nt_f32_stream_kernel<1, true> (dense1 incl. its ReLU / dropout epilogue) + "
             "gemm_nt_f32_proj_kernel (RNN input projections, both directions of a layer per launch)"
             if any(isinstance(c[8], tuple) and c[8][0] == "pw" for c in cfgs) else
             "gemm_%s_kernel<128,false,true> (pointwise 1x1 convs fwd + dense1 + RNN input GEMMs)" % ("bf16" if bf else "f32"))
    # memory-side bytes of the same launches in the step (bf16s: the six weights-resident pointwise forwards, dense1's stripe stream, the two layers' input projections)
    tr = pmc_step_traffic(eng, ["gemm_wres_fwd_kernel", "gemm_nt_f32_stream_kernel<1, true", "gemm_nt_f32_proj_kernel"]) \
        if eng.precision == "bf16s" else None
    return {"bound": "mfma", "kernel": kname,
            "traffic": None if tr is None or tr[1] != len(cfgs) else tr[0], "traffic_note": None if tr is None else PMC_NOTE % (eng.precision, eng.precision),
            "mfma_busy_counter": (lambda m: None if m is None else {"gemm_wres_fwd_kernel": m[0], "gemm_wres_kernel (data gradient)": m[1], "pw_wgrad_stream_kernel": m[2],
                                                                    "gemm_x3p_kernel": m[3]})(pmc_mfma_util(eng, ["gemm_wres_fwd_kernel", "gemm_wres_kernel", "pw_wgrad_stream_kernel", "gemm_x3p_kernel"])),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "launches": len(cfgs), "avg_launch_ms": round(1e3 * t / len(cfgs), 4), "flops_per_launch_set": flops,
            # with bf16 tensors every one of these GEMMs sits below the 312 FLOP/B ridge: the binding roof is HBM
            "hbm_bytes_per_launch_set": hbm_bytes, "hbm_achieved_GBps": round(hbm_bytes / t / 1e9, 1),
            "hbm_frac": round(hbm_bytes / t / 1e9 / PEAK_HBM_GBS, 4)}


def synthetic_batch(B, seed, imgh=100, imgw=32, max_len=23, num_classes=38, T=52, variable_width=None):
    """SURVEY 8d synthetic inputs: uint8 noise images normalised like Readf (utils.py:415-416, train.py mean/std),
    label lengths ~ U{1..max_len}, labels ~ U{0..36} padded with the blank (37), input_length = T - 2.  The IAM shape
    (imgh 200, BASELINE configs[2]) draws variable-width text: a random prefix of 40..200 rows of the time axis is noise, the
    rest the text's modal grey value (how open_img pads a short word, utils.py:372-400)."""
    rs = np.random.RandomState(seed)
    if variable_width is None:
        variable_width = imgh >= 200
    if variable_width:
        raw = np.empty((B, imgh, imgw, 1), dtype=np.uint8)
        for i in range(B):
            w = int(rs.randint(min(40, imgh), imgh + 1))
            text = rs.randint(0, 256, (w, imgw)).astype(np.uint8)
            val, counts = np.unique(text, return_counts=True)
            raw[i, :w, :, 0] = text
            raw[i, w:, :, 0] = val[np.where(counts == counts.max())[0][0]]
    else:
        raw = rs.randint(0, 256, (B, imgh, imgw, 1))
    x = ((raw.astype(np.float32) - 118.24236953981779) / 36.72835353999682).astype(np.float32)
    blank = num_classes - 1
    ll = rs.randint(1, max_len + 1, size=B)
    labels = np.full((B, max_len), blank, dtype=np.int64)
    for i in range(B):
        labels[i, :ll[i]] = rs.randint(0, blank, size=ll[i])
    return x, labels, np.full(B, T - 2, dtype=np.int64), ll.astype(np.int64)


def parity_check(B, n=16, precisions=("fp32", "bf16s"), gru=False):
    """BASELINE.json's metric is "images/sec + CTC-loss parity": the benchmarked configuration against the oracle (the checker, outside
    every timed region).  Inference BatchNorm makes the images independent, so the first `n` images of the benchmarked synthetic batch
    (seed 0, batch B) are run through the fp64 oracle alone and compared with rows 0..n-1 of the device's batch-B forward, per arithmetic
    mode: max |d logit|, max |d posterior|, max |d CTC cost| per sample (HIP CTC kernel on the device's posteriors vs oracle CTC on the
    oracle's; utils.py:98-103) absolute and relative, arg-max and greedy-decode agreement (utils.py:347-357 with beam_width 1).
    Weights: the Keras-family initialisation with randomised biases / BatchNorm parameters / moving statistics (the untouched
    initialisation decodes every image to "" -- agreement would be trivial)."""
    from crnn_mi355x.engine import Engine, _ptr, _stream
    from crnn_mi355x.init import initial_parameters
    from oracle import model as OM, ctc as OC
    cfg = OM.Config(gru=gru)
    probe = None
    out = {"images": n, "batch": B, "reference": "oracle/model.py forward (fp64 NumPy restatement of utils.py:58-96,247-258) + oracle/ctc.py on the first %d images "
                                                  "of the benchmarked batch, inference mode" % n,
           "tolerance": "north_star: fp32 logits within 1e-3, CTC loss within 1e-3, arg-max / greedy indices bit-exact"}
    x, lab, il, ll = synthetic_batch(B, seed=0, T=cfg.T)
    p64 = y_ref = None
    for precision in precisions:
        flags = None
        if precision == "fp32_two_plane_forward":     # the parity mode with CRNN_FLAG_TWO_PLANE_FORWARD (opt-in)
            precision, flags = "fp32", 131072 | int(os.environ.get("CRNN_FLAGS", "0"))
        eng = Engine(B, dropout=False, precision=precision, gru=gru, flags=flags)
        if flags is not None:
            precision = "fp32_two_plane_forward"
        if p64 is None:
            p = initial_parameters(eng.layout, eng.cfg.units, gru, seed=1)
            rs = np.random.RandomState(2)
            for k in p:
                if k.endswith(("_b", "_g")) or k == "stn_d2_w":
                    p[k] = (p[k] + rs.normal(size=p[k].shape) * (0.02 if k.startswith("stn_d2") else 0.3)).astype(np.float32)
            bn = {}
            for name, (off, ch, _) in eng.bn_layout.items():
                bn[name + "_mean"] = (rs.normal(size=ch) * 0.1).astype(np.float32)
                bn[name + "_var"] = (np.abs(rs.normal(size=ch)) * 0.5 + 0.5).astype(np.float32)
            p64 = {k: v.astype(np.float64) for k, v in p.items()}
            bn64 = {k: v.astype(np.float64) for k, v in bn.items()}
            t0 = time.perf_counter()
            y_ref, c_ref = OM.forward(cfg, p64, bn64, x[:n].astype(np.float64), train=False)
            loss_ref, _ = OC.ctc_loss_and_grad(y_ref, lab[:n], il[:n], ll[:n])
            g_ref, gl_ref = OC.ctc_greedy_decode(y_ref)
            out["oracle_seconds"] = round(time.perf_counter() - t0, 2)
        eng.set_params(p, bn)
        y = eng.forward(x, train=False)
        eng._ctc_inputs(lab, il, ll)
        scratch = eng.ws_tensor("dlogits")
        rc = eng.lib.crnn_ctc_loss_grad(_ptr(y), _ptr(eng._lab), _ptr(eng._il), _ptr(eng._ll), _ptr(eng.loss), _ptr(scratch), eng.B, eng.T, eng.C,
                                        eng.cfg.max_len, 2, 0.0, _stream())
        assert rc == 0, rc
        go, gl = eng.greedy_decode(y)
        eng.check_rnn_status()
        yh = y.float().cpu().numpy()[:n].astype(np.float64)
        logits = eng.ws_tensor("logits").float().cpu().numpy().reshape(B, eng.T, eng.C)[:n].astype(np.float64)
        loss = eng.loss.cpu().numpy()[:n].astype(np.float64)
        go, gl = go.cpu().numpy()[:n], gl.cpu().numpy()[:n]
        same_seq = [bool(gl[i] == gl_ref[i] and np.array_equal(go[i, :gl[i]], g_ref[i, :gl_ref[i]])) for i in range(n)]
        dl = np.abs(loss - loss_ref)
        out[precision] = {"max_abs_dlogit": float(np.abs(logits - c_ref["logits"]).max()), "max_abs_dposterior": float(np.abs(yh - y_ref).max()),
                          "max_abs_dctc_cost": float(dl.max()), "max_rel_dctc_cost": float((dl / np.abs(loss_ref)).max()),
                          "mean_ctc_cost_device": float(loss.mean()), "mean_ctc_cost_oracle": float(loss_ref.mean()),
                          "argmax_agreement": float((np.argmax(yh, -1) == np.argmax(y_ref, -1)).mean()),
                          "greedy_decodes_equal": float(np.mean(same_seq)),
                          "nonblank_argmax_fraction_oracle": float((np.argmax(y_ref, -1) != cfg.num_classes - 1).mean())}
        o = out[precision]
        o["within_tolerance"] = bool(o["max_abs_dlogit"] <= 1e-3 and o["max_abs_dctc_cost"] <= 1e-3 + 1e-5 * float(np.abs(loss_ref).max())
                                     and o["argmax_agreement"] == 1.0 and o["greedy_decodes_equal"] == 1.0)
        for k, v in list(o.items()):
            if isinstance(v, float):
                o[k] = float("%.4g" % v)
        del eng
        torch.cuda.empty_cache()
    return out


def fit_leg(B, steps, precision, warm=5):
    """Host-inclusive rate through the reference's own surface (train.py:201-209): CRNN(...).get_model() -> compile(Adam) ->
    Model.fit_generator over a Readf-style generator of HOST NumPy batches (float64 images as get_blank_matrices makes them, int64
    labels / lengths, the generator keeps rewriting the arrays it yielded).  Includes the float64 -> float32 conversion, the PCIe
    transfer (page-locked double-buffered staging, batch k+1 under step k), the per-step loss read-back and the callback plumbing."""
    import utils as U
    old = os.environ.get("CRNN_PRECISION")
    os.environ["CRNN_PRECISION"] = precision
    try:
        x, lab, il, ll = synthetic_batch(B, seed=0)
        X = x.astype(np.float64)
        inputs = {"the_input": X, "the_labels": lab.astype(np.int64), "input_length": il.reshape(-1, 1).astype(np.int64),
                  "label_length": ll.reshape(-1, 1).astype(np.int64), "source_str": np.array(["x"] * B)}
        outputs = {"ctc": np.zeros([B])}

        def gen():
            while True:
                np.add(X, 0.0, out=X)                   # the generator owns and rewrites these arrays between yields
                yield inputs, outputs
        model = U.CRNN(num_classes=38, shape=(100, 32, 1), GRU=False, time_dense_size=128, n_units=256, max_string_len=23).get_model()
        model.compile(loss={"ctc": lambda y_true, y_pred: y_pred}, optimizer=U.optimizers.Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5))
        g = gen()
        model.fit_generator(g, steps_per_epoch=warm, epochs=1, verbose=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H = model.fit_generator(g, steps_per_epoch=steps, epochs=1, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": round(B * steps / dt, 1), "unit": "images/sec", "ms_per_step": round(1e3 * dt / steps, 3), "batch": B, "steps": steps,
                "dtype": "f32" if precision == "fp32" else "bf16", "loss": round(float(H.history["loss"][-1]), 4),
                "workload": "Model.fit_generator over host NumPy batches (float64 images, batch %d) through utils.CRNN / compile / fit_generator" % B}
    finally:
        if old is None:
            os.environ.pop("CRNN_PRECISION", None)
        else:
            os.environ["CRNN_PRECISION"] = old
        torch.cuda.empty_cache()


def cpu_baseline(full=False):
    """The reference's CPU path is Keras-TF (train.py with --G 0), absent from this image; what IS timed here, on this box's host
    cores, is the torch-CPU fp32 restatement of the same graph (oracle/torch_port.py: training-mode forward, CTC cost, autograd
    backward, global-norm clip, Keras-form Adam) on the metric's literal batch: 64 synthetic 100x32 images (BASELINE configs[0]).
    Bounded sample (about 90 s of CPU work; round 6: 2 + 10 steps instead of 1 + 5): two untimed warm-up steps + ten timed steps with 16 threads;
    then two timed steps with 4 threads (the reference's own CPU setting, predict.py:88-93)."""
    from oracle import torch_port as TP
    # this graph does not scale over cores on the CPU (52-step Python LSTM loops, small ops): measured on the MI355X box's host
    # (2 x EPYC 9575F, 256 logical cores) 4 / 16 / 32 / 64 threads = 6.2 / 6.5 / 6.3 / 9.1 s per step, and minutes per step with all
    # 256 -- so "all cores" is capped at 16 threads and `cores` states the threads actually used
    cores = min(os.cpu_count() or 1, 16)
    nwarm, nsteps = (5, 20) if full else (2, 10)
    sall, _ = TP.train_step_benchmark(batch=64, threads=cores, steps=nsteps, warmup=nwarm)
    s4, _ = TP.train_step_benchmark(batch=64, threads=4, steps=nsteps if full else 2, warmup=0)     # the allocator / thread pools are warm by now
    return {"value": round(64 / sall, 2), "unit": "images/sec", "cores": int(cores), "kind": "port",
            "threads4": {"value": round(64 / s4, 2), "unit": "images/sec", "cores": 4, "sec_per_step": round(s4, 3)},
            "sec_per_step": round(sall, 3), "host_logical_cores": os.cpu_count(),
            "sample": "torch-CPU fp32 restatement of the Keras/TF graph (oracle/torch_port.py), full train step (fwd + CTC + bwd + clip + "
                      "Adam) at batch 64, 100x32: %d warm-up + %d timed steps (mean) with %d threads (more threads are slower on this graph); then %d timed "
                      "step(s) with 4 threads (the reference's CPU setting, predict.py:88-93); %s; Keras-TF itself is not in the image"
                      % (nwarm, nsteps, cores, nsteps if full else 2,
                         "BASELINE.md's full protocol (--cpu-baseline-full)" if full else
                         "a bounded sample (about 90 s of CPU work; --cpu-baseline-full runs BASELINE.md's 5 + 20 steps: 2.5 minutes)")}


def predict_leg(batch=1024, iters=20, precision="bf16s", cpu_sample=32):
    """BASELINE configs[4] (predict.py:166-171 + utils.py:347-357): inference-only path at batch 1024 -- forward (BatchNorm moving
    statistics, dropout off) + CTC beam search (beam_width 10, top_paths 1, merge_repeated) as the HIP wavefront kernel; p50 latency
    per image = p50 of (forward + beam search of the whole batch) / batch.  Beside it the greedy decode and the CPU restatement of the
    TF beam search (oracle/ctc.py, one host thread) on a bounded sample of the same posteriors."""
    from crnn_mi355x.engine import Engine
    from crnn_mi355x.init import initial_parameters
    B = batch
    eng = Engine(B, dropout=False, precision=precision)
    p = initial_parameters(eng.layout, eng.cfg.units, False, seed=1)
    rs = np.random.RandomState(2)
    for k in p:                                        # non-degenerate posteriors: the identity-STN / zero-bias init decodes to ""
        if k.endswith(("_b", "_g")) or k == "stn_d2_w":
            p[k] = (p[k] + rs.normal(size=p[k].shape) * (0.02 if k.startswith("stn_d2") else 0.3)).astype(np.float32)
    eng.set_params(p)
    x, lab, il, ll = synthetic_batch(B, seed=0, T=eng.T)
    xd = torch.from_numpy(x).cuda()

    def timed(fn, n):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), float(np.percentile(ts, 90))
    state = {}

    def fwd(): state["y"] = eng.forward(xd, train=False)
    def beam(): state["beam"] = eng.beam_decode(state["y"], beam_width=10)
    def greedy(): state["greedy"] = eng.greedy_decode(state["y"])
    def both(): fwd(); beam()
    for _ in range(3):
        both(); greedy()
    f50, _ = timed(fwd, iters)
    b50, _ = timed(beam, iters)
    g50, _ = timed(greedy, iters)
    t50, t90 = timed(both, iters)
    eng.check_rnn_status()
    out = {"workload": "BASELINE configs[4]: predict path, batch %d, 100x32x1, forward (inference BatchNorm) + CTC beam search bw=10" % B,
           "precision": precision, "iters": iters, "forward_ms_p50": round(f50, 3), "beam_decode_ms_p50": round(b50, 3),
           "greedy_decode_ms_p50": round(g50, 3), "forward_plus_beam_ms_p50": round(t50, 3), "forward_plus_beam_ms_p90": round(t90, 3),
           "latency_us_per_image_p50": round(1e3 * t50 / B, 3), "images_per_sec": round(B / (t50 * 1e-3), 1)}
    if cpu_sample:
        # CPU restatement of TF's beam search (the oracle: only this baseline leg uses it) on a bounded sample, one thread
        from oracle import ctc as OC
        y = state["y"].float().cpu().numpy()
        n = min(cpu_sample, B)
        t0 = time.perf_counter()
        ref = OC.ctc_beam_decode(y[:n].astype(np.float64), beam_width=10)
        cpu_ms = 1e3 * (time.perf_counter() - t0) / n
        o, ln, _ = [t.cpu().numpy() for t in state["beam"]]
        agree = sum(int(ln[i] == ref[1][i] and list(o[i, :ln[i]]) == list(ref[0][i, :ref[1][i]])) for i in range(n)) / n
        out.update({"cpu_beam_ms_per_image": round(cpu_ms, 3), "cpu_beam_sample": n, "cpu_beam_kind": "port (oracle/ctc.py, 1 thread)",
                    "beam_agreement_with_cpu_on_sample": agree,
                    "reference_published": "1.01 s/image for the per-image K.ctc_decode loop (reference README.md:75, unknown hardware)"})
    del eng
    torch.cuda.empty_cache()
    return out


def lstm_roofline(eng, iters=10):
    """The LSTM gate GEMM (north_star: >= 40 % MFMA utilisation target): the four recurrences of one step (2 Bidirectional layers x
    forward + BPTT) re-issued on the live buffers.  FLOPs = the recurrent products only (h_{t-1} U and dz_t U^T: 2 x 2 dirs x T x
    2 B u 4u per layer and pass); the hoisted input projections are part of gemm_roofline.  The recurrence is a chain of T dependent
    steps of 268 MFLOP each, i.e. latency-bound by construction: `us_per_step` is the number to read."""
    from crnn_mi355x.engine import _ptr, _stream
    lib, cfg = eng.lib, eng.cfg
    if cfg.gru:
        return None
    B, T, u = eng.B, eng.T, cfg.units
    bf = eng.precision != "fp32" and u % 128 == 0
    dt = 1 if bf else 0
    persist = not (cfg.flags & 1) and lib.crnn_lstm_persist_supported(u, dt) == 0
    W = eng.ws_tensor
    if bf:
        pbf = W("pbf")
        U = lambda n: ctypes.c_void_p(pbf.data_ptr() + 2 * eng.layout[n][0])
    else:
        U = lambda n: ctypes.c_void_p(eng.params.data_ptr() + 4 * eng.layout[n][0])
    uwf = 0 if (cfg.flags & 64) else 0x100        # the step's own workgroup -> cluster map (XCD-local unless CRNN_FLAG_RNN_LINEAR_CLUSTERS)
    xb = W("rnnx") if persist else None
    nx = lib.crnn_lstm_persist_xbuf_bytes(T, B, u, dt) if persist else 0
    off = lambda t, e: ctypes.c_void_p(t.data_ptr() + 4 * e)
    dcf, dcb = W("dcf"), W("dcb")

    def run():
        for l, (h0, h1, ldh, do0, do1) in ((1, (W("h1f"), W("h1b"), u, W("dr1"), W("dr1"))), (2, (W("h2"), off(W("h2"), u), 2 * u, W("dr2"), off(W("dr2"), u)))):
            p = lambda t: t if isinstance(t, ctypes.c_void_p) else _ptr(t)
            a = [_ptr(W("xw%df" % l)), _ptr(W("xw%db" % l)), _ptr(W("ut%df" % l)), _ptr(W("ut%db" % l)), p(h0), p(h1), ldh, _ptr(W("cs%df" % l)),
                 _ptr(W("cs%db" % l)), _ptr(W("gt%df" % l)), _ptr(W("gt%db" % l)), T, B, u, dt]
            g = [U("rnn%df_u" % l), U("rnn%db_u" % l), _ptr(W("cs%df" % l)), _ptr(W("cs%db" % l)), _ptr(W("gt%df" % l)), _ptr(W("gt%db" % l)), p(do0), p(do1),
                 ldh, _ptr(W("dz%df" % l)), _ptr(W("dz%db" % l))]
            if persist:
                rc = lib.crnn_lstm_fwd_persist(*a, _ptr(xb), nx, 0, uwf, _stream())
                rc |= lib.crnn_lstm_bwd_persist(*g, T, B, u, dt, _ptr(xb), nx, 0, uwf, _stream())
            else:
                rc = lib.crnn_lstm_fwd_ex(*a, _stream())
                rc |= lib.crnn_lstm_bwd_ex(*g, _ptr(dcf), _ptr(dcb), T, B, u, dt, _stream())
            assert rc == 0, rc
    times = []
    for it in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1) * 1e-3)
    t = float(np.median(times))
    flops = 2 * 2 * (2.0 * T * 2 * B * u * 4 * u)          # 2 layers x (forward + backward)
    peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS
    ach = flops / t / 1e12
    # the other half of the gate GEMM: the input projections x W + b hoisted over all T steps (four launches per forward), on the live buffers
    G = 4 * u
    proj = []
    for n, src, k in (("rnn1f", "dn1", cfg.tds), ("rnn1b", "dn1", cfg.tds), ("rnn2f", "r1", u), ("rnn2b", "r1", u)):
        proj.append((W(src), W("wt" + n[3:5]) if bf else None, W("xw" + n[3:5]), eng.params[eng.layout[n + "_w"][0]:], eng.params[eng.layout[n + "_b"][0]:], k))
    TB = T * B
    pt = []
    scratch = W("gemm_scratch")
    for it in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for xin, wt, out, wfp, bias, k in proj:
            if bf and u % 128 == 0 and not (cfg.flags & 2) and TB % 64 == 0 and k % 64 == 0:
                rc = lib.crnn_gemm_nt_f32_stream_bias(_ptr(xin), _ptr(wt), None, None, _ptr(out), _ptr(bias), TB, G, k, k, k, G, _stream())
            elif bf:
                rc = lib.crnn_gemm_bf16_ex(0, _ptr(xin), _ptr(wfp), _ptr(out), TB, G, k, k, G, G, _ptr(bias), 0, 0, 0, _ptr(scratch), 64 * 1024 * 1024, 0, 0, 0, _stream())
            else:
                fn = lib.crnn_gemm_f32x3 if (cfg.flags & 256) else lib.crnn_gemm_f32
                rc = fn(0, _ptr(xin), _ptr(wfp), _ptr(out), TB, G, k, k, G, G, _ptr(bias), 0, 0, 0, _ptr(scratch), 64 * 1024 * 1024, _stream())
            assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
        if it >= 2:
            pt.append(e0.elapsed_time(e1) * 1e-3)
    tp = float(np.median(pt))
    pflops = sum(2.0 * TB * k * G for *_, k in proj)
    return {"bound": "mfma", "kernel": ("lstm_fwd/bwd_persist_kernel (one launch per layer and pass: cluster of workgroups per batch tile, recurrent "
                                        "weights + cell state in registers, h/dz all-gather through a sentinel ring, LDS-staged MFMA operand)" if persist
                                        else "lstm_fwd/bwd_step_kernel (one launch per timestep)"),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "launches": 4 if persist else 4 * T,
            "ms_per_train_step": round(1e3 * t, 4), "us_per_step": round(1e6 * t / (4 * T), 3), "flops_recurrent_gemm": flops,
            "traffic": (lambda tr: None if tr is None else tr[0])(pmc_step_traffic(eng, ["lstm_fwd_persist_kernel", "lstm_bwd_persist_kernel"]) if persist else None),
            "traffic_note": PMC_NOTE % (eng.precision, eng.precision),
            "mfma_busy_counter": (lambda m: None if m is None else {"lstm_fwd_persist_kernel": m[0], "lstm_bwd_persist_kernel": m[1],
                                  "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x 2.4 GHz) of the step's launches (profiles/r06_pmc_sq_step_%s.txt)" % eng.precision})(
                                  pmc_mfma_util(eng, ["lstm_fwd_persist_kernel", "lstm_bwd_persist_kernel"])),
            "input_projections": {"achieved": round(pflops / tp / 1e12, 2), "unit": "TFLOP/s", "frac": round(pflops / tp / 1e12 / peak, 4), "launches": 4,
                                  "ms": round(1e3 * tp, 4), "flops": pflops,
                                  "note": "the hoisted half of the gate GEMM (x W + b over all T steps, forward): fp32 rows streamed once against a "
                                          "bf16 W^T -- bound by reading the 14-27 MB of x and writing 55 MB of xw per launch, not by the matrix cores"},
            "target_note": "north_star asks >= 40 % MFMA utilisation on the LSTM gate GEMM: the recurrent half is a chain of T dependent 268-MFLOP "
                           "steps (0.1 us of MFMA work each against a 1.7 us cross-workgroup hand-off), the hoisted half is HBM-bound; neither can "
                           "reach it at B = 256, T = 52 on any schedule (DESIGN.md section 4)",
            "note": "T = %d dependent steps per launch; each step's GEMM (2 x %d x %d x %d per direction) is ~0.1 us of MFMA work, the step time is "
                    "the cross-workgroup hand-off latency of h_t / dz_t" % (T, B, u, 4 * u)}


def timed_steps(eng, batch, opt, steps, warmup, it0=0):
    xd, labd, ild, lld = batch
    it = it0
    for _ in range(warmup):
        eng.train_step(xd, labd, ild, lld, opt, it); it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = eng.train_step(xd, labd, ild, lld, opt, it); it += 1
    torch.cuda.synchronize()
    return time.perf_counter() - t0, float(loss.mean().item())


def collective_transport_info(dist):
    """What the gradient exchange ran over, for the record (rank 0): backend, RCCL version, the NCCL_* / RCCL_* / HSA_* environment, visible GPUs
    and the number of xGMI links rocm-smi reports between them (0 on a single-GPU box or when the tool is missing).  Informational: never raises."""
    info = {"backend": dist.get_backend(), "visible_gpus": torch.cuda.device_count()}
    try:
        if dist.get_backend() == "nccl":
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:
        info["rccl_version"] = "unavailable (%s)" % type(e).__name__
    info["env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "CRNN_DIST"))}
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=20).stdout
        info["xgmi_links_reported"] = out.count("XGMI") // 2 if "XGMI" in out else 0      # the type matrix lists each pair twice
    except Exception as e:
        info["xgmi_links_reported"] = "unavailable (%s)" % type(e).__name__
    return info


def self_launch(n):
    """Spawn n ranks of this script (same argv) under torch.distributed.run on this node and relay rank 0's JSON line."""
    import socket
    import subprocess
    if os.environ.get("CRNN_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < n:
        print("bench.py: --gpus %d needs %d visible GPUs for RCCL, found %d" % (n, n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE configs[1]: 256)")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16s"], default="bf16s",
                    help="fp32 = parity mode (fp32 tensors, fp32-accurate three-plane bf16 products); bf16 = GEMM products in bf16, fp32 accumulate/storage")
    ap.add_argument("--imgh", type=int, default=100, help="image length (the time axis); 200 = the IAM shape of BASELINE configs[2]")
    ap.add_argument("--max-len", type=int, default=23, help="label capacity; 21 for the IAM shape")
    ap.add_argument("--gru", action="store_true", help="GRU recurrence (what the reference's train.py really builds, SURVEY F3) instead of "
                    "the LSTM BASELINE.json names")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="BASELINE.md's protocol for the CPU leg: 5 warm-up + 20 timed steps (about 2.5 minutes)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity object (oracle forward of 16 images: a few seconds of CPU)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp32 parity-mode and batch-64 timings")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size wins" % (args.gpus, world), file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CRNN_DIST_BACKEND", "nccl")     # "nccl" = RCCL; "gloo" only to exercise this path on one GPU
        if backend == "nccl":
            # (round 6) RCCL wants one GPU per rank: say so in one line instead of failing somewhere inside init_process_group / the first collective
            ndev, lws = torch.cuda.device_count(), int(os.environ.get("LOCAL_WORLD_SIZE", world))
            if ndev < lws:
                if rank == 0:
                    print(json.dumps({"error": "bench.py --gpus %d over RCCL needs %d GPUs on this node, torch.cuda.device_count() = %d "
                                               "(HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r); CRNN_DIST_BACKEND=gloo shares one GPU between the ranks"
                                               % (args.gpus, lws, ndev, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))}), flush=True)
                sys.exit(3)
        local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)

    from crnn_mi355x.engine import Engine
    from crnn_mi355x.init import initial_parameters
    from crnn_mi355x.optimizers import Adam
    from crnn_mi355x.parallel import GradAllReduce

    B = args.batch
    eng = Engine(B, imgh=args.imgh, max_len=args.max_len, dropout=True, precision=args.precision, gru=args.gru)
    eng.set_params(initial_parameters(eng.layout, eng.cfg.units, args.gru, seed=1))   # Keras-family init, identical on every rank
    x, lab, il, ll = synthetic_batch(B, seed=rank, imgh=args.imgh, max_len=args.max_len, T=eng.T)   # rank r draws its own shard
    xd = torch.from_numpy(x).cuda()
    labd = torch.from_numpy(lab.astype(np.int32)).cuda(); ild = torch.from_numpy(il.astype(np.int32)).cuda()
    lld = torch.from_numpy(ll.astype(np.int32)).cuda()
    opt = Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5)
    allreduce = GradAllReduce(eng, dist, world) if world > 1 else None
    preflight = None
    if world > 1:
        # (round 6) Before anything is timed: the gradient exchange in its blocking form and in its overlapped two-bucket form must leave the same averaged gradient on
        # every rank -- a transport or stream-ordering problem of the asynchronous schedule shows here, by name, instead of as a silently different loss curve.
        def grads_after(ar):
            eng.forward(xd, train=True, seed=0)
            if getattr(ar, "overlap", False):
                eng.backward_top(labd, ild, lld, seed=0)
                split = eng.grad_split
                ar.start(eng.grads[split:]); eng.backward_bottom(seed=0); ar.start(eng.grads[:split]); ar.finish(eng.grads)
            else:
                eng.backward(labd, ild, lld, seed=0); ar(eng.grads)
            torch.cuda.synchronize()
            return eng.grads.clone()
        g_block = grads_after(GradAllReduce(eng, dist, world, overlap=False))
        g_over = grads_after(allreduce)
        gmax = float(g_block.abs().max().item()) + 1e-30
        d_sched = float((g_block - g_over).abs().max().item()) / gmax            # blocking vs overlapped exchange, this rank
        chk = torch.stack([g_block.double().sum(), g_block.double().abs().sum(), g_over.double().sum(), g_over.double().abs().sum(),
                           torch.tensor(d_sched, dtype=torch.float64, device="cuda")])
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        allc = torch.stack(allc)
        d_ranks = float((allc[:, :4].max(0).values - allc[:, :4].min(0).values).abs().max().item())   # the averaged gradient, across ranks (either schedule)
        d_sched = float(allc[:, 4].max().item())
        preflight = {"what": "one forward + backward from the initial state with the blocking exchange (one all-reduce of the whole gradient buffer) and with the "
                             "overlapped two-bucket exchange: the averaged gradient buffers compared",
                     "max_abs_diff_blocking_vs_overlapped_rel_to_max_gradient": d_sched, "gradient_checksum_max_abs_diff_across_ranks": d_ranks,
                     # every rank must hold the same averaged gradient; the two schedules cut the buffer into different all-reduce calls, and a ring all-reduce adds
                     # the ranks' terms of an element in an order that depends on its position in the call: fp32 round-off of a sum of `world` terms (bit-equal at
                     # world size 2, where a + b has one order)
                     "overlapped_schedule_equals_blocking": d_ranks == 0.0 and d_sched <= (0.0 if world == 2 else 1e-5)}
        if rank == 0 and not preflight["overlapped_schedule_equals_blocking"]:
            print("bench.py: PREFLIGHT FAILED: %s" % json.dumps(preflight), file=sys.stderr, flush=True)
        del g_block, g_over
        # (the timed run starts from the initial state again)
        eng.set_params(initial_parameters(eng.layout, eng.cfg.units, args.gru, seed=1))
        for k in ("m", "v"):
            if k in eng.opt_state:
                eng.opt_state[k].zero_()
        opt = Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5)

    it = 0
    for _ in range(args.warmup):
        eng.train_step(xd, labd, ild, lld, opt, it, allreduce=allreduce); it += 1
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = eng.train_step(xd, labd, ild, lld, opt, it, allreduce=allreduce); it += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)                  # every rank's own clock around the same barrier-bracketed region: a straggler is visible by rank
        per_rank_ms = [round(1e3 * float(t.item()) / args.steps, 3) for t in allt]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    last_loss, giveups = eng.loss_and_status().tolist()
    eng.raise_if_rnn_gave_up(giveups)       # a persistent recurrence that lost its co-residency would have produced garbage silently

    dp_proof = None
    if dist is not None:
        # Self-proof of the N-rank run (after the timed region): the process group's own world size and backend, identical replicas
        # (max |difference| of a parameter checksum vector gathered from every rank: 0.0 when the all-reduce + same Adam step kept the
        # weights bit-identical), and how much of the gradient exchange is exposed (the same steps without the all-reduce, timed the
        # same way: replicas diverge there, which is why it runs last).
        chk = torch.stack([eng.params.double().sum(), eng.params.double().abs().sum(), (eng.params.double() ** 2).sum(),
                           eng.bn_mean.double().sum()])
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        allchk = torch.stack(allchk)
        spread = float((allchk[:, :3].max(0).values - allchk[:, :3].min(0).values).abs().max().item())
        bn_spread = float((allchk[:, 3].max() - allchk[:, 3].min()).abs().item())
        k2 = max(3, min(args.steps, 10))
        dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            eng.train_step(xd, labd, ild, lld, opt, it, allreduce=None); it += 1
        torch.cuda.synchronize(); dist.barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms_no_ar = 1e3 * float(tt.item()) / k2
        dp_proof = {"dist_world_size": dist.get_world_size(), "dist_backend": dist.get_backend(), "ranks_reporting": int(allchk.shape[0]),
                    "visible_gpus": torch.cuda.device_count(), "preflight": preflight,
                    "ms_per_step_per_rank": per_rank_ms, "ms_per_step_max_over_ranks": max(per_rank_ms),
                    "transport": collective_transport_info(dist),
                    "param_checksum_max_abs_diff_across_ranks": spread, "replicas_identical": spread == 0.0,
                    "bn_moving_mean_checksum_spread": bn_spread,
                    "allreduce_bytes_per_step": int(eng.n_total * 4), "ms_per_step_without_allreduce": round(ms_no_ar, 3),
                    "exposed_allreduce_ms": round(1e3 * dt / args.steps - ms_no_ar, 3),
                    "note": "checksums = (sum, sum|.|, sum of squares of the flat parameter buffer) in fp64 per rank after the timed steps; the BatchNorm "
                            "moving means are per-replica batch statistics by design (averaged by sync_bn_stats before validation / checkpoints), their "
                            "checksum spread is reported, not required to vanish; exposed = ms_per_step - the same step without the gradient exchange"}

    if rank == 0:
        res = {
            "metric": "text-line images/sec (train step)", "value": round(world * B * args.steps / dt, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %dx32x1 text lines, batch %d/GPU, max_len %d, time_dense_size 128, n_units 256 %s, "
                                   "STN on, dropout on, CTC, Adam(1e-4,b1=.5,clipnorm 5), %s" % (
                                       1 if args.imgh == 100 else 2, args.imgh, B, args.max_len, "BiGRU" if args.gru else "BiLSTM",
                                       {"fp32": "fp32 tensors, fp32-accurate GEMMs (three bf16 planes per operand on the bf16 MFMA)", "bf16": "bf16 MFMA products / fp32 accumulate+storage",
                                        "bf16s": "bf16 MFMA products, bf16 conv-stack tensors in HBM, fp32 accumulate/statistics/RNN/optimizer (outside the 1e-3 "
                                                 "parity tolerance: see parity_mode for the fp32 step)"}[args.precision]),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "final_loss": round(last_loss, 4)},
        }
        if dp_proof is not None:
            res["data_parallel"] = dp_proof
        if not args.no_roofline:
            res["roofline"] = depthwise_roofline(eng)
            res["gemm_roofline"] = pointwise_gemm_roofline(eng)
            dbr = depthwise_bwd_roofline(eng)
            if dbr is not None:
                res["dw_bwd_roofline"] = dbr
            lr = lstm_roofline(eng)
            if lr is not None:
                res["lstm_roofline"] = lr
            br = batchnorm_roofline(eng)
            if br is not None:
                res["bn_roofline"] = br
            if world == 1 and args.imgh == 100:      # the same depthwise pair in the parity mode's fp32 form, cold (config.dw_*_hbm_frac_cold_fp32)
                dfr = depthwise_fp32_cold(B)
                if dfr is not None:
                    res["dw_fp32_roofline"] = dfr
        if world == 1 and not args.no_secondary:
            adam = lambda: Adam(lr=1e-4, beta_1=0.5, beta_2=0.999, clipnorm=5)

            def leg(batch, steps, warm, reps=1, **kw):
                """One more engine on the same kind of synthetic batch, timed like the headline (wall clock around `steps` steps between
                device synchronisations); reps > 1: the MEDIAN repetition is reported, all are listed."""
                kw = dict(dict(imgh=args.imgh, max_len=args.max_len, dropout=True, precision=args.precision, gru=args.gru), **kw)
                e = Engine(batch, **kw)
                e.set_params(initial_parameters(e.layout, e.cfg.units, kw["gru"], seed=1))
                bt = tuple(torch.from_numpy(a if i == 0 else a.astype(np.int32)).cuda() for i, a in enumerate(
                    synthetic_batch(batch, seed=0, imgh=kw["imgh"], max_len=kw["max_len"], T=e.T)))
                o = adam()
                ts, ls = [], None
                for r in range(reps):
                    t, ls = timed_steps(e, bt, o, steps, warm if r == 0 else 0, it0=r * (steps + warm))
                    ts.append(t)
                e.check_rnn_status()
                del e
                torch.cuda.empty_cache()
                tm = float(np.median(ts))
                out = {"batch": batch, "value": round(batch * steps / tm, 1), "unit": "images/sec", "ms_per_step": round(1e3 * tm / steps, 3),
                       "dtype": "f32" if kw["precision"] == "fp32" else "bf16", "steps": steps, "final_loss": round(ls, 4)}
                if reps > 1:
                    out["ms_per_step_repetitions"] = [round(1e3 * t / steps, 3) for t in ts]
                    out["statistic"] = "median of %d repetitions" % reps
                return out
            if args.precision != "fp32":
                # the parity mode (fp32 storage + fp32-accurate three-plane GEMMs: the mode in which logits / CTC loss meet the 1e-3 tolerance and arg-max is
                # bit-exact against the oracle, tests/test_gpu_model.py) timed in the same run on the same workload
                res["parity_mode"] = dict(leg(B, max(3, min(args.steps, 10)), 2, precision="fp32"),
                                          note="fp32 tensors + fp32-accurate forward GEMMs (three bf16 planes per operand, six bf16 MFMAs per k-step; CRNN_FLAGS=256 = fp32 MFMA, 20.6 ms; the conv "
                                               "stack's backward GEMMs carry two planes = 16 significant bits per factor, `three_plane_backward` = three there too): "
                                               "the mode the 1e-3 logit / CTC-loss parity and bit-exact arg-max are asserted in; "
                                               "the headline bf16 line is outside that tolerance (bf16 conv-stack tensors: softmax within 2e-3, loss 2e-3 "
                                               "relative of the fp64 oracle).  Round 4: the depthwise stage on the fp32 forms of the row-stream kernels "
                                               "(block outputs and BatchNorm-2 backward statistics formed inside them)")
                # the same step on the round-3 schedule (halo-tile depthwise kernels, three-kernel depthwise-stage backward, every BatchNorm-2 pass on its own):
                # CRNN_FLAG_DW_TILE_KERNEL; same forward to summation order
                res["parity_mode"]["tile_schedule"] = dict(leg(B, max(3, min(args.steps, 10)), 2, precision="fp32", flags=32), flags=32)
                # plane counts of the conv stack's pointwise GEMMs (include/crnn_mi355x.h): default = three planes forward (fp32-accurate: what `parity`
                # checks), two planes backward (16 significant bits per factor, gradients within 1e-5); strict = three everywhere; and two everywhere
                # (its forward is checked against the oracle as parity["fp32_two_plane_forward"])
                # round 6: the pointwise convolutions with a reduction <= 256 run on the weights-resident plane kernels (gemm_wres3.hip); CRNN_FLAG_GEMM_TILE_KERNELS = the
                # round-5 schedule (every product on gemm_x3p_kernel), timed on the same box in the same run
                res["parity_mode"]["tile_gemm_schedule"] = dict(leg(B, max(3, min(args.steps, 10)), 2, precision="fp32", flags=2), flags=2)
                res["parity_mode"]["three_plane_backward"] = dict(leg(B, max(3, min(args.steps, 10)), 2, precision="fp32", flags=65536), flags=65536)
                res["parity_mode"]["two_plane_forward"] = dict(leg(B, max(3, min(args.steps, 10)), 2, precision="fp32", flags=131072), flags=131072)
            if B != 64:
                # the metric's literal batch size (BASELINE.json: "100x32 bs64"), same precision as the headline
                res["bs64"] = leg(64, max(5, args.steps), 3, reps=3)
                if args.precision != "fp32":   # ... and at the reference's own precision (the parity mode)
                    res["bs64_fp32"] = leg(64, max(5, min(args.steps, 10)), 3, reps=3, precision="fp32")
            if args.imgh == 100 and not args.gru and args.precision == "bf16s" and not (eng.cfg.flags & 1024):
                # opt-in schedules measured beside the default in the same run: block outputs formed inside the next block's depthwise kernels
                # (CRNN_FLAG_BN2_DW_FUSION), and with the BatchNorm-2 backward statistics taken there too (| CRNN_FLAG_BN2_STATS_FUSION)
                res["bn2_dw_fusion"] = dict(leg(B, max(5, min(args.steps, 20)), 3, flags=1024), flags=1024,
                                            note="opt-in: 4 BatchNorm-apply launches and 3 tensor passes per un-pooled block less, paid in VALU work on the "
                                                 "depthwise kernels' transform waves; bit-identical results")
                res["bn2_dw_stats_fusion"] = dict(leg(B, max(5, min(args.steps, 20)), 3, flags=1024 | 2048), flags=3072)
            if args.imgh == 100 and not args.gru:
                # the same step driven through the reference's surface from host batches (train.py:201-209)
                res["fit"] = fit_leg(B, max(10, args.steps), args.precision)
                res["fit"]["vs_device_resident"] = round(res["fit"]["value"] / res["value"], 3)
            if args.imgh == 100 and not args.gru and args.precision == "bf16s":
                # BASELINE configs[2]: the IAM shape (200x32 variable-width text, max_len 21, T = 102), STN on
                res["iam"] = dict(leg(B, max(3, min(args.steps, 10)), 2, imgh=200, max_len=21),
                                  workload="BASELINE configs[2]: 200x32x1 variable-width text lines (random 40..200-row prefix, modal-grey padding), "
                                           "max_len 21, T = 102, STN on, batch %d" % B)
                # the cell the reference's train.py really builds (train.py:119 shadows :109; utils.py:80-82): BiGRU, reset_after=False
                res["gru"] = dict(leg(B, max(3, min(args.steps, 10)), 2, gru=True),
                                  workload="configs[1] with the GRU recurrence the reference trains (train.py:119, utils.py:80-82)")
                # BASELINE configs[4]: predict path (predict.py:166-171), batch 1024, beam width 10
                res["predict"] = predict_leg(1024, iters=20, precision=args.precision, cpu_sample=0 if args.no_cpu_baseline else 32)
        if world == 1 and not args.no_parity and args.imgh == 100:
            # CTC-loss / logit / arg-max parity of the benchmarked configuration against the oracle (the metric's second half)
            res["parity"] = parity_check(B, 16, ("fp32", "fp32_two_plane_forward", "bf16s") if args.precision == "bf16s" else
                                         ("fp32", args.precision) if args.precision != "fp32" else ("fp32", "fp32_two_plane_forward"), gru=args.gru)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_baseline_full)
        # The contract in the one object the driver's record keeps whole (`config`): whether the 1e-3 / bit-exact tolerance holds in the parity mode
        # and in the benchmarked mode, and the rates of the modes / batches the metric string names beside the headline.  None = leg not run.
        cf = res["config"]
        par = res.get("parity", {})
        pm = res.get("parity_mode") if args.precision != "fp32" else {"ms_per_step": res["ms_per_step"], "value": res["value"]}
        get = lambda d, *ks: (get(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
        cf["parity_within_tolerance_fp32"] = get(par, "fp32", "within_tolerance")
        cf["parity_within_tolerance_headline"] = get(par, args.precision, "within_tolerance")
        cf["parity_max_abs_dlogit_fp32"] = get(par, "fp32", "max_abs_dlogit")
        cf["parity_max_abs_dlogit_headline"] = get(par, args.precision, "max_abs_dlogit")
        cf["parity_mode_ms_per_step"] = get(pm, "ms_per_step")
        cf["parity_mode_images_per_sec"] = get(pm, "value")
        cf["parity_mode_strict_ms_per_step"] = get(pm, "three_plane_backward", "ms_per_step")
        cf["parity_mode_backward_planes"] = "2 (16 significant bits per factor; CRNN_FLAG_THREE_PLANE_BACKWARD = strict)"
        b64 = res.get("bs64") if B != 64 else {"ms_per_step": res["ms_per_step"], "value": res["value"]}
        cf["bs64_ms_per_step"] = get(b64, "ms_per_step")
        cf["bs64_images_per_sec"] = get(b64, "value")
        cf["bs64_fp32_ms_per_step"] = get(res, "bs64_fp32", "ms_per_step") if args.precision != "fp32" else get(b64, "ms_per_step")
        cf["bs64_fp32_images_per_sec"] = get(res, "bs64_fp32", "value") if args.precision != "fp32" else get(b64, "value")
        cf["dw_fwd_hbm_frac_cold"] = get(res, "roofline", "frac")
        cf["dw_bwd_hbm_frac_cold"] = get(res, "dw_bwd_roofline", "frac")
        cf["dw_fwd_hbm_frac_cold_fp32"] = get(res, "dw_fp32_roofline", "fwd", "frac")      # the precision in which north_star's >= 0.60 IS met (round 6)
        cf["dw_bwd_hbm_frac_cold_fp32"] = get(res, "dw_fp32_roofline", "bwd", "frac")
        cf["pointwise_gemm_hbm_frac"] = get(res, "gemm_roofline", "hbm_frac")
        cf["lstm_gate_gemm_mfma_frac"] = get(res, "lstm_roofline", "frac")
        cf["predict_b1024_beam10_us_per_image_p50"] = get(res, "predict", "latency_us_per_image_p50")
        cf["fit_generator_images_per_sec"] = get(res, "fit", "value")
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
