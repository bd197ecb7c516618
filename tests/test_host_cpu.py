"""CPU tests (no GPU): the host-side mirror of the reference's Python surface against the golden vectors
generated from the reference itself (tests/golden/helpers_golden.json), the C-ABI library's symbol table and
layout functions, the data generator's batch contract, and the world_size-2 data-parallel reduction (gloo)."""
import ctypes
import json
import os
import sys

import numpy as np
import pytest

import utils as U  # crnn-ocr-lite_amd/utils.py (drop-in for the reference's module)
from crnn_mi355x import native
from crnn_mi355x.surface import param_layout, _cfg_struct

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "helpers_golden.json")))


def _reader(batch_size=4):
    classes = {ch: i for i, ch in enumerate(U.get_lexicon())}
    return U.Readf(img_size=(100, 32, 1), max_len=23, normed=True, batch_size=batch_size, classes=classes), classes


def test_lexicon_targets_blank_matrices_norm_match_reference():
    assert U.get_lexicon() == GOLD["lexicon"]
    reader, classes = _reader()
    for word, ids in GOLD["make_target"].items():
        w = word if word == "Zz9_-" else word.lower()
        assert [int(v) for v in reader.make_target(w)] == ids
    X, Y, il, ll = reader.get_blank_matrices()
    g = GOLD["blank_matrices"]
    assert list(X.shape) == g["X_shape"] and str(X.dtype) == g["X_dtype"]
    assert Y.tolist() == g["Y"] and str(Y.dtype) == g["Y_dtype"]
    assert il.tolist() == g["input_length"] and ll.tolist() == g["label_length"]
    out = U.norm(np.array(GOLD["norm_in"]), reader.mean, reader.std)
    assert str(out.dtype) == GOLD["norm_dtype"]
    np.testing.assert_array_equal(out, np.array(GOLD["norm_out"], dtype=np.float32))
    assert U.parse_mjsynth("/data/mj", ["./2425/1/115_Lube_45484.jpg 45484\n", "./1/2/3_a_1.jpg 1"]) == GOLD["parse_mjsynth"]


def test_edit_distances_and_label_text_match_reference():
    for a, b, d in GOLD["levenshtein"]:
        assert U.levenshtein(a, b) == d and isinstance(U.levenshtein(a, b), float)
    pairs = [(a, b) for a, b, _ in GOLD["levenshtein"] if b]
    assert U.edit_distance([a for a, _ in pairs], [b for _, b in pairs]) == pytest.approx(GOLD["edit_distance"], rel=1e-12)
    assert U.normalized_edit_distance([a for a, _ in pairs], [b for _, b in pairs]) == pytest.approx(GOLD["normalized_edit_distance"], rel=1e-12)
    inv = {i: ch for i, ch in enumerate(U.get_lexicon())}
    dec = U.DecodeCTCPred(top_paths=1, beam_width=10, inverse_classes=inv)
    for labels, text in GOLD["labels_to_text"]:
        assert dec.labels_to_text(labels) == text
    for labels, text in GOLD["labels_to_text_fn"]:
        assert U.labels_to_text(labels, inv) == text
    W, b = U.get_initial_weights(50)
    assert list(W.shape) == GOLD["initial_weights"]["W_shape"] and np.abs(W).max() == 0 and b.tolist() == GOLD["initial_weights"]["b"]


def test_early_stopping_iter_trace_matches_reference():
    g = GOLD["early_stopping"]

    class _M:
        stop_training = False
        w = 0

        def get_weights(self):
            return self.w

        def set_weights(self, w):
            self.w = w

    es = U.EarlyStoppingIter(monitor="loss", min_delta=g["min_delta"], patience=g["patience"], restore_best_weights=True, mode="auto")
    es.set_model(_M())
    es.on_train_begin()
    trace = []
    for i, l in enumerate(g["losses"]):
        es.model.w = i
        es.on_batch_end(i, {"loss": l})
        trace.append([bool(es.model.stop_training), float(es.best), int(es.stopped_iter)])
        if es.model.stop_training:
            break
    assert trace == g["trace"] and es.model.w == g["final_weights"]


def test_library_exports_every_declared_symbol_and_layout_is_keras_order():
    decl = native.parse_header()
    assert len(decl) >= 40
    lib = ctypes.CDLL(native.LIB_PATH)          # loads without a GPU
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    # measurement / test hooks are NOT in the product library or its header: their own .so (include/crnn_testhooks.h)
    assert not [n for n in decl if n.startswith("crnn_debug")]
    hdecl = native.parse_header(native.HOOKS_HEADER)
    hlib = ctypes.CDLL(native.HOOKS_PATH)
    assert set(hdecl) == {"crnn_debug_copy", "crnn_debug_occupy"} and all(hasattr(hlib, n) and not hasattr(lib, n) for n in hdecl)
    cfg = _cfg_struct(64, (100, 32, 1), 38, 23, 128, 256, False)
    lay = param_layout(cfg)
    names = list(lay)
    assert names[:8] == ["stn_c1_k", "stn_c1_b", "stn_c2_k", "stn_c2_b", "stn_d1_w", "stn_d1_b", "stn_d2_w", "stn_d2_b"]
    assert names[-2:] == ["dense2_w", "dense2_b"] and len(names) == 66
    assert sum(int(np.prod(d)) for _, _, d in lay.values()) == 3282865      # LSTM variant
    assert all(off % 4 == 0 for off, _, _ in lay.values())                    # 16-byte aligned tensors
    L = native.lib()
    assert L.crnn_bn_total(ctypes.byref(cfg)) == 3969 and L.crnn_time_steps(ctypes.byref(cfg)) == 52
    cfg200 = _cfg_struct(8, (200, 32, 1), 38, 21, 128, 256, False)
    assert L.crnn_time_steps(ctypes.byref(cfg200)) == 102 and param_layout(cfg200)["stn_d1_w"][2] == (1760, 50)
    assert L.crnn_workspace_bytes(ctypes.byref(cfg)) > 0
    bad = _cfg_struct(8, (100, 32, 1), 38, 23, 128, 100, False)               # units not a multiple of 64
    assert L.crnn_workspace_bytes(ctypes.byref(bad)) == 0


def test_schedule_flags_of_the_binding_equal_the_header_and_row_stream_shape_rules():
    """native.FLAG_* mirror the CRNN_FLAG_* macros of include/crnn_mi355x.h (the header is the contract: a drifted constant would silently
    select another schedule), and the shape rules of the two row-stream depthwise kernels -- host arithmetic, no GPU -- accept every block of
    the CRNN at the 100x32 and 200x32 input shapes and refuse maps whose rows do not fill the step row."""
    import re
    text = open(native.HEADER).read()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+CRNN_FLAG_(\w+)\s+(\d+)", text)}
    ours = {k[5:]: v for k, v in vars(native).items() if k.startswith("FLAG_")}
    assert macros == ours and len(set(macros.values())) == len(macros), (macros, ours)
    assert all(v & (v - 1) == 0 for v in macros.values())                    # single bits
    L = ctypes.CDLL(native.LIB_PATH)
    for B in (1, 64, 256, 1024):
        for H in (104, 204):
            for (h, w, c) in [(H, 36, 64), (H, 36, 128), (H // 2, 18, 256), (H // 2, 18, 256), (H // 2, 9, 512), (H // 2, 9, 512)]:
                assert L.crnn_dwconv_fwd_stream_supported(B, h, w, c) == 0 and L.crnn_dwconv_bwd_stream_supported(B, h, w, c) == 0, (B, h, w, c)
                rows_f, rows_b = L.crnn_dwconv_fwd_stream_rows(B, h, w, c), L.crnn_dwconv_bwd_stream_rows(B, h, w, c)
                assert rows_f >= B and rows_f % B == 0 and h % (rows_f // B) == 0          # whole row bands per image
                assert rows_b >= B and rows_b % B == 0 and h % (rows_b // B) == 0
    # round 5: image widths 48 and 64 (step rows of 416 / 544 columns, bf16 rows cut into channel ranges) take the row-stream kernels in both storage types
    for fn in ("crnn_dwconv_fwd_stream_supported_ex", "crnn_dwconv_bwd_stream_supported_ex"):
        getattr(L, fn).restype = ctypes.c_int
    for imgw in (48, 64):
        h, w, cin = 104, imgw + 4, 1
        for i, (co, ph, pw) in enumerate(((64, 1, 1), (128, 1, 1), (256, 2, 2), (256, 1, 1), (512, 1, 2), (512, 1, 1), (512, 1, 1)), 1):
            if i >= 2:
                for B in (5, 64, 256):
                    assert L.crnn_dwconv_fwd_stream_supported(B, h, w, cin) == 0 and L.crnn_dwconv_bwd_stream_supported(B, h, w, cin) == 0, (imgw, B, h, w, cin)
                    assert L.crnn_dwconv_fwd_stream_supported_ex(B, h, w, cin, 0) == 0 and L.crnn_dwconv_bwd_stream_supported_ex(B, h, w, cin, 0) == 0, (imgw, B, h, w, cin)
                    rows_f = L.crnn_dwconv_fwd_stream_rows(B, h, w, cin)
                    assert rows_f >= B and rows_f % B == 0 and h % (rows_f // B) == 0
            h, w, cin = h // ph, w // pw, co
    # refused: step rows that fill fewer than five compute waves (forward) / no more than two (backward), channel counts that are no whole groups of 8
    for (h, w, c) in [(13, 18, 64), (52, 18, 252), (13, 9, 128), (104, 36, 1)]:
        assert L.crnn_dwconv_fwd_stream_supported(8, h, w, c) == -3 and L.crnn_dwconv_fwd_stream_rows(8, h, w, c) == 0
    for (h, w, c) in [(52, 18, 252), (104, 36, 1), (8, 4, 64), (13, 9, 64)]:
        assert L.crnn_dwconv_bwd_stream_supported(8, h, w, c) == -3 and L.crnn_dwconv_bwd_stream_rows(8, h, w, c) == 0


def test_fp32_row_stream_and_localisation_net_shape_rules():
    """Round 4, host arithmetic only: the fp32 forms of the row-stream kernels (rows of 18 KiB as channel ranges) take every block of the CRNN at the
    100x32 and 200x32 shapes, with the same number of statistic rows per image band as the bf16 forms report bands; the one-workgroup-per-sample
    localisation net takes images whose first convolution map pools into whole windows and fits LDS."""
    L = ctypes.CDLL(native.LIB_PATH)
    for fn in ("crnn_dwconv_fwd_stream_supported_ex", "crnn_dwconv_fwd_stream_rows_ex", "crnn_dwconv_bwd_stream_supported_ex", "crnn_dwconv_bwd_stream_rows_ex",
               "crnn_dwconv_fwd_stream_pro_supported_ex", "crnn_dwconv_bwd_stream_pro_supported_ex", "crnn_loc_net_fused_supported"):
        getattr(L, fn).restype = ctypes.c_int
    L.crnn_loc_net_bwd_scratch.restype = ctypes.c_long
    F32, BF16 = 0, 1
    for B in (1, 64, 256):
        for H in (104, 204):
            for (h, w, c) in [(H, 36, 64), (H, 36, 128), (H // 2, 18, 256), (H // 2, 9, 512)]:
                for dt in (F32, BF16):
                    assert L.crnn_dwconv_fwd_stream_supported_ex(B, h, w, c, dt) == 0 and L.crnn_dwconv_bwd_stream_supported_ex(B, h, w, c, dt) == 0, (B, h, w, c, dt)
                    assert L.crnn_dwconv_fwd_stream_pro_supported_ex(B, h, w, c, dt) == 0 and L.crnn_dwconv_bwd_stream_pro_supported_ex(B, h, w, c, dt) == 0
                    for rows in (L.crnn_dwconv_fwd_stream_rows_ex(B, h, w, c, dt), L.crnn_dwconv_bwd_stream_rows_ex(B, h, w, c, dt)):
                        assert rows >= B and rows % B == 0 and h % (rows // B) == 0
                assert L.crnn_dwconv_fwd_stream_rows_ex(B, h, w, c, BF16) == L.crnn_dwconv_fwd_stream_rows(B, h, w, c)
    assert L.crnn_dwconv_fwd_stream_supported_ex(8, 13, 9, 64, F32) == -3 and L.crnn_dwconv_bwd_stream_supported_ex(8, 104, 36, 1, F32) == -3
    assert L.crnn_dwconv_fwd_stream_supported_ex(8, 104, 52, 64, F32) == 0 and L.crnn_dwconv_fwd_stream_pro_supported_ex(8, 104, 52, 64, F32) == -3   # (round 5: width 48 streams; its prologue form stays with the 9 KiB step row)
    assert L.crnn_dwconv_fwd_stream_supported_ex(8, 104, 36, 64, 7) == -2
    assert L.crnn_loc_net_fused_supported(100, 32) == 0 and L.crnn_loc_net_fused_supported(200, 32) == 0 and L.crnn_loc_net_fused_supported(40, 32) == 0
    assert L.crnn_loc_net_fused_supported(60, 48) == 0
    assert L.crnn_loc_net_fused_supported(30, 32) == -3          # first convolution map 11 x 12: an odd side, no whole pooling windows
    assert L.crnn_loc_net_fused_supported(1000, 32) == -3        # a sample's maps beyond LDS
    assert L.crnn_loc_net_bwd_scratch(256) == 256 * (25 * 20 * 20 + 20 + 25 * 20 + 20)


def test_dense_backward_shape_rule_and_scratch_size():
    """Round 5, host arithmetic only: dense2's one-pass backward (dense.hip) takes 2 * units a multiple of 128 up to 512 (units is a multiple of 64) and at most 40 classes; its scratch is
    one partial gradient ([K][C] + [C], rounded up to whole 16 bytes) per workgroup, at most 256 workgroups of at least one 8-row step."""
    L = ctypes.CDLL(native.LIB_PATH)
    L.crnn_dense_bwd_small_scratch_bytes.restype = ctypes.c_size_t
    L.crnn_dense_bwd_small_scratch_bytes.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
    L.crnn_dense_bwd_small_supported.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
    for (M, K, C, want) in [(13312, 512, 38, 0), (3328, 512, 38, 0), (52 * 8, 128, 11, 0), (7, 128, 2, 0), (7, 64, 2, -3), (64, 192, 38, -3), (13312, 512, 40, 0), (13312, 512, 41, -3),
                            (13312, 640, 38, -3), (13312, 1024, 38, -3), (13312, 96, 38, -3), (13312, 512, 64, -3), (0, 512, 38, -3)]:
        assert L.crnn_dense_bwd_small_supported(M, K, C) == want, (M, K, C)
        nb = L.crnn_dense_bwd_small_scratch_bytes(M, K, C)
        if want:
            assert nb == 0
        else:
            per = (K * C + C + 3) // 4 * 4 * 4
            assert nb % per == 0 and 1 <= nb // per <= 256 and nb // per <= (M + 7) // 8
    assert L.crnn_dense_bwd_small_scratch_bytes(13312, 512, 38) == 256 * (512 * 38 + 38 + 2) * 4        # 52 rows per workgroup
    assert L.crnn_dense_bwd_small_scratch_bytes(3328, 512, 38) == 256 * (512 * 38 + 38 + 2) * 4         # batch 64: 13 rows per workgroup


def test_dense1_stream_kernels_shape_rules():
    """Round 5, host arithmetic only: dense1's forward on the stripe stream takes whole 64-row stripes, 128 or 256 output columns and whole 64-k chunks; its
    weight gradient on the pixel stream takes up to 8192 output rows (64 tiles of 128) and sizes its scratch by cus / tiles row ranges when the tiles
    outnumber an XCD's CUs (36 tiles -> 7 ranges); its data gradient on the weights-resident kernel takes up to 8192 output columns at K <= 128."""
    L = ctypes.CDLL(native.LIB_PATH)
    L.crnn_dense_fwd_stream_supported.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_long]
    L.crnn_gemm_tn_bf16_stream_supported.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
    L.crnn_gemm_tn_bf16_stream_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
    L.crnn_gemm_tn_bf16_stream_scratch_bytes.restype = ctypes.c_size_t
    for (M, N, K, want) in [(13312, 128, 4608, 0), (3328, 128, 4608, 0), (26112, 256, 4608, 0), (416, 128, 4608, -3), (13312, 64, 4608, -3), (13312, 128, 4600, -3), (0, 128, 4608, -3)]:
        assert L.crnn_dense_fwd_stream_supported(M, N, K) == want, (M, N, K)
    for (M, N, K, want) in [(4608, 128, 13312, 0), (4608, 128, 3328, 0), (9216, 128, 26112, -3), (4608, 128, 416, -3), (4600, 128, 13312, -3), (4608, 96, 13312, -3)]:
        assert L.crnn_gemm_tn_bf16_stream_supported(M, N, K) == want, (M, N, K)
    assert L.crnn_gemm_tn_bf16_stream_scratch_bytes(4608, 128, 13312) == 7 * 4608 * 128 * 4          # 36 tiles x 7 row ranges (256 CUs)
    assert L.crnn_gemm_tn_bf16_stream_scratch_bytes(4600, 128, 13312) == 0
    assert L.crnn_gemm_wres_supported(4608, 128) == 0 and L.crnn_gemm_wres_supported(4608, 256) == -3 and L.crnn_gemm_wres_supported(8320, 128) == -3


def test_row_stream_kernels_keep_their_row_loops_spill_free():
    """The bf16 row-stream kernels sit at their 168-register ceiling; a harmless-looking edit (an address spelled with one multiplication instead of two)
    once put a 16-byte spill into the depthwise-stage backward's row loop and cost the kernel 22 %.  hipcc cross-compiles gfx950 without a GPU: no
    instantiation of the two files may have scratch traffic inside a loop (scripts/check_loop_spills.sh is the same check by hand)."""
    import re, shutil, subprocess, tempfile
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    csrc = os.path.join(os.path.dirname(native.LIB_PATH), "csrc")
    inc = os.path.dirname(native.HEADER)
    for src in ("dwconv_stream.hip", "dwconv_bwd_stream.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", inc, "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", out],
                                  stderr=subprocess.DEVNULL)
            text = open(out).read()
        kernels = list(re.finditer(r"^(_ZN\S*dw_\w+_stream_kernel\S*):", text, re.M))
        assert len(kernels) >= 7, (src, len(kernels))
        for m in kernels:
            body = text[m.start():text.index(".Lfunc_end", m.start())]
            in_loop, bad = False, []
            for line in body.split("\n"):
                if re.match(r"^(\.LBB\S+:|; %bb\.\d+:)", line):
                    in_loop = ("in Loop" in line) or ("Loop Header" in line)
                if "scratch_" in line and in_loop:
                    bad.append(line.strip())
            assert not bad, "%s: %s spills inside a loop: %s" % (src, m.group(1), bad[:3])


def test_weights_resident_gemms_and_row_stream_kernels_have_no_spilled_vector_registers():
    """VERDICT round 4, hygiene: an instantiation of the weights-resident pointwise kernels (gemm_wres.hip; `gemm_wres_fwd_kernel<1, 2, 2>` sat at 256
    registers with 47 spilled and 192 bytes of scratch per lane until K = 64 was routed to the narrow shape) or of the streaming weight-gradient kernel
    must not spill vector registers at all: their IO / storer waves' steady state has no slack for scratch traffic (round 6: the parity mode's plane GEMMs too --
    gemm_pres.hip budgets up to 512 registers per wave and must stay clear of scratch).  The row-stream depthwise kernels may keep
    the handful of spills outside their row loops that the loop check above allows.  hipcc's resource remarks, cross-compiled without a GPU."""
    import re, shutil, subprocess, tempfile
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    csrc = os.path.join(os.path.dirname(native.LIB_PATH), "csrc")
    inc = os.path.dirname(native.HEADER)
    for src, minimum, allowed in (("gemm_wres.hip", 15, 0), ("gemm_wres3.hip", 12, 0), ("gemm_pres.hip", 9, 0), ("gemm_wgrad3.hip", 4, 0), ("gemm_wgrad.hip", 2, 0), ("dwconv_stream.hip", 7, 4), ("dwconv_bwd_stream.hip", 7, 4),
                                  ("dense.hip", 2, 0)):
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", inc, "-c", os.path.join(csrc, src), "-o", os.path.join(td, "k.o"),
                                "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
        assert len(names) == len(spills) and len(names) >= minimum, (src, len(names), len(spills))
        bad = {n: v for n, v in zip(names, spills) if v > allowed}
        assert not bad, "%s: instantiations with spilled vector registers: %s" % (src, bad)
        if src == "dwconv_bwd_stream.hip":
            # the step's own instantiation, whole kernel (round 6: 3 -> 2 spilled registers, 8 bytes of scratch per lane, none inside a loop: the lane index and the
            # dx row offset are re-formed where they are used; what is left are two column constants of the DK waves the allocator parks across the row loop)
            step = [v for n, v in zip(names, spills) if "dw_bwd_stream_kernelILi3ELb0ELb0ELb0ELb0E" in n]
            assert step and step[0] <= 2, step


def test_bench_counter_traffic_lookup_and_contract_fields():
    """bench.py, host logic only: `pmc_step_traffic` / `pmc_mfma_util` read the committed counter passes (profiles/r06_pmc_step_<mode>.json) for the workload they
    were collected on -- batch 256, 100x32, LSTM, default flags -- and return None for any other (a `traffic` must never be quoted for a configuration it was
    not counted on); the contract fields the driver's record keeps (`config.*`) are spelled in the source."""
    import importlib.util, types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(native.HEADER), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    def eng(B=256, imgh=100, imgw=32, gru=0, flags=0, precision="bf16s"):
        return types.SimpleNamespace(B=B, precision=precision, cfg=types.SimpleNamespace(imgh=imgh, imgw=imgw, gru=gru, flags=flags))
    tr = bench.pmc_step_traffic(eng(), ["dw_bwd_stream_kernel"])
    assert tr is not None and tr[1] == 6 and 3.4e9 < tr[0] < 3.7e9          # six launches per step, 3.435 GB algorithmic
    tr = bench.pmc_step_traffic(eng(), ["bn_act_pool_drop_kernel"])
    assert tr is not None and tr[1] == 7 and 2.6e9 < tr[0] < 2.8e9          # (round 6: + q at the pool windows' arg-max of blocks 3 and 5, 0.18 GB)
    nt = bench.pmc_step_traffic(eng(), ["gemm_nt_f32_proj_kernel"])
    assert nt is not None and nt[1] == 2                                      # the two layers' input projections (both directions per launch)
    one = bench.pmc_step_traffic(eng(), ["gemm_nt_f32_stream_kernel"], {"gemm_nt_f32_stream_kernel": {159744}})
    assert one is not None and one[1] == 4                                    # (grid filter: the stripe kernel's four launches -- dense1 and dense2 forward, the two layers' input gradients)
    for other in (eng(B=64), eng(imgh=200), eng(gru=1), eng(flags=1024), eng(imgw=48)):
        assert bench.pmc_step_traffic(other, ["dw_bwd_stream_kernel"]) is None and bench.pmc_mfma_util(other, ["gemm_wres_fwd_kernel"]) is None
    mf = bench.pmc_mfma_util(eng(), ["gemm_wres_fwd_kernel", "lstm_fwd_persist_kernel", "no_such_kernel"])
    assert 0.1 < mf[0] < 0.4 and 0.01 < mf[1] < 0.2 and mf[2] is None
    fp = bench.pmc_mfma_util(eng(precision="fp32"), ["gemm_x3p_kernel"])
    assert fp is not None and 0.1 < fp[0] < 0.6
    src = open(bench.__file__).read()
    for key in ("parity_within_tolerance_fp32", "parity_within_tolerance_headline", "parity_mode_ms_per_step", "parity_mode_strict_ms_per_step",
                "bs64_ms_per_step", "bs64_fp32_ms_per_step", "dw_fwd_hbm_frac_cold", "dw_bwd_hbm_frac_cold"):
        assert 'cf["%s"]' % key in src, key


def test_model_surface_weights_roundtrip_without_gpu(tmp_path):
    init_model = U.CRNN(num_classes=38, shape=(100, 32, 1), max_string_len=23, time_dense_size=128, n_units=256)
    model = init_model.get_model()
    assert (init_model.pooling_counter_h, init_model.pooling_counter_w) == (1, 2)   # downsample_factor = 2 (train.py:202)
    ws = model.get_weights()
    assert len(ws) == 94 and ws[8].shape == (3, 3, 1, 1) and ws[13].shape == (1, 1, 1, 64)
    assert model.count_params() == (3282865, 7938)
    np.testing.assert_array_equal(ws[7], [1, 0, 0, 0, 1, 0])                         # identity STN (utils.py:239-245)
    assert np.abs(ws[6]).max() == 0
    os.makedirs(tmp_path / "m")
    U.save_model_json(model, str(tmp_path), "m")
    model.save_weights(str(tmp_path / "m" / "final_weights.h5"))
    m2 = U.load_custom_model(str(tmp_path / "m"))
    for a, b in zip(ws, m2.get_weights()):
        np.testing.assert_array_equal(a, b)
    pred = U.init_predictor(m2)
    assert pred.predictor and pred.get_layer("softmax").output.shape == (None, 52, 38)
    lines = []
    model.summary(print_fn=lines.append)
    assert any("Total params: 3,290,803" in l for l in lines)                        # LSTM total incl. BN statistics


def _write_png(path, w, h, seed):
    from PIL import Image
    rs = np.random.RandomState(seed)
    a = np.full((h, w), 230, np.uint8)
    a[h // 3: 2 * h // 3, w // 5: 4 * w // 5] = rs.randint(0, 60, size=(2 * h // 3 - h // 3, 4 * w // 5 - w // 5))
    Image.fromarray(a).save(path)


def test_readf_generator_contract_and_first_pass_tail(tmp_path):
    words = ["hello", "world", "ab", "q9z", "seven"]
    names = []
    for i, wd in enumerate(words):
        p = str(tmp_path / ("%d_%s_%d.png" % (i, wd, i)))
        _write_png(p, 60 + 7 * i, 24, i)
        names.append(p)
    np.random.seed(0)
    reader, classes = _reader(batch_size=2)
    gen = reader.run_generator(names, downsample_factor=2)
    b1, o1 = next(gen)
    assert set(b1) == {"the_input", "the_labels", "input_length", "label_length", "source_str"} and o1["ctc"].shape == (2,)
    assert b1["the_input"].shape == (2, 100, 32, 1) and b1["the_input"].dtype == np.float64
    assert b1["input_length"].ravel().tolist() == [50.0, 50.0] and b1["label_length"].ravel().tolist() == [5.0, 5.0]
    assert b1["the_labels"][0, :5].tolist() == [classes[c] for c in "hello"] and b1["the_labels"][0, 5] == 37
    assert list(b1["source_str"]) == ["hello", "world"]
    next(gen)
    b3, _ = next(gen)                      # tail of the first pass: full-size arrays, one fresh row
    assert list(b3["source_str"]) == ["seven"] and b3["the_input"].shape == (2, 100, 32, 1)
    b4, _ = next(gen)                      # second pass continues filling the same arrays (counter never resets)
    assert list(b4["source_str"]) == ["seven", "hello"]
    x = b1["the_input"]
    assert abs(float(x.mean())) < 4 and np.isfinite(x).all()
    img, word = U.open_img(names[0], (100, 32, 1), p=0.)
    assert img.shape == (100, 32) and img.dtype == np.uint8 and word == "hello"
    assert (img > 127).mean() < 0.5        # bright background was inverted (utils.py:402-405)


def _dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crnn-ocr-lite_amd"))
    from crnn_mi355x.parallel import allreduce_mean_, shard
    from oracle import model as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = M.Config(imgh=36, imgw=32, num_classes=7, max_len=4, time_dense_size=12, n_units=8)
    p, bn = M.init_params(cfg, seed=5, dtype=np.float64)
    x, lab, il, ll = M.synthetic_batch(cfg, 4, seed=2, dtype=np.float64)
    lo, hi = shard(4, rank, world)
    # per-replica BatchNorm statistics (no SyncBN): each rank differentiates its own shard
    _, _, g, _ = M.loss_and_grads(cfg, p, bn, x[lo:hi], lab[lo:hi], il[lo:hi], ll[lo:hi])
    flat = torch.from_numpy(np.concatenate([g[k].ravel() for k in p]))
    staged = flat.clone()
    allreduce_mean_(flat, dist, world)
    # the overlapped form used by Engine.train_step: tail of the buffer first (asynchronously), then the head, then join
    from crnn_mi355x.parallel import GradAllReduce
    ar = GradAllReduce(None, dist, world, overlap=True)
    split = staged.numel() // 3
    ar.start(staged[split:]); ar.start(staged[:split]); ar.finish(staged)
    assert torch.equal(staged, flat), "two-stage all-reduce differs from the single one"
    # replicas that start from different weights adopt rank 0's; moving statistics are averaged on demand
    from types import SimpleNamespace
    from crnn_mi355x.parallel import broadcast_state, sync_bn_stats
    fake = SimpleNamespace(params=torch.full((5,), float(rank + 1)), bn_mean=torch.full((3,), float(rank)), bn_var=torch.full((3,), 1.0 + rank))
    broadcast_state(fake, dist, world)
    assert torch.equal(fake.params, torch.full((5,), 1.0)) and torch.equal(fake.bn_mean, torch.zeros(3)) and torch.equal(fake.bn_var, torch.ones(3))
    fake.bn_mean += rank; fake.bn_var += 2 * rank
    sync_bn_stats(fake, dist, world)
    assert torch.allclose(fake.bn_mean, torch.full((3,), 0.5)) and torch.allclose(fake.bn_var, torch.full((3,), 2.0))
    q.put((rank, flat.numpy().copy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_shards_are_equal_sized_so_every_rank_issues_the_same_collectives():
    from crnn_mi355x.parallel import shard
    assert [shard(51, r, 2) for r in range(2)] == [(0, 25), (25, 50)]
    assert [shard(8, r, 8) for r in range(8)] == [(r, r + 1) for r in range(8)]
    assert shard(10, 0, 1) == (0, 10)
    sizes = {hi - lo for lo, hi in (shard(1003, r, 8) for r in range(8))}
    assert sizes == {125}


def test_bench_gpus_flag_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` outside a launcher re-executes itself under torch.distributed.run with N processes."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setenv("CRNN_DIST_BACKEND", "gloo")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("CRNN_DIST_BACKEND", "nccl")
    assert bench.self_launch(4) == 2          # no GPUs here: refuses instead of silently running one rank


def test_data_parallel_gradient_allreduce_world2_gloo():
    import torch.multiprocessing as mp
    from oracle import model as M
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
    np.testing.assert_array_equal(res[0][1], res[1][1])           # identical averaged gradient on both ranks
    assert res[0][2] == (0, 2) and res[1][2] == (2, 4)
    cfg = M.Config(imgh=36, imgw=32, num_classes=7, max_len=4, time_dense_size=12, n_units=8)
    p, bn = M.init_params(cfg, seed=5, dtype=np.float64)
    x, lab, il, ll = M.synthetic_batch(cfg, 4, seed=2, dtype=np.float64)
    gs = [M.loss_and_grads(cfg, p, bn, x[a:b], lab[a:b], il[a:b], ll[a:b])[2] for a, b in ((0, 2), (2, 4))]
    ref = np.concatenate([(0.5 * (gs[0][k] + gs[1][k])).ravel() for k in p])
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-12, atol=1e-15)


def test_reference_keras_model_json_artefacts_load_as_the_same_architecture():
    """the Keras-2.2.2 model.json files the reference ships (facts kept in tests/golden/
    keras_model_json.json by make_golden.py) must build the very architecture their model_summary.txt counts."""
    arts = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "keras_model_json.json")))
    assert set(arts) == {"OCR_IAM_ver1", "OCR_Stickies_ver1", "OCR_mjsynth_FULL_2"}
    for name, art in arts.items():
        m = U.model_from_json(json.dumps(art["model_json"]))
        c = m.config
        assert c["GRU"] is True and c["n_units"] == 256 and c["time_dense_size"] == 128 and c["num_classes"] == 38, (name, c)
        assert tuple(c["shape"]) == (100, 32, 1)
        assert c["max_string_len"] == (21 if name == "OCR_IAM_ver1" else c["max_string_len"])
        trainable, non_trainable = m.count_params()
        assert trainable == art["param_counts"]["Trainable params"] == 2823089
        assert non_trainable == art["param_counts"]["Non-trainable params"] == 7938
        assert not m.predictor
    # a graph that is not CRNN.get_model's is refused, not silently rebuilt
    bad = json.loads(json.dumps(arts["OCR_mjsynth_FULL_2"]["model_json"]))
    bad["config"]["layers"] = [l for l in bad["config"]["layers"] if l["class_name"] != "DepthwiseConv2D"][:-1]
    with pytest.raises(ValueError):
        U.model_from_json(json.dumps(bad))
    # this package's own to_json round-trips ...
    own = U.CRNN(num_classes=38, shape=(100, 32, 1), GRU=False, max_string_len=23).get_model()
    assert U.model_from_json(own.to_json()).config == own.config
    # ... and IS the Keras-2.2.2 functional-model JSON of the same graph: reduced to the keys make_golden.py kept from the
    # reference's models/*/model.json it equals those artefacts layer by layer (names, order, hyper-parameters), and it carries the
    # structural keys Keras' model_from_json needs (inbound_nodes of every layer, input_layers, output_layers)
    keep = ("batch_input_shape", "units", "filters", "kernel_size", "pool_size", "padding", "rate", "merge_mode", "activation",
            "axis", "momentum", "epsilon", "output_size", "depth_multiplier", "use_bias", "max_value")
    for name, art in arts.items():
        ref_layers = art["model_json"]["config"]["layers"]
        max_len = [l for l in ref_layers if l["name"] == "the_labels"][0]["config"]["batch_input_shape"][1]
        mine = json.loads(U.CRNN(num_classes=38, shape=(100, 32, 1), GRU=True, max_string_len=max_len).get_model().to_json())
        assert (mine["class_name"], mine["keras_version"], mine["backend"]) == ("Model", "2.2.2", "tensorflow")
        red = []
        for l in mine["config"]["layers"]:
            cfg = {k: l["config"][k] for k in keep if k in l["config"]}
            if l["class_name"] == "Bidirectional":
                inner = l["config"]["layer"]
                cfg["layer"] = {"class_name": inner["class_name"], "config": {k: inner["config"][k] for k in
                                ("units", "activation", "recurrent_activation", "return_sequences", "implementation", "reset_after") if k in inner["config"]}}
            red.append({"class_name": l["class_name"], "name": l["name"], "config": cfg})
        assert red == ref_layers, name
        names = {l["name"] for l in mine["config"]["layers"]}
        for l in mine["config"]["layers"]:
            assert l["config"]["name"] == l["name"]
            for node in l["inbound_nodes"]:
                assert all(src[0] in names for src in node)
            assert (l["class_name"] == "InputLayer") == (l["inbound_nodes"] == [])
        assert mine["config"]["input_layers"] == [[n, 0, 0] for n in ("the_input", "the_labels", "input_length", "label_length")]
        assert mine["config"]["output_layers"] == [["ctc", 0, 0]]
    pred = json.loads(U.init_predictor(own).to_json())
    assert pred["config"]["output_layers"] == [["softmax", 0, 0]] and not any(l["class_name"] == "Lambda" for l in pred["config"]["layers"])
    assert U.model_from_json(json.dumps(pred)).predictor


# ------------------------------------------------------------------------------------------------ parallel loader
def _write_images(folder, n, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed)
    words = ["hello", "world", "overfilled", "cellist", "amd", "ocr", "keras", "x"]
    names = []
    for i in range(n):
        a = (rs.rand(20 + i % 9, 40 + 7 * (i % 11), 3) * 255).astype(np.uint8)
        path = os.path.join(str(folder), "%d_%s_%d.png" % (i, words[i % len(words)], i))
        Image.fromarray(a).save(path)
        names.append(path)
    return names


def _take(gen, n):
    out = []
    for _ in range(n):
        inputs, _ = next(gen)
        out.append({k: np.array(v) for k, v in inputs.items()})
    return out


def test_modal_value_fast_path_equals_unique_argmax():
    from crnn_mi355x import data as D
    rs = np.random.RandomState(1)
    for trial in range(50):
        a = rs.randint(0, 1 + trial % 7 * 40, size=(rs.randint(1, 30), rs.randint(1, 30))).astype(np.uint8 if trial % 2 else np.int64)
        vals, counts = np.unique(a, return_counts=True)
        assert D._modal_value(a) == vals[np.argmax(counts)]
    assert D._modal_value(np.where(np.zeros((3, 3)) > 1, 255, 0)) == 0


def test_worker_processes_yield_the_serial_loaders_batches(tmp_path):
    """Readf(workers=N): same visiting order, same arrays.  Without random padding (predict.py uses transform_p=0) the
    batches are identical to the single-threaded loop, short first-pass tail and wrap-around included; with random
    padding they depend only on `seed`, not on the worker count; page + bounding-box inputs go through the same path."""
    import utils as U
    names = _write_images(tmp_path, 37)
    classes = {c: i for i, c in enumerate(U.get_lexicon())}
    kw = dict(img_size=(100, 32, 1), max_len=23, normed=True, batch_size=8, classes=classes)
    serial = _take(U.Readf(transform_p=0., **kw).run_generator(names), 7)          # 4 full + tail of 5 + wrap
    par = U.Readf(transform_p=0., workers=2, chunk=5, **kw)
    got = _take(par.run_generator(names), 7)
    par.close()
    for i, (a, b) in enumerate(zip(serial, got)):
        rows = 5 if i == 4 else 8            # the first-pass tail re-yields full-size arrays whose last rows are np.empty garbage
        for k in a:
            assert np.array_equal(a[k][:rows], b[k][:rows]), (i, k)
    # random padding: reproducible per seed across worker counts, different across seeds
    runs = []
    for workers, seed in ((2, 5), (3, 5), (2, 6)):
        r = U.Readf(transform_p=0.7, workers=workers, seed=seed, chunk=4, **kw)
        runs.append(_take(r.run_generator(names), 3))
        r.close()
    assert all(np.array_equal(a["the_input"], b["the_input"]) for a, b in zip(runs[0], runs[1]))
    assert any(not np.array_equal(a["the_input"], b["the_input"]) for a, b in zip(runs[0], runs[2]))
    # pages with word boxes (predict.py --validate on IAM-style annotations): (word, x0, y0, x1, y1) crops
    from PIL import Image
    page = str(tmp_path / "page.png")
    Image.fromarray((np.random.RandomState(3).rand(120, 300, 3) * 255).astype(np.uint8)).save(page)
    boxes = {page: [("ab", 5, 5, 40, 90), (None, 50, 10, 100, 200), ("xyz", 10, 100, 60, 280)]}
    s2 = _take(U.Readf(transform_p=0., **dict(kw, batch_size=3)).run_generator([page], bboxs=boxes), 2)
    p2r = U.Readf(transform_p=0., workers=2, **dict(kw, batch_size=3))
    p2 = _take(p2r.run_generator([page], bboxs=boxes), 2)
    p2r.close()
    for a, b in zip(s2, p2):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
