#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over scripts/dw_f32_pmc.py into profiles/<prefix>_pmc_dw_f32.json: HBM bytes per launch of the
fp32 row-stream kernels against their algorithmic bytes.   usage: pmc_f32_summary.py <gpurun_out tag> <profiles prefix>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from dw_f32_pmc import SHAPES, B, REP
tag, pref = sys.argv[1], sys.argv[2]
KERNELS = [("forward", "dw_fwd_stream_kernel<9, 4, false, false, false, true>", lambda n: 2 * n * 4),
           ("forward, BatchNorm-2 prologue + dropout", "dw_fwd_stream_kernel<9, 5, false, true, true, true>", lambda n: 2 * n * 4 + n // 8),
           ("depthwise-stage backward", "dw_bwd_stream_kernel<3, false, false, false, true>", lambda n: 4 * n * 4),
           ("depthwise-stage backward, prologue + BatchNorm-2 statistics", "dw_bwd_stream_kernel<4, true, true, true, true>", lambda n: 4 * n * 4 + n // 8)]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    src = os.path.join(ROOT, "gpurun_out", "%s_pmc_f32_%s" % (tag, c), "dw_counter_collection.csv")
    rows = list(csv.DictReader(open(src)))
    per[c] = {}
    for _, kn, _ in KERNELS:
        v = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in rows if kn in r["Kernel_Name"] and r["Counter_Name"] == c)
        assert len(v) == REP * len(SHAPES), (kn, len(v))
        per[c][kn] = [[x for _, x in v[REP * i:REP * i + REP]] for i in range(len(SHAPES))]
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) over scripts/dw_f32_pmc.py; FETCH_SIZE (KB) doubled per "
                 "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); last of %d launches per shape" % REP, "batch": B, "kernels": {}}
for label, kn, alg in KERNELS:
    shapes = {}
    ta = tb = 0.0
    for i, (h, w, ch) in enumerate(SHAPES):
        n = B * h * w * ch
        rd = 2.0 * 1024 * per["FETCH_SIZE"][kn][i][-1]; wr = 1024.0 * per["WRITE_SIZE"][kn][i][-1]
        shapes["%dx%dx%d" % (h, w, ch)] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "algorithmic_bytes": alg(n), "ratio": round((rd + wr) / alg(n), 4)}
        ta += rd + wr; tb += alg(n)
    res["kernels"][label] = {"kernel": kn, "shapes": shapes, "traffic_over_algorithmic": round(ta / tb, 4)}
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_pmc_dw_f32.json" % pref), "w"), indent=1)
for k, v in res["kernels"].items(): print("%-62s traffic / algorithmic = %.4f" % (k, v["traffic_over_algorithmic"]))
