#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over scripts/dw_bench.py (scripts/gpu_visit.sh pmcdw) into
profiles/<round>_pmc_dwconv.json: HBM bytes per launch of dwconv_tile_kernel<0> for every shape of the step.
usage: pmc_summary.py <gpurun_out tag> <profiles prefix>      e.g.  pmc_summary.py r1o r01"""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pref = sys.argv[1], sys.argv[2]
B = 256
# launch grids of dw_bench.py: shape -> workgroups (C/slab x tiles), slab = 64 channels bf16 / 32 fp32
SHAPES = [("104x36x64", 104, 36, 64), ("104x36x128", 104, 36, 128), ("52x18x256", 52, 18, 256), ("52x9x512", 52, 9, 512)]
res = {"kernel": "dwconv_tile_kernel<0> (forward with statistics epilogue, and data gradient)", "batch": B,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) over scripts/dw_bench.py "
                 "(scripts/gpu_visit.sh pmcdw); FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
       "modes": {}}
for mode, esz in (("bf16", 2), ("fp32", 4)):
    per = {}
    if not all(os.path.exists(os.path.join(ROOT, "gpurun_out", "%s_pmc_%s_%s" % (tag, mode, c), "dw_counter_collection.csv"))
               for c in ("FETCH_SIZE", "WRITE_SIZE")):
        print("no %s counter passes for %s" % (mode, tag)); continue      # a round may collect one storage mode only
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(ROOT, "gpurun_out", "%s_pmc_%s_%s" % (tag, mode, c), "dw_counter_collection.csv")
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(src)):
            if "dwconv_tile_kernel<0" in r["Kernel_Name"] and r["Counter_Name"] == c:
                agg[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        per[c] = agg
        keep = os.path.join(ROOT, "profiles", "%s_pmc_dwconv_%s_%s.csv" % (pref, mode, c))
        with open(src) as f, open(keep, "w") as g:          # keep only the depthwise rows (the raw file also holds torch's fills)
            for i, line in enumerate(f):
                if i == 0 or "dwconv_tile_kernel" in line:
                    g.write(line)
    shapes = {}
    from math import ceil
    for name, h, w, ch in SHAPES:
        slab = 128 // esz
        th = 8 if w == 36 else (13 if w == 18 else 26)
        grid = (ch // slab) * B * ceil(h / th) * 256      # threads
        if not per["FETCH_SIZE"] or not per["WRITE_SIZE"]:
            break                                          # this pass ran the row-stream kernels only
        key = min(per["FETCH_SIZE"], key=lambda g: abs(g - grid))
        f, wv = per["FETCH_SIZE"][key], per["WRITE_SIZE"][key]
        rd = 2.0 * 1024 * sum(f) / len(f); wr = 1024.0 * sum(wv) / len(wv)
        alg = 2.0 * B * h * w * ch * esz
        shapes[name] = {"grid_threads": key, "launches_sampled": len(f), "FETCH_SIZE_KB_avg": sum(f) / len(f), "WRITE_SIZE_KB_avg": sum(wv) / len(wv),
                        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                        "algorithmic_bytes_per_launch": alg, "ratio": (rd + wr) / alg}
    res["modes"][mode] = {"storage_bytes_per_element": esz, "shapes": shapes}
# the row-stream forward (bf16 only): every shape launches the same grid, so the launches are told apart by their order in
# dw_bench.py (7 launches = 2 warm-up + 5 timed per shape, six shapes in the order of the step)
ORDER = ["104x36x64", "104x36x128", "52x18x256", "52x18x256", "52x9x512", "52x9x512"]
DIMS = {n: (h, w, ch) for n, h, w, ch in SHAPES}
try:
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(ROOT, "gpurun_out", "%s_pmc_bf16_%s" % (tag, c), "dw_counter_collection.csv")
        rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(src))
                if "dw_fwd_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        rows.sort()
        assert len(rows) == 7 * len(ORDER), len(rows)
        per[c] = [[v for _, v in rows[7 * i:7 * i + 7]] for i in range(len(ORDER))]
        keep = os.path.join(ROOT, "profiles", "%s_pmc_dwstream_%s.csv" % (pref, c))
        with open(src) as f, open(keep, "w") as g:
            for i, line in enumerate(f):
                if i == 0 or "dw_fwd_stream_kernel" in line:
                    g.write(line)
    shapes = {}
    for i, name in enumerate(ORDER):
        h, w, ch = DIMS[name]
        f, wv = per["FETCH_SIZE"][i], per["WRITE_SIZE"][i]
        rd = 2.0 * 1024 * sum(f) / len(f); wr = 1024.0 * sum(wv) / len(wv)
        alg = 2.0 * B * h * w * ch * 2
        shapes[name] = {"launches_sampled": len(f), "FETCH_SIZE_KB_avg": sum(f) / len(f), "WRITE_SIZE_KB_avg": sum(wv) / len(wv),
                        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                        "algorithmic_bytes_per_launch": alg, "ratio": (rd + wr) / alg}
    res["stream"] = {"kernel": "dw_fwd_stream_kernel (forward with statistics; bf16 storage)", "storage_bytes_per_element": 2, "shapes": shapes}
    print("stream", {k: round(v["ratio"], 3) for k, v in shapes.items()})
except Exception as e:
    print("no row-stream counters:", repr(e))
out = os.path.join(ROOT, "profiles", "%s_pmc_dwconv.json" % pref)
json.dump(res, open(out, "w"), indent=1)
for m, d in res["modes"].items():
    print(m, {k: round(v["ratio"], 3) for k, v in d["shapes"].items()})
