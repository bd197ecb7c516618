#!/usr/bin/env python3
"""Timeline of ONE predict iteration (forward at inference + beam search) from a rocprofv3 --kernel-trace csv of scripts/predict_bench.py:
the launches between the last two beam-search kernels.  usage: trace_predict.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
beam = [i for i, r in enumerate(rows) if "beam" in r["Kernel_Name"]]
# beam decode may be several kernels: an iteration ends at the last beam kernel of a run of beam kernels
ends = [i for k, i in enumerate(beam) if k + 1 == len(beam) or beam[k + 1] != i + 1]
a, b = ends[-3] + 1, ends[-2] + 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0; busy = 0
agg = collections.OrderedDict()
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    name = re.sub(r"unsigned short", "bf16", name)
    g = "%sx%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    print("%9.1f us  + %5.1f gap  %8.1f us  %-14s %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, g, name[:110]))
    busy += e - s; prev = e
    k = name.split("<")[0]; c, t = agg.get(k, (0, 0)); agg[k] = (c + 1, t + e - s)
print("iteration span %.3f ms, kernel busy %.3f ms, %d launches" % ((prev - t0) / 1e6, busy / 1e6, len(seg)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-34s %3d  %8.3f ms" % (k, c, t / 1e6))
