#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 --kernel-trace csv: every launch between two adam_kernel
dispatches, with its grid, duration and the idle gap before it.  usage: trace_step.py <kernel_trace.csv> [step_index]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
args = [a for a in sys.argv[2:] if not a.startswith("--")]
k = int(args[0]) if args else len(adam) - 2
seg = rows[adam[k] + 1: adam[k + 1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0
busy = 0; agg = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    name = re.sub(r"unsigned short", "bf16", name)
    grid = "%sx%sx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
    if "--agg" not in sys.argv:
        print("%9.1f us  +%6.1f gap  %8.1f us  %-14s %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, grid, name[:90]))
    busy += e - s; prev = max(prev, e)
    a = agg.setdefault(name.split("<")[0], [0, 0]); a[0] += 1; a[1] += e - s
print("step span %.3f ms, kernel busy %.3f ms, %d launches" % ((prev - t0) / 1e6, busy / 1e6, len(seg)))
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("  %-32s %4d  %8.3f ms" % (n, c, t / 1e6))
