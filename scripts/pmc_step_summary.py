#!/usr/bin/env python3
"""Fold the counter passes of one GPU visit over the train step (scripts/gpu_visit.sh `pmc:` steps over bench.py, 5 steps each) into
profiles/<pref>_pmc_sq_step_<mode>.txt (per kernel: MFMA-pipe utilisation, where the wave cycles go, VALU / LDS / VMEM instruction mix, memory-side
bytes per launch) and profiles/<pref>_pmc_step_<mode>.json (bytes per launch by kernel and grid, what bench.py's *_roofline objects quote as `traffic`).

usage: pmc_step_summary.py <visit tag> <profiles prefix> <mode>      e.g.  pmc_step_summary.py r05c r05 bf16s
reads  gpurun_out/<tag>_pmc_<group>_<mode>/**/*counter_collection.csv  for group in sqa, sqb, fetch, write
       gpurun_out/<tag>_kernel_stats_<mode>.csv                          (rocprofv3 --stats of the same command: average duration per kernel)

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over
the 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16) -> utilisation = busy / (1024 x duration x 2.4 GHz); FETCH_SIZE is in KB and tallies 128-byte requests at
64 bytes on gfx950 (doubled here), WRITE_SIZE in KB; both count the L2's memory-side requests (last-level-cache hits included)."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pref, mode = sys.argv[1], sys.argv[2], sys.argv[3]
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
OUT = os.path.join(ROOT, "gpurun_out")
CLK = 2.4e9


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    if len(s) > 90: s = s[:90]
    return s.replace("unsigned short", "bf16").replace("unsigned int", "u32")


agg = collections.defaultdict(lambda: collections.defaultdict(float))     # (kernel, grid) -> counter -> sum
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for group in ("sqa", "sqb", "fetch", "write"):
    for f in glob.glob(os.path.join(OUT, "%s_pmc_%s_%s" % (tag, group, mode), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
dur = {}          # (kernel, grid threads) -> average seconds, from the counter-free trace of the same command (`prof:<mode>` step); else by name from --stats
tr = glob.glob(os.path.join(OUT, "%s_prof_%s" % (tag, mode), "**", "*kernel_trace.csv"), recursive=True)
if tr:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        acc[(short(r["Kernel_Name"]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dur = {k: sum(v) / len(v) * 1e-9 for k, v in acc.items()}
else:
    sf = os.path.join(OUT, "%s_kernel_stats_%s.csv" % (tag, mode))
    if os.path.exists(sf):
        for r in csv.DictReader(open(sf)):
            dur[short(r["Name"])] = float(r["AverageNs"]) * 1e-9
rows = []
js = {"source": "rocprofv3 --pmc passes (counters only, --kernel-trace) over `bench.py --steps 3 --warmup 2 --precision %s` (scripts/gpu_visit.sh %s), %d train steps per pass; "
                "bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per launch (memory-side requests of the L2s: last-level-cache hits included), in the step's own "
                "order and cache state" % (mode, tag, STEPS), "steps_per_pass": STEPS, "kernels": {}}
for k, c in agg.items():
    n = {cn: len(v) for cn, v in calls[k].items()}
    per = lambda cn: c.get(cn, 0.0) / max(1, n.get(cn, 0))
    wc = per("SQ_WAVE_CYCLES") or float("nan")
    d = dur.get(k, dur.get(k[0]))
    fetch, write = 2.0 * 1024 * per("FETCH_SIZE"), 1024.0 * per("WRITE_SIZE")
    ncall = max(n.values()) if n else 0
    row = {"kernel": k[0], "grid_threads": k[1], "calls_per_step": round(ncall / STEPS, 2), "avg_us": None if d is None else round(d * 1e6, 1),
           "mfma_util": None if d is None else per("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * d * CLK),
           "parked": per("SQ_WAIT_ANY") / wc, "issue_stall": per("SQ_WAIT_INST_ANY") / wc, "issuing": per("SQ_ACTIVE_INST_ANY") / wc,
           "valu": per("SQ_ACTIVE_INST_VALU") / wc, "lds": per("SQ_ACTIVE_INST_LDS") / wc, "lds_stall": per("SQ_WAIT_INST_LDS") / wc,
           "bank_conflict": per("SQ_LDS_BANK_CONFLICT") / wc,
           "insts": {x: per("SQ_INSTS_" + x) for x in ("VALU", "LDS", "VMEM_RD", "VMEM_WR", "SALU")},
           "fetch_bytes": fetch, "write_bytes": write}
    rows.append(row)
    js["kernels"]["%s|%d" % k] = {"calls_per_step": row["calls_per_step"], "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                                  "bytes_per_launch": fetch + write, "avg_us": row["avg_us"],
                                  "mfma_util": None if row["mfma_util"] is None else round(row["mfma_util"], 4)}
rows.sort(key=lambda r: -(r["calls_per_step"] * (r["avg_us"] or 0)))
pct = lambda v: "  -  " if v != v else "%4.1f%%" % (100 * v)
lines = ["# %s step, batch 256: per-kernel counters of the final code of the round (scripts/pmc_step_summary.py %s %s %s); kernels by time per step" % (mode, tag, pref, mode),
         "# mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average duration x 2.4 GHz); parked / stall / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over",
         "# SQ_WAVE_CYCLES (disjoint); valu, lds = the issuing share by pipe; lds-stall = SQ_WAIT_INST_LDS (inside stall); MB = memory-side bytes per launch (FETCH x 2, WRITE)",
         "%-64s %9s %5s %8s %6s %7s %6s %7s %6s %6s %9s %7s %9s %9s %8s" % ("kernel", "grid", "n/stp", "avg us", "mfma", "parked", "stall", "issuing", "valu", "lds", "lds-stall",
                                                                            "bankcf", "read MB", "write MB", "TB/s")]
for r in rows:
    if r["calls_per_step"] < 0.5 and (r["avg_us"] or 0) < 20:
        continue
    tb = "" if not r["avg_us"] else "%.2f" % ((r["fetch_bytes"] + r["write_bytes"]) / (r["avg_us"] * 1e-6) / 1e12)
    lines.append("%-64s %9d %5.1f %8s %6s %7s %6s %7s %6s %6s %9s %7s %9.1f %9.1f %8s" % (
        r["kernel"][:64], r["grid_threads"], r["calls_per_step"], "-" if r["avg_us"] is None else "%.1f" % r["avg_us"],
        "  -  " if r["mfma_util"] is None else pct(r["mfma_util"]), pct(r["parked"]), pct(r["issue_stall"]), pct(r["issuing"]), pct(r["valu"]), pct(r["lds"]),
        pct(r["lds_stall"]), pct(r["bank_conflict"]), r["fetch_bytes"] / 1e6, r["write_bytes"] / 1e6, tb))
    i = r["insts"]
    if any(i.values()):
        lines.append("    instructions per launch: VALU %.3g  LDS %.3g  VMEM read %.3g  write %.3g  SALU %.3g" % (i["VALU"], i["LDS"], i["VMEM_RD"], i["VMEM_WR"], i["SALU"]))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_pmc_sq_step_%s.txt" % (pref, mode)), "w").write("\n".join(lines) + "\n")
json.dump(js, open(os.path.join(ROOT, "profiles", "%s_pmc_step_%s.json" % (pref, mode)), "w"), indent=1)
print("\n".join(lines[:40]))
