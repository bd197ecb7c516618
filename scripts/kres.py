#!/usr/bin/env python3
"""kernel resource summary: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python scripts/kres.py [filter]"""
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
    if "error" in line: print(line.rstrip())
for k, v in sorted(rows.items()):
    if flt in k:
        print("%-75s vgpr %3d agpr %3d spill %3d scratch %4d lds %6d occ %d" % (k[:75], v.get("VGPRs", -1), v.get("AGPRs", 0), v.get("VGPRs Spill", 0),
              v.get("ScratchSize [bytes/lane]", 0), v.get("LDS Size [bytes/block]", 0), v.get("Occupancy [waves/SIMD]", 0)))
