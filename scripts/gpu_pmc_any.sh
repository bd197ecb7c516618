#!/bin/bash
# Counter-only PMC passes over a micro-benchmark, one pass per quoted counter group; per-kernel sums of every counter.
# usage: bash scripts/gpu_pmc_any.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <python script and args...>
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf $OUT/pmcany_${TAG}_$i
  timeout 600 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmcany_${TAG}_$i -o pmc -- python $ROOT/$@ > $OUT/pmcany_${TAG}_$i.log 2>&1
  echo "pass $i ($g) exit $?"
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob("$OUT/pmcany_${TAG}_*/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        d = (k, r["Dispatch_Id"])
        if d not in seen: seen.add(d)
    for k, _ in seen: calls[k] = max(calls[k], sum(1 for kk, _ in seen if kk == k))
for k, c in sorted(agg.items(), key=lambda x: -x[1].get("SQ_BUSY_CYCLES", x[1].get("GRBM_GUI_ACTIVE", 0))):
    if "gemm" not in k and "dwconv" not in k and "bn_" not in k and "lstm" not in k and "dw_bwd" not in k: continue
    print(k, "calls", calls[k])
    print("   ", "  ".join("%s=%.4g" % (n, v / max(1, calls[k])) for n, v in sorted(c.items())))
PY
