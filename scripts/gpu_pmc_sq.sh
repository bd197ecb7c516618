#!/bin/bash
# SQ counter pass (counters only) over a micro-benchmark: where the wave cycles go (parked / issue-stalled / issuing)
# usage: bash scripts/gpu_pmc_sq.sh <tag> <python script and args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_$TAG
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_$TAG -o pmc -- python $ROOT/$@ > $OUT/pmc_$TAG.log 2>&1
echo "exit $?"
ls $OUT/pmc_$TAG | head
python3 - <<PY
import csv, collections, glob
f = glob.glob("$OUT/pmc_$TAG/*counter_collection.csv")
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in agg.items():
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print("%-60s waves %8.0f  parked %4.1f%%  issue-stall %4.1f%%  active %4.1f%%  valu %4.1f%%  lds %4.1f%%  bankconf %4.1f%%" % (
            k, c.get("SQ_WAVES", 0), 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
PY
