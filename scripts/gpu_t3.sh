#!/bin/bash
# kernel stats of the predict path (batch 1024)
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/trp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trp -o t --output-format csv -- python $R/scripts/predict_bench.py > /dev/null 2>&1
f=$(find /tmp/trp -name '*kernel_stats.csv' | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-70s calls %6s  total %8.3f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cp $f $R/gpurun_out/predict_kernel_stats.csv
