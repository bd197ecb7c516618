#!/bin/bash
# One GPU visit: parity tests (xdist-isolated), bench, rocprofv3 kernel-trace stats of the same bench command.
# usage: bash scripts/gpu_all.sh [tag]   -> gpurun_out/<tag>_*
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 1200 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/${TAG}_summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
cd $ROOT
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.log | tail -3
cat $OUT/${TAG}_bench.json | cut -c1-1500
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -25 "$f" | cut -c1-200
cat $OUT/${TAG}_summary.txt
