#!/bin/bash
# rocprofv3 kernel stats for one bench configuration: bash scripts/gpu_prof2.sh <tag> <bench args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/${TAG}_prof_bench.log 2>&1
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -size +30M -delete
head -28 $OUT/${TAG}_prof/bench_kernel_stats.csv | cut -c1-150
