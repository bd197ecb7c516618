#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; summaries land in gpurun_out/prof_*
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof exit $?"
find $OUT/prof -name "*stats*" | head
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
head -40 "$f"
tail -2 $OUT/prof_bench.log | cut -c1-300
# keep the merge small: drop the big per-dispatch trace, keep stats
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
