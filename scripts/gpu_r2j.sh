#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r2j_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2j_prof -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $OUT/r2j_prof.log 2>&1
f=$(find $OUT/r2j_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r2j_timeline.txt
grep -E "gemm_bf16_kernel<128, true|gemm_nt|splitk" $OUT/r2j_timeline.txt | cut -c1-140
grep -A 12 "step span" $OUT/r2j_timeline.txt
find $OUT/r2j_prof -name "*kernel_trace.csv" -size +30M -delete
