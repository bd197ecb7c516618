#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r4w2_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4w2_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-roofline > $OUT/r4w2_prof.log 2>&1
f=$(find $OUT/r4w2_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r4w2_step_timeline.txt
rm -rf $OUT/r4w2_prof
grep -n "loc_\|sampler\|c1_\|pw1_\|dwconv_naive\|step span" $OUT/r4w2_step_timeline.txt | head -20
