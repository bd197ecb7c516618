#!/bin/bash
# fp32 step timelines: default schedule vs BatchNorm-2 fusion (+ statistics)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {
  n=$1; shift
  rm -rf $OUT/r4z_prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4z_prof_$n -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-roofline --precision fp32 > $OUT/r4z_prof_${n}.log 2>&1
  f=$(find $OUT/r4z_prof_$n -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_step.py $f > $OUT/r4z_step_timeline_$n.txt
  rm -rf $OUT/r4z_prof_$n
}
CRNN_FLAGS=0 prof f0
CRNN_FLAGS=1024 prof f1024
CRNN_FLAGS=3072 prof f3072
grep "step span" $OUT/r4z_step_timeline_*.txt
