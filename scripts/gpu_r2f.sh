#!/bin/bash
# step timelines (rocprofv3 kernel trace) at batch 256 and batch 64
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out
mkdir -p $OUT; export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
for bsz in 256 64; do
  rm -rf $OUT/r2f_prof$bsz
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2f_prof$bsz -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --batch $bsz --no-cpu-baseline --no-roofline --no-secondary > $OUT/r2f_prof${bsz}.log 2>&1
  f=$(find $OUT/r2f_prof$bsz -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_step.py $f > $OUT/r2f_timeline_b$bsz.txt
  tail -32 $OUT/r2f_timeline_b$bsz.txt
  find $OUT/r2f_prof$bsz -name "*kernel_trace.csv" -size +30M -delete
done
