#!/bin/bash
# step timeline of the batch-64 step (the batch BASELINE's metric string quotes): where 2.6 ms go
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd /tmp && export TMPDIR=/tmp
for prec in bf16s fp32; do
  rm -rf $OUT/r4bs64_prof_$prec
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4bs64_prof_$prec -o bench -- python $ROOT/bench.py --batch 64 --precision $prec --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-roofline > $OUT/r4bs64_prof_$prec.log 2>&1
  f=$(find $OUT/r4bs64_prof_$prec -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_step.py $f > $OUT/r4bs64_step_timeline_$prec.txt
  grep -v amdgpu $OUT/r4bs64_prof_$prec.log | tail -1 | cut -c1-160
  tail -12 $OUT/r4bs64_step_timeline_$prec.txt
done
find $OUT -name "*kernel_trace.csv" -size +30M -delete
