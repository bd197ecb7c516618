#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or pwconv or lstm or gru" --tb=short -p no:cacheprovider -n 4 2>&1 | tail -4
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config']['final_loss'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r2k_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r2k_prof -o bench -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $OUT/r2k_prof.log 2>&1
f=$(find $OUT/r2k_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r2k_timeline.txt
grep -E "gemm_bf16_kernel<128, true" $OUT/r2k_timeline.txt | cut -c1-140
grep -A 8 "step span" $OUT/r2k_timeline.txt
find $OUT/r2k_prof -name "*kernel_trace.csv" -size +30M -delete
