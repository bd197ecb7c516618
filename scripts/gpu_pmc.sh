#!/bin/bash
# PMC passes (separate runs, counters only) over the depthwise micro-benchmark: HBM fetch / write bytes per launch
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o dw -- python $ROOT/scripts/dw_bench.py > $OUT/pmc_$c.log 2>&1
  echo "$c exit $?"
  ls $OUT/pmc_$c | head
done
