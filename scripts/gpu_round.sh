#!/bin/bash
# One GPU visit that produces everything profiles/ records for a round: parity tests, the bench line in the three
# precisions, the rocprofv3 kernel statistics of the default bench command, HBM PMC passes over the depthwise
# micro-benchmark (bf16 and fp32 storage) and the predict-path benchmark.   usage: bash scripts/gpu_round.sh <tag>
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 1200 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/${TAG}_summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --precision bf16 --no-cpu-baseline > $OUT/${TAG}_bench_bf16.json 2>> $OUT/${TAG}_bench.err
timeout 600 python scripts/predict_bench.py > $OUT/${TAG}_predict.json 2> $OUT/${TAG}_predict.err
echo "predict exit $?" >> $OUT/${TAG}_summary.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
for mode in bf16 fp32; do
  arg=""; [ $mode = bf16 ] && arg="--bf16"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${TAG}_pmc_${mode}_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${mode}_$c -o dw -- python $ROOT/scripts/dw_bench.py $arg > $OUT/${TAG}_pmc_${mode}_$c.log 2>&1
    echo "pmc $mode $c exit $?" >> $OUT/${TAG}_summary.txt
  done
done
find $OUT -name "*kernel_trace.csv" -size +30M -delete
cd $ROOT
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.log | tail -3
for f in bench bench_fp32 bench_bf16 predict; do cut -c1-700 $OUT/${TAG}_$f.json; echo; done
cat $OUT/${TAG}_summary.txt
