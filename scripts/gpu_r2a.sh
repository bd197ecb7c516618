#!/bin/bash
# round-2 visit A: persistent LSTM parity + micro-benchmark, data-parallel tests
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "persistent" --tb=short -p no:cacheprovider > $OUT/r2a_persist.log 2>&1
echo "persist tests exit $?" > $OUT/r2a_summary.txt
tail -15 $OUT/r2a_persist.log
timeout 300 python scripts/lstm_bench.py > $OUT/r2a_lstm_bench.json 2> $OUT/r2a_lstm_bench.err
echo "lstm bench exit $?" >> $OUT/r2a_summary.txt
cat $OUT/r2a_lstm_bench.json; tail -5 $OUT/r2a_lstm_bench.err
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -x --tb=short -p no:cacheprovider > $OUT/r2a_cli.log 2>&1
echo "cli tests exit $?" >> $OUT/r2a_summary.txt
tail -15 $OUT/r2a_cli.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2a_bench.json 2> $OUT/r2a_bench.err
echo "bench exit $?" >> $OUT/r2a_summary.txt
cut -c1-600 $OUT/r2a_bench.json; tail -3 $OUT/r2a_bench.err
CRNN_FLAGS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/r2a_bench_step.json 2>> $OUT/r2a_bench.err
cut -c1-300 $OUT/r2a_bench_step.json
cat $OUT/r2a_summary.txt
