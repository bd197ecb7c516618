#!/bin/bash
# first GPU visit: per-file pytest runs (each in its own process, bounded), then a short bench
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
for f in test_gpu_ops test_gpu_model; do
  timeout 900 python -m pytest tests/$f.py -q -m gpu -x --tb=short -p no:cacheprovider > gpurun_out/$f.log 2>&1
  echo "$f exit $?" >> gpurun_out/summary.txt
  tail -5 gpurun_out/$f.log
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_first.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
tail -3 gpurun_out/bench_first.log
cat gpurun_out/summary.txt
