#!/bin/bash
# GPU visit: per-file pytest runs (xdist workers => a faulting kernel cannot take the other tests down), then a short bench
mkdir -p gpurun_out
rm -f gpurun_out/summary.txt
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -n 4 --tb=short -p no:cacheprovider > gpurun_out/test_gpu_ops.log 2>&1
echo "test_gpu_ops exit $?" >> gpurun_out/summary.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -q -m gpu -n 3 --tb=short -p no:cacheprovider > gpurun_out/test_gpu_model.log 2>&1
echo "test_gpu_model exit $?" >> gpurun_out/summary.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_first.log 2>&1
echo "bench exit $?" >> gpurun_out/summary.txt
grep -E "passed|failed|error" gpurun_out/test_gpu_ops.log | tail -3
grep -E "passed|failed|error" gpurun_out/test_gpu_model.log | tail -3
tail -2 gpurun_out/bench_first.log
cat gpurun_out/summary.txt
