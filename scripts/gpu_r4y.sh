#!/bin/bash
# fp32 row-stream forms: tests, micro-benchmark, parity-mode step A/B (flags 0 | 1024 | 3072 | 32 tile | 16 three-kernel backward)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "fp32_row_stream or prologue or fused_depthwise" 2>&1 | tail -15 > gpurun_out/r4y_pytest_ops.txt
tail -4 gpurun_out/r4y_pytest_ops.txt
timeout 400 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -s -k "fp32_row_stream or small or config1 or nostn or two_pass" 2>&1 | grep -v "^\[" | tail -12 > gpurun_out/r4y_pytest_model.txt
tail -6 gpurun_out/r4y_pytest_model.txt
timeout 200 python scripts/dws_f32_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4y_dws_f32_bench.txt
B="timeout 300 python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2; do
  for f in 0 1024 3072 32 16; do echo -n "flags $f  "; CRNN_FLAGS=$f $B 2>/dev/null | cut -c60-170; done
done
