#!/bin/bash
# fp32 default schedule (BatchNorm-2 fusions in the fp32 row-stream kernels, statistics in the DX waves): full tests, micro-benchmark, step A/B
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r4y2_pytest.txt
tail -4 gpurun_out/r4y2_pytest.txt
timeout 200 python scripts/dws_f32_bench.py 2>&1 | grep -v amdgpu | grep "bwd\|BN2" | tee gpurun_out/r4y2_dws_f32_bench.txt
B="timeout 300 python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do
  for f in 0 4096 32; do echo -n "flags $f  "; CRNN_FLAGS=$f $B 2>/dev/null | cut -c60-170; done
done
