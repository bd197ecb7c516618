#!/bin/bash
# fp32 row-stream depthwise-stage backward: tests, then parity-mode step A/B (CRNN_FLAGS=16 = the three-kernel sequence)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 500 python -m pytest tests -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r4x_pytest.txt
tail -3 gpurun_out/r4x_pytest.txt
B="timeout 300 python bench.py --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do
  echo -n "fused  "; CRNN_FLAGS=0 $B 2>/dev/null | cut -c60-170
  echo -n "3-kern "; CRNN_FLAGS=16 $B 2>/dev/null | cut -c60-170
done
