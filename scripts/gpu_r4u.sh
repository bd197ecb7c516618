#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 100 python scripts/dbs_bench.py 2>/dev/null | grep -v amdgpu | cut -c1-230 | head -3
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "fused_depthwise or prologue or fp32_row" 2>&1 | tail -2
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2; do echo -n "bf16s "; $B 2>/dev/null | cut -c60-170; echo -n "fp32  "; $B --precision fp32 2>/dev/null | cut -c60-170; done
