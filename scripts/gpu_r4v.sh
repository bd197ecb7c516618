#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "localisation" 2>&1 | tail -12
timeout 400 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "localisation or small or config1 or nostn or odd or iam" 2>&1 | tail -8
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do echo -n "fused  "; $B 2>/dev/null | cut -c60-170; echo -n "kernels "; CRNN_FLAGS=8192 $B 2>/dev/null | cut -c60-170; done
echo -n "fp32 fused   "; $B --precision fp32 2>/dev/null | cut -c60-170; echo -n "fp32 kernels "; CRNN_FLAGS=8192 $B --precision fp32 2>/dev/null | cut -c60-170
