#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "block1" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^\[" | tail -12
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do echo -n "folded  "; $B 2>/dev/null | cut -c60-170; echo -n "kernels "; CRNN_FLAGS=16384 $B 2>/dev/null | cut -c60-170; done
