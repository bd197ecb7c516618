#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "persistent_lstm" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py -q -m gpu -x --tb=short -p no:cacheprovider -n 4 2>&1 | tail -3
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do echo -n "bf16s "; $B 2>/dev/null | cut -c60-170; done
