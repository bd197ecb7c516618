#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "block1 or small or config1 or nostn or odd or iam" 2>&1 | tail -3
bash scripts/gpu_r4w2.sh
