#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | grep -v "^\[" | tail -30 | cut -c1-400
