#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r4y3_pytest.txt
tail -8 gpurun_out/r4y3_pytest.txt
