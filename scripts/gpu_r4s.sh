#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
mkdir -p gpurun_out
for i in 1 2; do timeout 300 python scripts/gemm_x3_bench.py libx3split0.so libx3split1.so libx3split2.so 2>&1 | grep -v amdgpu | tee -a gpurun_out/r4s_x3split.txt; done
