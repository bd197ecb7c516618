#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 200 python scripts/dw_trace.py 2>&1 | grep -v amdgpu | head -12
echo "--- streaming kernel"; timeout 120 python scripts/dw_bench.py --bf16 2>&1 | grep -v amdgpu | grep -v wgrad | tail -14
