#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "persistent_nt" --tb=short -p no:cacheprovider 2>&1 | tail -12
timeout 300 python scripts/gemm_nt_bench.py 2>&1 | grep -v amdgpu
