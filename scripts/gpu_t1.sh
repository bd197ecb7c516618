#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "random_shapes" --tb=short -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
