#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py -q -m gpu -x --tb=short -p no:cacheprovider -n 4 2>&1 | tail -3
timeout 600 python scripts/predict_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print({k:p[k] for k in p if 'ms' in k or 'img' in k})"
