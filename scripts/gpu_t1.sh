#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "folded_batchnorm or weights_resident" --tb=short -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "predict or inference or config1 or batch_1024 or workspace" 2>&1 | tail -3
timeout 600 python scripts/predict_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
p=json.loads(sys.stdin.read()); print({k:p[k] for k in p if 'ms' in k or 'img' in k})"
