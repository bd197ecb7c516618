#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "fused_depthwise" --tb=short -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|Error" | head -30
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "producer_fused" --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 100 python scripts/fused_bench.py --product-only
for fl in 0 16; do
CRNN_FLAGS=$fl timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags=$fl', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
