#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -3
for fl in 0; do
CRNN_FLAGS=$fl timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags=$fl', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
FLAGS_LIST="0" HEADN=14 bash scripts/gpu_t2.sh
