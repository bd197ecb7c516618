#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "side_stream or producer_fused or config1" 2>&1 | tail -4
for ov in 1 0; do
CRNN_CONV_OVERLAP=$ov timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$ov', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
