#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or pwconv" --tb=short -p no:cacheprovider -n 4 2>&1 | tail -4
for xs in 0 1; do
CRNN_XSPLIT=$xs timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xsplit=$xs', d['ms_per_step'], d['value'], d['config']['final_loss'])"
done
CRNN_FLAGS=2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile-kernel dgrad', d['ms_per_step'], d['value'], d['config']['final_loss'])"
