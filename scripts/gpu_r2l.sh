#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "streaming_dwconv or dwconv" --tb=short -p no:cacheprovider 2>&1 | tail -8
echo "--- tile kernel"; timeout 120 python scripts/dw_bench.py --bf16 --tile 2>&1 | grep -v amdgpu | tail -20
echo "--- streaming kernel"; timeout 120 python scripts/dw_bench.py --bf16 2>&1 | grep -v amdgpu | tail -20
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['config']['final_loss'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
