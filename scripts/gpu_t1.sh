#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "streaming_nt" --tb=short -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -n 4 2>&1 | tail -3
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'], d['config']['final_loss'])"
FLAGS_LIST="0" HEADN=9 bash scripts/gpu_t2.sh
python scripts/trace_step.py gpurun_out/trace_flags0.csv | grep -E "gemm_nt_f32_stream" | awk '{printf "%s ", $6}'; echo
