#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd
for args in "--batch 64 --no-roofline --no-secondary" "--no-roofline" ""; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', '| main', d['ms_per_step'], '| bs64', d.get('bs64',{}).get('ms_per_step'), '| parity', d.get('parity_mode',{}).get('ms_per_step'))"
done
