#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -n 4 2>&1 | tail -4
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2; do echo -n "lstm "; $B 2>/dev/null | cut -c60-170; echo -n "gru  "; $B --gru 2>/dev/null | cut -c60-170; done
