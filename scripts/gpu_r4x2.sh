#!/bin/bash
# two-plane backward GEMMs of the parity mode: tests, step A/B
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "plane or parity_mode or fp32" > $OUT/r4x2_pytest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/r4x2_pytest.log | cut -c1-250
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --no-roofline --precision fp32"
for f in 0 65536 196608 0 65536; do
  echo -n "CRNN_FLAGS=$f: "; CRNN_FLAGS=$f $B 2>/dev/null | cut -c1-140
done
