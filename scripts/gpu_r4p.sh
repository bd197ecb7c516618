#!/bin/bash
# weight planes of the parity mode: op + model tests, per-shape timings, step A/B
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "planes or three_plane" > $OUT/r4p_pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/r4p_pytest.log
timeout 200 python scripts/x3_planes_bench.py 2>/dev/null | grep -v amdgpu | tee $OUT/r4p_x3_planes_bench.txt
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity --no-roofline --precision fp32"
for i in 1 2; do
  $B 2>/dev/null | cut -c1-140
  CRNN_FLAGS=32768 $B 2>/dev/null | cut -c1-140
done
