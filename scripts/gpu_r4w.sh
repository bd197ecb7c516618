#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
for i in 1 2; do timeout 200 python scripts/wres_fwd_ablate.py 0 1 2>&1 | grep -v amdgpu; done
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "wres or weights_resident" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity | cut -c60-160; done
