#!/bin/bash
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 300 python scripts/dws_pro_bench.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_dws_pro_bench.txt
cat $OUT/${TAG}_dws_pro_bench.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py -q -m gpu --tb=short -p no:cacheprovider -s -k "small_model_no_dropout or iam_shape_full or config1_shape_full or world4 or two_ranks or two_pass_path or staged or stn_callable" > $OUT/${TAG}_pytest_sel.log 2>&1
echo "pytest_sel exit $?" > $OUT/${TAG}_summary.txt
grep -E "discontinuous|worst gradient|^FAILED|passed|failed|^E  " $OUT/${TAG}_pytest_sel.log | cut -c1-400 | head -40
[ -f $OUT/dp_world4_failure.log ] && grep -v "Gloo\|amdgpu" $OUT/dp_world4_failure.log | tail -30
