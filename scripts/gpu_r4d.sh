#!/bin/bash
TAG=${1:-r04d}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -s -k "prologue or rng_statistics or two_pass_path" > $OUT/${TAG}_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/${TAG}_summary.txt
grep -E "^FAILED|passed|failed|^E  |world 4" $OUT/${TAG}_pytest_new.log | cut -c1-300 | head -20
[ -f $OUT/dp_world4_failure.log ] && grep -v "Gloo\|amdgpu" $OUT/dp_world4_failure.log | grep "world\|Error" | head
timeout 300 python scripts/dws_pro_bench.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_dws_pro_bench.txt
cat $OUT/${TAG}_dws_pro_bench.txt
for i in 1 2; do
CRNN_FLAGS=1024 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bn2_dw_fusion_$i.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bf16s_$i.json 2>> $OUT/${TAG}_bench.err
done
for f in bench_bn2_dw_fusion_1 bench_bf16s_1 bench_bn2_dw_fusion_2 bench_bf16s_2; do echo -n "$f: "; cut -c60-200 $OUT/${TAG}_$f.json; echo; done
grep -v amdgpu $OUT/${TAG}_bench.err | tail -8
cat $OUT/${TAG}_summary.txt
