#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/r2d_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/r2d_summary.txt
grep -E "passed|failed|error" $OUT/r2d_pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/r2d_pytest_gpu.log | head
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "batch256_against or batch1024 or iam_shape_full" -p no:cacheprovider 2>&1 | grep -E "batch256|passed|failed|iam" | head
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r2d_bench.json 2> $OUT/r2d_bench.err ) 2>&1 | grep real
echo "bench exit $?" >> $OUT/r2d_summary.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2d_bench.json"))
for k in ("value", "ms_per_step", "parity_mode", "bs64", "cpu_baseline", "lstm_roofline"):
    print(k, d.get(k))
print("roofline", d["roofline"]["frac"], "gemm", d["gemm_roofline"]["frac"], d["gemm_roofline"]["hbm_frac"])
PY
tail -3 $OUT/r2d_bench.err
cat $OUT/r2d_summary.txt
