#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 1500 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/r2n_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?"
grep -E "passed|failed|error" $OUT/r2n_pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" $OUT/r2n_pytest_gpu.log | head
( time timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/r2n_bench.json 2> $OUT/r2n_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2n_bench.json"))
for k in ("value", "ms_per_step", "parity_mode", "bs64", "cpu_baseline"):
    print(k, d.get(k))
print("lstm", {k: d["lstm_roofline"][k] for k in ("achieved", "frac", "ms_per_train_step", "us_per_step")})
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "gemm", d["gemm_roofline"]["frac"], d["gemm_roofline"]["hbm_frac"])
PY
