#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 300 python scripts/gemm_ablate.py 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/lstm_bench.py > $OUT/r2c_lstm_bench.json 2> $OUT/r2c_lstm_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2c_lstm_bench.json"))
for mode in ("bf16", "fp32"):
    for k, v in d[mode].items():
        print(mode, k, v["fwd_us"], v["bwd_us"], v["status"])
PY
