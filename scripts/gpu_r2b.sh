#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "lstm" --tb=short -p no:cacheprovider > $OUT/r2b_lstm_tests.log 2>&1
echo "lstm tests exit $?" > $OUT/r2b_summary.txt
tail -8 $OUT/r2b_lstm_tests.log
timeout 300 python scripts/lstm_bench.py > $OUT/r2b_lstm_bench.json 2> $OUT/r2b_lstm_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2b_lstm_bench.json"))
for mode in ("bf16", "fp32"):
    for k, v in d[mode].items():
        print(mode, k, v["fwd_us"], v["bwd_us"], v["status"])
PY
for uw in 1 2; do UW=$uw timeout 120 python scripts/lstm_trace.py; done
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -n 4 > $OUT/r2b_model_tests.log 2>&1
echo "model tests exit $?" >> $OUT/r2b_summary.txt
tail -5 $OUT/r2b_model_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/r2b_bench.json 2> $OUT/r2b_bench.err
cut -c1-200 $OUT/r2b_bench.json
cat $OUT/r2b_summary.txt
