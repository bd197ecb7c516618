#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
nproc; lscpu | grep -E "Model name|Socket|Thread|Core" | head -5
( time timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r2e_bench.json 2> $OUT/r2e_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2e_bench.json"))
    for k in ("value", "ms_per_step", "parity_mode", "bs64", "lstm_roofline"):
        print(k, d.get(k))
    print("roofline", d["roofline"]["frac"], "gemm", d["gemm_roofline"]["frac"], d["gemm_roofline"]["hbm_frac"])
except Exception as e:
    print("bench json:", e)
PY
tail -5 $OUT/r2e_bench.err
for n in 4 16 32 64; do
timeout 100 python - <<PY
import time, os, sys
from oracle import torch_port as TP
t0 = time.time()
s, l = TP.train_step_benchmark(batch=64, threads=$n, steps=1, warmup=0)
print("threads", $n, "sec/step", round(s, 2), "wall", round(time.time() - t0, 1), flush=True)
PY
done
