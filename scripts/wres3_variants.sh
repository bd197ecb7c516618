#!/bin/bash
# Timing-experiment builds of gemm_wres3.hip alone (W3_EXP masks; wrong results) -> scripts/_trace/libw3_exp<mask>.so; run with W3_LIB=... scripts/wres3_bench.py
cd $(dirname $0)/..
mkdir -p scripts/_trace
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DW3_EXP=$m -I include crnn-ocr-lite_amd/csrc/gemm_wres3.hip -o scripts/_trace/libw3_exp$m.so &
done
wait
ls -la scripts/_trace/*.so
