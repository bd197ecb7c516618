#!/bin/bash
# Variant builds of gemm_pres.hip alone -> scripts/_trace/libpres_<name>.so; run with PRES_LIB=... scripts/pres_bench.py.  usage: pres_variants.sh name "-DPRES_..." [name "-D..."]...
cd $(dirname $0)/..
mkdir -p scripts/_trace
while [ $# -ge 2 ]; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -I include crnn-ocr-lite_amd/csrc/gemm_pres.hip -o scripts/_trace/libpres_$1.so &
  shift 2
done
wait
ls -la scripts/_trace/libpres_*.so
