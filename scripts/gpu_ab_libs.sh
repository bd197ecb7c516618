#!/bin/bash
# A/B of whole-library variants on one box: bench.py (headline only) with each scripts/_trace/libcrnn_<name>.so in place of the product library.
# usage: gpu_ab_libs.sh name...      ("product" = the library as built)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd $ROOT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd
cp crnn-ocr-lite_amd/libcrnn_mi355x.so /tmp/libcrnn_product.so
for rep in 1 2; do
for n in "$@"; do
  if [ $n = product ]; then cp /tmp/libcrnn_product.so crnn-ocr-lite_amd/libcrnn_mi355x.so; else cp scripts/_trace/libcrnn_$n.so crnn-ocr-lite_amd/libcrnn_mi355x.so; fi
  echo -n "$n: "; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS:---no-roofline} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('in_step_effective',{}).get('frac'))"
done; done
cp /tmp/libcrnn_product.so crnn-ocr-lite_amd/libcrnn_mi355x.so
