#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
cp crnn-ocr-lite_amd/libcrnn_mi355x.so /tmp/lib_new.so
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2 3; do
  cp scripts/_trace/libcrnn_deep0.so crnn-ocr-lite_amd/libcrnn_mi355x.so; echo -n "deep0 "; $B 2>/dev/null | cut -c60-160
  cp /tmp/lib_new.so crnn-ocr-lite_amd/libcrnn_mi355x.so; echo -n "deep1 "; $B 2>/dev/null | cut -c60-160
done
