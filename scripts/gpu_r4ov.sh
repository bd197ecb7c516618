#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
B="timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity"
for i in 1 2; do
  echo -n "default      "; $B 2>/dev/null | cut -c60-170
  echo -n "rnn overlap  "; CRNN_RNN_OVERLAP=1 $B 2>/dev/null | cut -c60-170
  echo -n "conv overlap "; CRNN_CONV_OVERLAP=1 $B 2>/dev/null | cut -c60-170
  echo -n "both         "; CRNN_RNN_OVERLAP=1 CRNN_CONV_OVERLAP=1 $B 2>/dev/null | cut -c60-170
done
