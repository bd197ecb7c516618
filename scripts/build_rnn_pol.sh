#!/bin/bash
# Cache-policy variants of the persistent recurrences' exchange for scripts/lstm_bench.py (CRNN_RNN_LIB=...): CRNN_RNN_POL 1 = plain
# stores + sc1 loads, 2 = sc0 / sc0, 3 = sc0|sc1 both (system scope).  Only meaningful with the XCD-local cluster map.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/scripts/_trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include"
cd $ROOT/crnn-ocr-lite_amd/csrc
for m in 1 2 3; do hipcc $F -DCRNN_RNN_POL=$m rnn_persist.hip -o $O/librnnp_pol$m.so & done
wait
ls -la $O | grep rnnp
