#!/bin/bash
# Trace build of the persistent LSTM (in-kernel s_memrealtime stamps) for scripts/lstm_trace.py.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/scripts/_trace; mkdir -p $O
cd $ROOT/crnn-ocr-lite_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include -DCRNN_RNN_TRACE rnn_persist.hip -o $O/librnn_trace.so && ls -la $O/librnn_trace.so
