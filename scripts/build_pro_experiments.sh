#!/bin/bash
# Ablation builds of the prologue-form row-stream kernels for scripts/dws_pro_bench.py (hipcc cross-compiles gfx950 without a GPU).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/scripts/_trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include"
cd $ROOT/crnn-ocr-lite_amd/csrc
rm -f $O/libdwsp_* $O/libdbsp_*
hipcc $F -DCRNN_DWS_EXP=32 dwconv_stream.hip conv.hip -o $O/libdwsp_noxf.so &        # transform waves keep only their barriers
hipcc $F -DCRNN_DWS_EXP=4 dwconv_stream.hip conv.hip -o $O/libdwsp_nofma.so &        # compute waves without their fmas
hipcc $F -DCRNN_DWS_EXP=2 dwconv_stream.hip conv.hip -o $O/libdwsp_nostore.so &
hipcc $F -DCRNN_DWS_EXP=36 dwconv_stream.hip conv.hip -o $O/libdwsp_noxf_nofma.so &
wait
ls $O | grep -E "dwsp|dbsp"
