#!/bin/bash
# Ablation / depth variants of the two row-stream depthwise kernels for scripts/dws_bench.py and scripts/dbs_bench.py (CPU-side build:
# hipcc cross-compiles gfx950).  Outputs scripts/_trace/libdws_*.so, libdbs_*.so (git-ignored; they travel with the gpurun snapshot).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/scripts/_trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include"
cd $ROOT/crnn-ocr-lite_amd/csrc
rm -f $O/libdws_* $O/libdbs_*
for m in 1 2 4; do hipcc $F -DCRNN_DWS_EXP=$m dwconv_stream.hip -o $O/libdws_exp$m.so & done      # no DMA | no stores | no fmas
for d in 2 7; do hipcc $F -DCRNN_DWS_D=$d dwconv_stream.hip -o $O/libdws_d$d.so & done                # rows in flight
wait
# round 5: bf16 maps as two (four) channel ranges per row on six-wave workgroups, two per CU; ranges 8 workgroup ids apart (same XCD) or adjacent; deeper ring
hipcc $F -DCRNN_DWS_BF16_SPLIT=2 dwconv_stream.hip -o $O/libdws_split2.so &
hipcc $F -DCRNN_DWS_BF16_SPLIT=2 -DCRNN_DWS_XSTRIDE=1 dwconv_stream.hip -o $O/libdws_split2x1.so &
hipcc $F -DCRNN_DWS_BF16_SPLIT=2 -DCRNN_DWS_D5=7 dwconv_stream.hip -o $O/libdws_split2d7.so &
hipcc $F -DCRNN_DWS_BF16_SPLIT=4 dwconv_stream.hip -o $O/libdws_split4.so &
wait
for m in 1 2 4 8; do hipcc $F -DCRNN_DBS_EXP=$m dwconv_bwd_stream.hip conv.hip -o $O/libdbs_exp$m.so & done   # no DMA | no stores | no dk fmas | no dx fmas
for d in 2 5; do hipcc $F -DCRNN_DBS_D=$d dwconv_bwd_stream.hip conv.hip -o $O/libdbs_d$d.so & done
wait
ls $O | grep -E "dws|dbs"
