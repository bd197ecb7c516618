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
# (round 5's channel-range experiment -- bf16 maps as two / four ranges on six-wave workgroups, two per CU: profiles/r05_dw_fwd_split_experiment.txt -- was built
#  from macros that are gone; dws_geom now cuts rows into channel ranges only where a row does not fit nine waves)
for m in 1 2 4 8; do hipcc $F -DCRNN_DBS_EXP=$m dwconv_bwd_stream.hip conv.hip -o $O/libdbs_exp$m.so & done   # no DMA | no stores | no dk fmas | no dx fmas
for d in 2 5; do hipcc $F -DCRNN_DBS_D=$d dwconv_bwd_stream.hip conv.hip -o $O/libdbs_d$d.so & done
wait
ls $O | grep -E "dws|dbs"
# dense2's one-pass backward (scripts/dense_bench.py): the timing build (per-workgroup stamps) and its ablations
rm -f $O/libdense_*
hipcc $F -DCRNN_DSB_TRACE dense.hip conv.hip -o $O/libdense_trace.so &
for m in 1 2 8 10 24; do hipcc $F -DCRNN_DSB_TRACE -DCRNN_DSB_EXP=$m dense.hip conv.hip -o $O/libdense_trace_exp$m.so & done   # no multiply-adds | no LDS dy reads | no dx store | 2+8 | no stores, no x DMA
wait
ls $O | grep dense
# dense1's forward micro-benchmark (scripts/dense1_bench.py): the stripe-stream file with the tuning knobs live
hipcc $F -DCRNN_EXPERIMENT_HOOKS gemm_wgrad.hip -o $O/libcrnn_hooks.so
