#!/bin/bash
# Variant builds of the three-plane GEMM's plane split for scripts/gemm_x3_bench.py: 0 = three v_cvt_pk_bf16_f32, 1 = the last plane by v_perm_b32 (same bits),
# 2 = middle plane by truncation too (different rounding of the middle plane)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/scripts/_trace; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include"
cd $ROOT/crnn-ocr-lite_amd/csrc
for v in 0 1 2; do hipcc $F -DX3P_SPLIT=$v gemm.hip -o $O/libx3split$v.so & done
wait
ls -la $O | grep x3split
