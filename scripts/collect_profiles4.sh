#!/bin/bash
# Copy the judged summaries of a scripts/gpu_round4.sh visit from gpurun_out/ (scratch) into profiles/ (tracked).  usage: collect_profiles4.sh [tag]
T=${1:-r04}
ROOT=$(cd "$(dirname "$0")/.." && pwd); G=$ROOT/gpurun_out; P=$ROOT/profiles
cd $ROOT
for f in bench_bf16s bench_fp32 bench_fp32_no_bn2_fusion bench_fp32_tile_schedule bench_fp32_three_plane_backward bench_fp32_two_plane_forward bench_bf16 bench_iam bench_gru bench_bn2_dw_fusion bench_bn2_dw_stats_fusion bench_step_kernels bench_no_bn_stats_fusion bench_bf16s_again predict; do cp $G/${T}_$f.json $P/r04_$f.json; done
for f in step_timeline_bf16s step_timeline_bn2_dw_fusion step_timeline_bn2_dw_stats_fusion step_timeline_fp32 step_timeline_fp32_tile_schedule dw_f32_bench dw_prologue_bench dw_fwd_stream_bench dw_bwd_stream_bench wres_fwd_depth x2_bench summary; do cp $G/${T}_$f.txt $P/r04_$f.txt; done
cp $G/${T}_kernel_stats_bf16s.csv $P/r04_bench_bf16s_kernel_stats.csv
cp $G/${T}_kernel_stats_bn2_dw_fusion.csv $P/r04_bench_bn2_dw_fusion_kernel_stats.csv
tail -15 $G/${T}_pytest_gpu.log > $P/r04_pytest_gpu.txt
python scripts/pmc_summary.py $T r04
python scripts/pmc_f32_summary.py $T r04
cp $G/${T}_kernel_stats_fp32.csv $P/r04_bench_fp32_kernel_stats.csv
ls $P | grep "^r04_" | wc -l
