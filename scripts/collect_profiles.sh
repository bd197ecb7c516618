#!/bin/bash
# Copy the judged summaries of a scripts/gpu_round3.sh visit from gpurun_out/ (scratch) into profiles/ (tracked).  usage: collect_profiles.sh [tag]
T=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd); G=$ROOT/gpurun_out; P=$ROOT/profiles
cd $ROOT
for f in bench_bf16s bench_fp32 bench_fp32_mfma_gemms bench_bf16 bench_iam bench_gru bench_gru_step_kernels bench_step_kernels bench_linear_clusters \
         bench_no_bn_stats_fusion bench_deferred_sums bench_bf16s_again predict lstm_bench lstm_trace; do cp $G/${T}_$f.json $P/${T}_$f.json; done
for f in step_timeline step_timeline_gru step_timeline_fp32 occupy_probe gemm_x3_bench dw_fwd_stream_bench dw_bwd_stream_bench summary; do cp $G/${T}_$f.txt $P/${T}_$f.txt; done
for f in gemm_x3_trace gemm_x3_bench_variants gemm_x3p_experiments mfma_valu_probe; do [ -f $G/${T}_$f.txt ] && cp $G/${T}_$f.txt $P/${T}_$f.txt; done
cp $(find $G/${T}_prof -name "*kernel_stats.csv" | head -1) $P/${T}_bench_bf16s_kernel_stats.csv
tail -15 $G/${T}_pytest_gpu.log > $P/${T}_pytest_gpu.txt
python scripts/pmc_summary.py $T $T
ls $P | grep "^${T}_" | wc -l
