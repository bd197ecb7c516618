#!/bin/bash
# One GPU visit that produces what profiles/ records for round 4.   usage: bash scripts/gpu_round4.sh [tag]
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
rm -f $OUT/rccl_skip_reason.txt $OUT/dp_world4_failure.log
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/${TAG}_summary.txt
[ -f $OUT/rccl_skip_reason.txt ] && cat $OUT/rccl_skip_reason.txt >> $OUT/${TAG}_summary.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_bf16s.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity"
$B --precision fp32 > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=4096 $B --precision fp32 --no-roofline > $OUT/${TAG}_bench_fp32_no_bn2_fusion.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=32 $B --precision fp32 --no-roofline > $OUT/${TAG}_bench_fp32_tile_schedule.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=65536 $B --precision fp32 --no-roofline > $OUT/${TAG}_bench_fp32_three_plane_backward.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=131072 $B --precision fp32 --no-roofline > $OUT/${TAG}_bench_fp32_two_plane_forward.json 2>> $OUT/${TAG}_bench.err
$B --precision bf16 > $OUT/${TAG}_bench_bf16.json 2>> $OUT/${TAG}_bench.err
$B --imgh 200 --max-len 21 > $OUT/${TAG}_bench_iam.json 2>> $OUT/${TAG}_bench.err
$B --gru > $OUT/${TAG}_bench_gru.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=1024 $B > $OUT/${TAG}_bench_bn2_dw_fusion.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=3072 $B --no-roofline > $OUT/${TAG}_bench_bn2_dw_stats_fusion.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=1 $B --no-roofline > $OUT/${TAG}_bench_step_kernels.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=128 $B --no-roofline > $OUT/${TAG}_bench_no_bn_stats_fusion.json 2>> $OUT/${TAG}_bench.err
$B --no-roofline > $OUT/${TAG}_bench_bf16s_again.json 2>> $OUT/${TAG}_bench.err
timeout 300 python scripts/predict_bench.py > $OUT/${TAG}_predict.json 2> $OUT/${TAG}_predict.err
timeout 200 python scripts/dws_pro_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_prologue_bench.txt
timeout 200 python scripts/dws_f32_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_f32_bench.txt
timeout 100 python scripts/dws_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_fwd_stream_bench.txt
timeout 100 python scripts/dbs_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_bwd_stream_bench.txt
timeout 200 python scripts/wres_fwd_ablate.py 0 1 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_wres_fwd_depth.txt
timeout 100 python scripts/x2_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_x2_bench.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  n=$1; shift
  rm -rf $OUT/${TAG}_prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$n -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity "$@" > $OUT/${TAG}_prof_${n}.log 2>&1
  echo "rocprof $n exit $?" >> $OUT/${TAG}_summary.txt
  f=$(find $OUT/${TAG}_prof_$n -name "*kernel_trace.csv" | head -1)
  python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline_$n.txt
  cp $(find $OUT/${TAG}_prof_$n -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_$n.csv
}
prof bf16s
CRNN_FLAGS=1024 prof bn2_dw_fusion --no-roofline
CRNN_FLAGS=3072 prof bn2_dw_stats_fusion --no-roofline
prof fp32 --precision fp32 --no-roofline
CRNN_FLAGS=32 prof fp32_tile_schedule --precision fp32 --no-roofline
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_bf16_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_bf16_$c -o dw -- python $ROOT/scripts/dw_bench.py --bf16 > $OUT/${TAG}_pmc_bf16_$c.log 2>&1
  echo "pmc bf16 $c exit $?" >> $OUT/${TAG}_summary.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_f32_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_f32_$c -o dw -- python $ROOT/scripts/dw_f32_pmc.py > $OUT/${TAG}_pmc_f32_$c.log 2>&1
  echo "pmc f32 $c exit $?" >> $OUT/${TAG}_summary.txt
done
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.log | tail -3
for f in bench_bf16s bench_fp32 bench_fp32_no_bn2_fusion bench_fp32_tile_schedule bench_fp32_three_plane_backward bench_fp32_two_plane_forward bench_bf16 bench_iam bench_gru bench_bn2_dw_fusion bench_bn2_dw_stats_fusion bench_step_kernels bench_no_bn_stats_fusion bench_bf16s_again predict; do echo -n "$f: "; cut -c1-170 $OUT/${TAG}_$f.json; echo; done
grep -v amdgpu $OUT/${TAG}_bench.err | tail -5
grep "step span" $OUT/${TAG}_step_timeline_*.txt
cat $OUT/${TAG}_summary.txt
