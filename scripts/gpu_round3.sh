#!/bin/bash
# One GPU visit that produces what profiles/ records for round 3.   usage: bash scripts/gpu_round3.sh [tag]
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/${TAG}_summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_bf16s.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=256 timeout 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_fp32_mfma_gemms.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision bf16 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_bf16.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --imgh 200 --max-len 21 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_iam.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_gru.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=1 timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_gru_step_kernels.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_step_kernels.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=64 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_linear_clusters.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=128 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_no_bn_stats_fusion.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=512 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_deferred_sums.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_bf16s_again.json 2>> $OUT/${TAG}_bench.err
timeout 300 python scripts/predict_bench.py > $OUT/${TAG}_predict.json 2> $OUT/${TAG}_predict.err
timeout 200 python scripts/lstm_bench.py > $OUT/${TAG}_lstm_bench.json 2>/dev/null
XCD=1 UW=2 timeout 100 python scripts/lstm_trace.py > $OUT/${TAG}_lstm_trace.json 2>/dev/null
XCD=0 UW=2 timeout 100 python scripts/lstm_trace.py >> $OUT/${TAG}_lstm_trace.json 2>/dev/null
timeout 100 python scripts/occupy_probe.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_occupy_probe.txt
timeout 200 python scripts/gemm_x3_bench.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_gemm_x3_bench.txt
timeout 100 python scripts/dws_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_fwd_stream_bench.txt
timeout 100 python scripts/dbs_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_bwd_stream_bench.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
f=$(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline.txt
rm -rf $OUT/${TAG}_prof_gru
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_gru -o bench -- python $ROOT/bench.py --gru --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_prof_gru_bench.log 2>&1
f=$(find $OUT/${TAG}_prof_gru -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f --agg > $OUT/${TAG}_step_timeline_gru.txt
rm -rf $OUT/${TAG}_prof_fp32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_fp32 -o bench -- python $ROOT/bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_prof_fp32_bench.log 2>&1
f=$(find $OUT/${TAG}_prof_fp32 -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline_fp32.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_bf16_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_bf16_$c -o dw -- python $ROOT/scripts/dw_bench.py --bf16 > $OUT/${TAG}_pmc_bf16_$c.log 2>&1
  echo "pmc bf16 $c exit $?" >> $OUT/${TAG}_summary.txt
done
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.log | tail -3
for f in bench_bf16s bench_fp32 bench_fp32_mfma_gemms bench_bf16 bench_iam bench_gru bench_gru_step_kernels bench_step_kernels bench_linear_clusters bench_no_bn_stats_fusion bench_deferred_sums bench_bf16s_again predict; do echo -n "$f: "; cut -c1-170 $OUT/${TAG}_$f.json; echo; done
grep -v amdgpu $OUT/${TAG}_bench.err | tail -5
cat $OUT/${TAG}_summary.txt
