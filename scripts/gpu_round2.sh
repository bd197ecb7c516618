#!/bin/bash
# One GPU visit that produces what profiles/ records for round 2.   usage: bash scripts/gpu_round2.sh [tag]
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/${TAG}_summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_bf16s.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision bf16 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_bf16.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --imgh 200 --max-len 21 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_iam.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_gru.json 2>> $OUT/${TAG}_bench.err
CRNN_FLAGS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_step_kernels.json 2>> $OUT/${TAG}_bench.err
timeout 600 python scripts/predict_bench.py > $OUT/${TAG}_predict.json 2> $OUT/${TAG}_predict.err
echo "predict exit $?" >> $OUT/${TAG}_summary.txt
timeout 200 python scripts/lstm_bench.py > $OUT/${TAG}_lstm_bench.json 2>/dev/null
timeout 100 python scripts/lstm_trace.py > $OUT/${TAG}_lstm_trace.json 2>/dev/null
UW=2 timeout 100 python scripts/lstm_trace.py >> $OUT/${TAG}_lstm_trace.json 2>/dev/null
timeout 200 python scripts/gemm_nt_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_gemm_nt_bench.txt
timeout 200 python scripts/gemm_ablate.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_gemm_ablate.txt
timeout 200 python scripts/rnn_gemm_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_rnn_gemm_bench.txt
timeout 100 python scripts/fused_bench.py --product-only 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_bwd_fused_bench.txt
timeout 100 python scripts/dws_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_fwd_stream_bench.txt
timeout 100 python scripts/dbs_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_bwd_stream_bench.txt
[ -f scripts/_trace/libwres_exp0.so ] && timeout 200 python scripts/wres_fwd_ablate.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_wres_fwd_ablate.txt
[ -f scripts/_trace/libingest2.so ] && timeout 100 python scripts/experiments/ingest_probe2.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_ingest_probe.txt
[ -f scripts/_trace/libingest.so ] && timeout 100 python scripts/experiments/ingest_probe.py 2>/dev/null | grep -v amdgpu >> $OUT/${TAG}_ingest_probe.txt
[ -f scripts/_trace/libmfma.so ] && timeout 100 python scripts/experiments/mfma_probe.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_mfma_probe.txt
timeout 200 python scripts/host_overhead.py 2>/dev/null | tail -1 > $OUT/${TAG}_host_overhead.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
f=$(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline.txt
for mode in bf16 fp32; do
  arg=""; [ $mode = bf16 ] && arg="--bf16"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${TAG}_pmc_${mode}_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${mode}_$c -o dw -- python $ROOT/scripts/dw_bench.py $arg > $OUT/${TAG}_pmc_${mode}_$c.log 2>&1
    echo "pmc $mode $c exit $?" >> $OUT/${TAG}_summary.txt
  done
done
cd $ROOT
bash scripts/gpu_pmc_sq.sh ${TAG}_sq_dw scripts/dw_bench.py --bf16 2>&1 | grep -E "dwconv|dw_fwd_stream" > $OUT/${TAG}_pmc_sq_dwconv.txt
bash scripts/gpu_pmc_any.sh ${TAG}_step "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" -- bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | grep -A1 -E "gemm_bf16_kernel|gemm_nt_kernel|gemm_wres|dw_bwd_fused|dw_bwd_stream|dw_fwd_stream|pw_wgrad_stream|lstm_.*persist|dwconv_tile|bn_bwd" > $OUT/${TAG}_pmc_sq_step.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep -E "passed|failed|error" $OUT/${TAG}_pytest_gpu.log | tail -3
for f in bench_bf16s bench_fp32 bench_bf16 bench_iam bench_gru bench_step_kernels predict; do cut -c1-400 $OUT/${TAG}_$f.json; echo; done
cat $OUT/${TAG}_summary.txt
