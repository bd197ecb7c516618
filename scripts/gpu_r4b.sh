#!/bin/bash
# Round 4, visit B: prologue-kernel op tests, A/B bench against the unfused schedule, a step timeline.
TAG=${1:-r04b}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x --tb=short -p no:cacheprovider -k "prologue or rng_statistics or two_pass_path" > $OUT/${TAG}_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/${TAG}_summary.txt
tail -5 $OUT/${TAG}_pytest_new.log
for i in 1 2; do
CRNN_FLAGS=1024 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bn2_dw_fusion_$i.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bf16s_$i.json 2>> $OUT/${TAG}_bench.err
done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --no-roofline > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
f=$(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline.txt
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +30M -delete
for f in bench_bn2_dw_fusion_1 bench_bf16s_1 bench_bn2_dw_fusion_2 bench_bf16s_2; do echo -n "$f: "; cut -c60-200 $OUT/${TAG}_$f.json; echo; done
grep -v amdgpu $OUT/${TAG}_bench.err | tail -8
grep -E "dw_fwd_stream|dw_bwd_stream" $OUT/${TAG}_step_timeline.txt | cut -c1-120
grep "step span" $OUT/${TAG}_step_timeline.txt
cat $OUT/${TAG}_summary.txt
