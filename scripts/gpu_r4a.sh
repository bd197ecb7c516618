#!/bin/bash
# Round 4, visit A: new op tests first (prologue kernels, RNG), then the whole GPU suite, the bench line, the A/B against the unfused schedule, one step timeline.
TAG=${1:-r04a}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x --tb=short -p no:cacheprovider -k "prologue or rng_statistics" > $OUT/${TAG}_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/${TAG}_summary.txt
tail -5 $OUT/${TAG}_pytest_new.log
timeout 700 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/${TAG}_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" >> $OUT/${TAG}_summary.txt
grep -E "^FAILED|^ERROR|passed|failed" $OUT/${TAG}_pytest_gpu.log | tail -25
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_bf16s.json 2> $OUT/${TAG}_bench.err
echo "bench exit $?" >> $OUT/${TAG}_summary.txt
CRNN_FLAGS=1024 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bn2_dw_fusion.json 2>> $OUT/${TAG}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-parity > $OUT/${TAG}_bench_bf16s_again.json 2>> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity > $OUT/${TAG}_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/${TAG}_summary.txt
f=$(find $OUT/${TAG}_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/${TAG}_step_timeline.txt
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
cd $ROOT
find $OUT -name "*kernel_trace.csv" -size +30M -delete
for f in bench_bf16s bench_bn2_dw_fusion bench_bf16s_again; do echo -n "$f: "; cut -c1-200 $OUT/${TAG}_$f.json; echo; done
grep -v amdgpu $OUT/${TAG}_bench.err | tail -8
tail -3 $OUT/${TAG}_step_timeline.txt | head -1; grep "step span" $OUT/${TAG}_step_timeline.txt
cat $OUT/${TAG}_summary.txt
