#!/bin/bash
# Round-3 visit E: lost-cluster test alone (diagnostics), fp32 step timeline, split-K reduce A/B.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 300 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s -k "lost_cluster" > $OUT/r3e_pytest_lost.log 2>&1
echo "pytest_lost exit $?" > $OUT/r3e_summary.txt
grep -v amdgpu $OUT/r3e_pytest_lost.log | tail -15
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3e_bench.json 2> $OUT/r3e_bench.err
cut -c1-140 $OUT/r3e_bench.json; echo
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "gemm or dense or model or config1 or step" > $OUT/r3e_pytest_gemm.log 2>&1
echo "pytest_gemm exit $?" >> $OUT/r3e_summary.txt
tail -5 $OUT/r3e_pytest_gemm.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r3e_prof32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3e_prof32 -o bench -- python $ROOT/bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3e_prof32_bench.log 2>&1
echo "rocprof32 exit $?" >> $OUT/r3e_summary.txt
f=$(find $OUT/r3e_prof32 -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r3e_step_timeline_fp32.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
tail -45 $OUT/r3e_step_timeline_fp32.txt
cat $OUT/r3e_summary.txt
