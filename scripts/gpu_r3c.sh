#!/bin/bash
# Round-3 visit C: BatchNorm-backward statistics inside the data-gradient GEMM, give-up test (time-bounded waits), exchange cache policies.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "batchnorm_backward_statistics or lost_cluster or producer_fused or persistent" > $OUT/r3c_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/r3c_summary.txt
tail -30 $OUT/r3c_pytest_new.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3c_bench.json 2> $OUT/r3c_bench.err
CRNN_FLAGS=128 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3c_bench_nofuse.json 2>> $OUT/r3c_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3c_bench2.json 2>> $OUT/r3c_bench.err
for f in bench bench_nofuse bench2; do cut -c1-140 $OUT/r3c_$f.json; echo; done
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "not (batchnorm_backward_statistics or lost_cluster or producer_fused or persistent)" > $OUT/r3c_pytest_rest.log 2>&1
echo "pytest_rest exit $?" >> $OUT/r3c_summary.txt
tail -6 $OUT/r3c_pytest_rest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r3c_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3c_prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3c_prof_bench.log 2>&1
echo "rocprof exit $?" >> $OUT/r3c_summary.txt
f=$(find $OUT/r3c_prof -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r3c_step_timeline.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
tail -60 $OUT/r3c_step_timeline.txt
cd $ROOT
for m in 1 2 3; do
  CRNN_RNN_LIB=$ROOT/scripts/_trace/librnnp_pol$m.so timeout 120 python scripts/lstm_bench.py --pol-only > $OUT/r3c_lstm_bench_pol$m.json 2> $OUT/r3c_lstm_bench_pol$m.err
  grep -v amdgpu $OUT/r3c_lstm_bench_pol$m.err | tail -2 | cut -c1-300
  cut -c1-700 $OUT/r3c_lstm_bench_pol$m.json; echo
done
grep -v amdgpu $OUT/r3c_bench.err | tail -5
cat $OUT/r3c_summary.txt
