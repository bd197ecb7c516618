#!/bin/bash
# Round-3 visit P: predict path with the pooled blocks' BatchNorm + ReLU6 + MaxPooling in the pointwise GEMM's epilogue.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "pools_in_its_epilogue or window_major" > $OUT/r3p_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/r3p_summary.txt
tail -25 $OUT/r3p_pytest_new.log
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/r3p_pytest_gpu.log 2>&1
echo "pytest_all exit $?" >> $OUT/r3p_summary.txt
tail -25 $OUT/r3p_pytest_gpu.log
timeout 300 python scripts/predict_bench.py > $OUT/r3p_predict.json 2> $OUT/r3p_predict.err
cut -c1-700 $OUT/r3p_predict.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r3p_profp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/r3p_profp -o p -- python $ROOT/scripts/predict_bench.py --iters 6 --cpu-sample 2 > $OUT/r3p_profp.log 2>&1
f=$(find $OUT/r3p_profp -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_predict.py $f > $OUT/r3p_predict_timeline.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep -A12 "iteration span" $OUT/r3p_predict_timeline.txt
cat $OUT/r3p_summary.txt
