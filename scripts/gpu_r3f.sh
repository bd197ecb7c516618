#!/bin/bash
# Round-3 visit F: three-plane fp32-accurate GEMMs in the parity mode, lost-cluster test.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -s -k "three_plane or lost_cluster" > $OUT/r3f_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/r3f_summary.txt
grep -v amdgpu $OUT/r3f_pytest_new.log | grep -E "x3 error|lost cluster|passed|failed|Error|assert" | sort | uniq -c | sort -rn | head -40
timeout 300 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3f_bench_fp32_x3.json 2> $OUT/r3f_bench.err
CRNN_FLAGS=256 timeout 300 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3f_bench_fp32_mfma.json 2>> $OUT/r3f_bench.err
for f in fp32_x3 fp32_mfma; do cut -c1-140 $OUT/r3f_bench_$f.json; echo; done
timeout 1200 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "not (three_plane or lost_cluster)" > $OUT/r3f_pytest_rest.log 2>&1
echo "pytest_rest exit $?" >> $OUT/r3f_summary.txt
tail -25 $OUT/r3f_pytest_rest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r3f_prof32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3f_prof32 -o bench -- python $ROOT/bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3f_prof32_bench.log 2>&1
f=$(find $OUT/r3f_prof32 -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r3f_step_timeline_fp32.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep "gemm_bf16_kernel\|step span" $OUT/r3f_step_timeline_fp32.txt | head -60
grep -v amdgpu $OUT/r3f_bench.err | tail -5
cat $OUT/r3f_summary.txt
