#!/bin/bash
# Round-3 visit M: producer-wave three-plane GEMM with 8 staging waves.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
GEMM_LIB=libgemm_x3p_256.so timeout 200 python scripts/gemm_x3_trace.py > $OUT/r3m_trace_256.txt 2>&1
GEMM_LIB=libgemm_x3p.so timeout 200 python scripts/gemm_x3_trace.py > $OUT/r3m_trace_512.txt 2>&1
cut -c1-400 $OUT/r3m_trace_256.txt $OUT/r3m_trace_512.txt
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "three_plane or gemm" > $OUT/r3m_pytest_new.log 2>&1
echo "pytest_new exit $?" > $OUT/r3m_summary.txt
tail -5 $OUT/r3m_pytest_new.log
CRNN_FLAGS=256 timeout 300 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3m_bench_fp32_x3.json 2> $OUT/r3m_bench.err
cut -c1-140 $OUT/r3m_bench_fp32_x3.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r3m_prof32
CRNN_FLAGS=256 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3m_prof32 -o bench -- python $ROOT/bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3m_prof32_bench.log 2>&1
f=$(find $OUT/r3m_prof32 -name "*kernel_trace.csv" | head -1)
python $ROOT/scripts/trace_step.py $f > $OUT/r3m_step_timeline_fp32.txt
find $OUT -name "*kernel_trace.csv" -size +30M -delete
grep "gemm_x3p\|step span" $OUT/r3m_step_timeline_fp32.txt | head -50
grep -v amdgpu $OUT/r3m_bench.err | tail -5
cat $OUT/r3m_summary.txt
