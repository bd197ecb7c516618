#!/bin/bash
# Round-3 visit N: three-plane GEMMs (16-k stages, two workgroups per CU) as the parity mode's default; whole GPU suite.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider > $OUT/r3n_pytest_gpu.log 2>&1
echo "pytest exit $?" > $OUT/r3n_summary.txt
tail -15 $OUT/r3n_pytest_gpu.log
timeout 300 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3n_bench_fp32.json 2> $OUT/r3n_bench.err
CRNN_FLAGS=256 timeout 300 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3n_bench_fp32_mfma.json 2>> $OUT/r3n_bench.err
for f in bench_fp32 bench_fp32_mfma; do cut -c1-140 $OUT/r3n_$f.json; echo; done
timeout 200 python scripts/gemm_x3_bench.py 2>&1 | grep -v amdgpu > $OUT/r3n_gemm_x3_bench.txt; cat $OUT/r3n_gemm_x3_bench.txt
grep -v amdgpu $OUT/r3n_bench.err | tail -5
cat $OUT/r3n_summary.txt
