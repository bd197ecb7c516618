#!/bin/bash
# Round-3 visit A: status word / bench objects / XCD-local cluster experiments.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -x > $OUT/r3a_pytest_gpu.log 2>&1
echo "pytest_gpu exit $?" > $OUT/r3a_summary.txt
tail -15 $OUT/r3a_pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r3a_bench_bf16s.json 2> $OUT/r3a_bench.err
echo "bench exit $?" >> $OUT/r3a_summary.txt
CRNN_FLAGS=64 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3a_bench_linear_clusters.json 2>> $OUT/r3a_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3a_bench_xcd_clusters.json 2>> $OUT/r3a_bench.err
timeout 200 python scripts/lstm_bench.py > $OUT/r3a_lstm_bench.json 2>/dev/null
for m in 1 2 3; do
  CRNN_RNN_LIB=$ROOT/scripts/_trace/librnnp_pol$m.so timeout 200 python scripts/lstm_bench.py --pol-only > $OUT/r3a_lstm_bench_pol$m.json 2>/dev/null
done
cut -c1-600 $OUT/r3a_bench_bf16s.json; echo
cut -c1-300 $OUT/r3a_bench_linear_clusters.json; echo
cut -c1-300 $OUT/r3a_bench_xcd_clusters.json; echo
tail -3 $OUT/r3a_bench.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r3a_lstm_bench*.json"))):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(os.path.basename(f))
    for mode in ("bf16","fp32"):
        for k,v in d.get(mode,{}).items():
            print("  ",mode,k,v.get("fwd_us"),v.get("bwd_us"),"status",v.get("status"),v.get("giveups"))
PY
cat $OUT/r3a_summary.txt
