#!/bin/bash
# Round-3 visit D: XCD census + plain exchange stores, occupancy probe, traces.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "(gru or persistent) and not lost_cluster" > $OUT/r3d_pytest_rnn.log 2>&1
echo "pytest_rnn exit $?" > $OUT/r3d_summary.txt
tail -12 $OUT/r3d_pytest_rnn.log
timeout 120 python scripts/occupy_probe.py > $OUT/r3d_occupy_probe.txt 2>&1
grep -v amdgpu $OUT/r3d_occupy_probe.txt | tail -12
timeout 200 python scripts/lstm_bench.py > $OUT/r3d_lstm_bench.json 2>/dev/null
XCD=1 UW=2 timeout 100 python scripts/lstm_trace.py > $OUT/r3d_lstm_trace.json 2>/dev/null
XCD=0 UW=2 timeout 100 python scripts/lstm_trace.py >> $OUT/r3d_lstm_trace.json 2>/dev/null
cat $OUT/r3d_lstm_trace.json
for f in "" "--gru"; do
  timeout 300 python bench.py --steps 20 --warmup 5 $f --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3d_bench$f.json 2>> $OUT/r3d_bench.err
  CRNN_FLAGS=64 timeout 300 python bench.py --steps 20 --warmup 5 $f --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3d_bench${f}_linear.json 2>> $OUT/r3d_bench.err
  cut -c1-140 $OUT/r3d_bench$f.json; echo; cut -c1-140 $OUT/r3d_bench${f}_linear.json; echo
done
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "not (gru or persistent)" > $OUT/r3d_pytest_rest.log 2>&1
echo "pytest_rest exit $?" >> $OUT/r3d_summary.txt
tail -5 $OUT/r3d_pytest_rest.log
python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r3d_lstm_bench.json")))
for mode in ("bf16","fp32"):
    for k,v in d.get(mode,{}).items():
        print("  ",mode,k,v.get("fwd_us"),v.get("bwd_us"),"status",v.get("status"),v.get("giveups"))
PY
grep -v amdgpu $OUT/r3d_bench.err | tail -5
cat $OUT/r3d_summary.txt
