#!/bin/bash
# Round-3 visit B: persistent GRU, give-up test, cache-policy experiment of the exchange.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
timeout 600 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "gru or persistent" > $OUT/r3b_pytest_rnn.log 2>&1
echo "pytest_rnn exit $?" > $OUT/r3b_summary.txt
tail -25 $OUT/r3b_pytest_rnn.log
timeout 900 python -m pytest tests -q -m gpu -n 4 --tb=short -p no:cacheprovider -k "not (gru or persistent)" > $OUT/r3b_pytest_rest.log 2>&1
echo "pytest_rest exit $?" >> $OUT/r3b_summary.txt
tail -8 $OUT/r3b_pytest_rest.log
timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3b_bench_gru.json 2> $OUT/r3b_bench.err
CRNN_FLAGS=1 timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3b_bench_gru_step.json 2>> $OUT/r3b_bench.err
CRNN_FLAGS=64 timeout 300 python bench.py --steps 20 --warmup 5 --gru --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3b_bench_gru_linear.json 2>> $OUT/r3b_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > $OUT/r3b_bench_lstm.json 2>> $OUT/r3b_bench.err
for f in gru gru_step gru_linear lstm; do cut -c1-140 $OUT/r3b_bench_$f.json; echo; done
for m in 1 2 3; do
  CRNN_RNN_LIB=$ROOT/scripts/_trace/librnnp_pol$m.so timeout 120 python scripts/lstm_bench.py --pol-only > $OUT/r3b_lstm_bench_pol$m.json 2> $OUT/r3b_lstm_bench_pol$m.err
  tail -2 $OUT/r3b_lstm_bench_pol$m.err | cut -c1-300
  cut -c1-600 $OUT/r3b_lstm_bench_pol$m.json; echo
done
grep -v amdgpu $OUT/r3b_bench.err | tail -5
cat $OUT/r3b_summary.txt
