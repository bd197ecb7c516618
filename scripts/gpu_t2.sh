#!/bin/bash
# kernel trace of the default step and of a CRNN_FLAGS variant: per-kernel aggregates of one step each
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
R=$PWD; cd /tmp; export TMPDIR=/tmp
for fl in ${FLAGS_LIST:-0 16}; do
rm -rf /tmp/tr$fl
CRNN_FLAGS=$fl timeout 300 rocprofv3 --kernel-trace -d /tmp/tr$fl -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2>&1
f=$(find /tmp/tr$fl -name '*kernel_trace.csv' | head -1)
echo "== flags=$fl"; python $R/scripts/trace_step.py $f --agg | head -${HEADN:-28}
cp $f $R/gpurun_out/trace_flags$fl.csv
done
