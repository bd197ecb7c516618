#!/bin/bash
# ONE parameterised GPU visit (round 5; replaces the gpu_r3*.sh / gpu_r4*.sh one-offs).  usage, from the repo root on the GPU box:
#   bash scripts/gpu_visit.sh TAG STEP [STEP ...]
# steps (each writes gpurun_out/TAG_<step>.*; a step's failure does not stop the others):
#   tests:<pytest -k expr>   the GPU tests selected by the expression ("tests:" alone = the whole -m gpu suite)
#   bench[:args]             bench.py (default line) with extra args
#   dws | dbs | probe        depthwise forward / backward micro-benchmarks (product library + scripts/_trace variants), access-pattern probe
#   prof:name[:args]         rocprofv3 --kernel-trace --stats of bench.py (short run) -> step timeline + kernel stats
#   pmc:name:COUNTERS[:args] rocprofv3 --pmc pass (counters comma separated, no trace domains beside --kernel-trace) of bench.py
#   pmcdw                    FETCH_SIZE / WRITE_SIZE passes over scripts/dw_bench.py --bf16 (cold depthwise forward; fold with scripts/pmc_summary.py)
#   py:script[:args]         python scripts/<script> args
#   ab:libs:args             bench.py headline with each of the comma-separated libraries (product | scripts/_trace/libcrnn_<name>.so)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$ROOT:$ROOT/crnn-ocr-lite_amd:$ROOT/tests
cd $ROOT
SUM=$OUT/${TAG}_summary.txt
: > $SUM
BENCH_SHORT="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity"
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  t0=$(date +%s)
  case $kind in
    tests)
      n=$(echo "$rest" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40)
      if [ -z "$rest" ]; then timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/${TAG}_pytest_gpu.log 2>&1
      else timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s -k "$rest" > $OUT/${TAG}_pytest_$n.log 2>&1; fi
      rc=$?; f=$OUT/${TAG}_pytest_${n:-gpu}.log; [ -z "$rest" ] && f=$OUT/${TAG}_pytest_gpu.log
      echo "tests[$rest] exit $rc: $(grep -E ' passed| failed| error' $f | tail -1)" >> $SUM
      [ -f $OUT/rccl_skip_reason.txt ] && cat $OUT/rccl_skip_reason.txt >> $SUM ;;
    bench)
      n=$(echo "$rest" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40)
      timeout 1500 python bench.py $(echo $rest | tr ':' ' ') > $OUT/${TAG}_bench_${n:-default}.json 2> $OUT/${TAG}_bench_${n:-default}.err
      echo "bench[$rest] exit $?: $(cut -c1-150 $OUT/${TAG}_bench_${n:-default}.json)" >> $SUM ;;
    dws) for wd in ${rest:-32}; do timeout 300 python scripts/dws_bench.py 256 $wd 2>/dev/null | grep -v amdgpu; done > $OUT/${TAG}_dw_fwd_stream_bench.txt; echo "dws exit $?" >> $SUM ;;
    dbs) timeout 300 python scripts/dbs_bench.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_dw_bwd_stream_bench.txt; echo "dbs exit $?" >> $SUM ;;
    probe) timeout 300 scripts/probes/pattern_probe > $OUT/${TAG}_pattern_probe.txt 2>&1; echo "probe exit $?" >> $SUM ;;
    py)
      s=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""
      timeout 600 python scripts/$s $(echo $a | tr ':' ' ') 2> $OUT/${TAG}_${s%.py}.err | grep -v amdgpu > $OUT/${TAG}_${s%.py}.txt; echo "py[$rest] exit $?" >> $SUM ;;
    prof)
      n=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/${TAG}_prof_$n &&
        timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$n -o bench -- python $ROOT/bench.py $BENCH_SHORT $(echo $a | tr ':' ' ') > $OUT/${TAG}_prof_${n}.log 2>&1 )
      echo "prof[$rest] exit $?" >> $SUM
      f=$(find $OUT/${TAG}_prof_$n -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python scripts/trace_step.py $f > $OUT/${TAG}_step_timeline_$n.txt
      k=$(find $OUT/${TAG}_prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$k" ] && cp $k $OUT/${TAG}_kernel_stats_$n.csv
      find $OUT/${TAG}_prof_$n -name "*kernel_trace.csv" -size +40M -delete ;;
    pmc)
      n=${rest%%:*}; r2=${rest#*:}; c=${r2%%:*}; a=${r2#*:}; [ "$a" = "$r2" ] && a=""
      ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/${TAG}_pmc_$n &&
        timeout 900 rocprofv3 --pmc $(echo $c | tr ',' ' ') --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$n -o pmc -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-parity --no-roofline $(echo $a | tr ':' ' ') > $OUT/${TAG}_pmc_${n}.log 2>&1 )
      echo "pmc[$rest] exit $?" >> $SUM
      find $OUT/${TAG}_pmc_$n -name "*kernel_trace.csv" -delete ;;
    pmcdw)   # HBM bytes per launch of the depthwise forward in the cold micro-benchmark (scripts/dw_bench.py --bf16): FETCH_SIZE and WRITE_SIZE passes -> scripts/pmc_summary.py TAG rNN
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && rm -rf $OUT/${TAG}_pmc_bf16_$c &&
          timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_bf16_$c -o dw -- python $ROOT/scripts/dw_bench.py --bf16 > $OUT/${TAG}_pmc_bf16_$c.log 2>&1 )
        echo "pmcdw $c exit $?" >> $SUM
        find $OUT/${TAG}_pmc_bf16_$c -name "*kernel_trace.csv" -delete
      done ;;
    ab)   # ab:lib1,lib2:bench args -- whole-library A/B (scripts/gpu_ab_libs.sh; "product" = the library as built, others scripts/_trace/libcrnn_<name>.so)
      l=${rest%%:*}; a=${rest#*:}; [ "$a" = "$rest" ] && a=""
      n=$(echo "$a" | tr -c 'a-zA-Z0-9' '_' | cut -c1-30)
      RF="--no-roofline"; case "$a" in *roofline*) RF=""; a=$(echo "$a" | sed 's/:*roofline//');; esac      # "roofline" among the args: keep the roofline objects (slower)
      BENCH_ARGS="$RF --no-parity $(echo $a | tr ':' ' ')" bash scripts/gpu_ab_libs.sh $(echo $l | tr ',' ' ') > $OUT/${TAG}_ab_$n.txt 2>&1
      echo "ab[$rest]: $(tr '\n' ' ' < $OUT/${TAG}_ab_$n.txt)" >> $SUM ;;
    *) echo "unknown step $step" >> $SUM ;;
  esac
  echo "  ($(( $(date +%s) - t0 )) s)" >> $SUM
done
cat $SUM
