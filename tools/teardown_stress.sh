#!/bin/bash
# N runs of tools/teardown_stress.py in fresh processes; prints the exit codes' histogram and keeps failing runs' stderr.
#   tools/teardown_stress.sh N OUTDIR [args of teardown_stress.py ...]      (ASAN=1: host-ASan library + preloaded runtime)
N=$1; OUT=$2; shift 2
mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ok=0; bad=0
for i in $(seq 1 $N); do
  if [ "$ASAN" = "1" ]; then
    LD_PRELOAD=$($ROOT/tools/build_asan_host.sh --runtime) TEF_HIP_LIB=$ROOT/taming_event_flow_amd/build/asan/libtef_hip_asan.so \
      ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:detect_odr_violation=0 \
      timeout 600 python $ROOT/tools/teardown_stress.py "$@" > $OUT/run_$i.out 2> $OUT/run_$i.err
  else
    timeout 300 python $ROOT/tools/teardown_stress.py "$@" > $OUT/run_$i.out 2> $OUT/run_$i.err
  fi
  rc=$?
  if [ $rc -eq 0 ] && grep -q "clean exit" $OUT/run_$i.out; then ok=$((ok+1)); rm -f $OUT/run_$i.out $OUT/run_$i.err; else bad=$((bad+1)); echo "run $i rc=$rc" >> $OUT/summary.txt; tail -5 $OUT/run_$i.err >> $OUT/summary.txt; fi
done
echo "args: $@  ASAN=$ASAN  ok=$ok bad=$bad" | tee -a $OUT/summary.txt
