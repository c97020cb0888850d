#!/bin/bash
# The round-3 heap corruption only ever showed inside pytest.  N runs of the multi-stream / captured-window tests in fresh
# pytest processes WITHOUT the harness's collect-and-synchronize (TEF_TEST_NO_COLLECT=1); counts the aborts.
#   tools/pytest_teardown_probe.sh N OUTDIR [extra env assignments, e.g. TEF_TWO_STREAMS=0]
N=$1; OUT=$2; shift 2
mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ok=0; bad=0
for i in $(seq 1 $N); do
  env TEF_TEST_NO_COLLECT=1 "$@" timeout 600 python -m pytest $ROOT/tests/test_train_gpu.py -m gpu -q -x \
      -k "multi_stream or two_stream or graph_replay or cut_short" > $OUT/run_$i.out 2> $OUT/run_$i.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); rm -f $OUT/run_$i.out $OUT/run_$i.err; else bad=$((bad+1)); echo "run $i rc=$rc" >> $OUT/summary.txt; tail -3 $OUT/run_$i.out >> $OUT/summary.txt; tail -8 $OUT/run_$i.err >> $OUT/summary.txt; fi
done
echo "env: $@  ok=$ok bad=$bad" | tee -a $OUT/summary.txt
