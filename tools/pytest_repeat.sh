#!/bin/bash
# N fresh pytest processes of one test selection; counts failures / aborts and keeps their output.
#   tools/pytest_repeat.sh N OUTDIR "k-expression" [env assignments ...]
N=$1; OUT=$2; SEL=$3; shift 3
mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ok=0; bad=0
for i in $(seq 1 $N); do
  env "$@" timeout 600 python -m pytest $ROOT/tests/test_train_gpu.py -m gpu -q -x -k "$SEL" > $OUT/run_$i.out 2> $OUT/run_$i.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); rm -f $OUT/run_$i.out $OUT/run_$i.err; else bad=$((bad+1)); echo "run $i rc=$rc" >> $OUT/summary.txt; grep -E "^E  |Error|rc=" $OUT/run_$i.out | head -6 >> $OUT/summary.txt; tail -4 $OUT/run_$i.err >> $OUT/summary.txt; fi
done
echo "sel: $SEL env: $@  ok=$ok bad=$bad" | tee -a $OUT/summary.txt
