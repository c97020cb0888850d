#!/bin/bash
# training-window timing of the current tree (round 6 glue work): captured and eager window, 2 repetitions each
cd "$(dirname "$0")/.."
for r in 1 2; do
  for m in "--graph" ""; do
    timeout -k 10 300 python bench.py --mode train $m --steps 8 --warmup 2 --no-cpu-baseline 2>>gpurun_out/train_ab.err | M="$m" python -c "
import json,sys,os
t=sys.stdin.read()
try:
    d=json.loads(t); print('train', os.environ['M'] or 'eager', d['ms_per_step'], {k:v for k,v in d.items() if k in ('launches_per_window','kernel_ms_per_window')})
except Exception as e:
    print('FAILED', repr(e), t[-300:])"
  done
done
