#!/bin/bash
# Residency probe (round 6): per-kernel time per (head, sample) pair as the loss window's hand-over buffers
# (traj + meta + yr + ar + cyx, ~21 MB per pair) shrink from 680 MB (32 pairs) below the 256 MiB Infinity Cache.
cd "$(dirname "$0")/.."
out=gpurun_out/residency.txt; : > $out
for cfg in "8 4" "4 4" "2 4" "1 4" "8 2" "8 1" "8 4"; do
  set -- $cfg
  timeout -k 10 240 python bench.py --batch $1 --heads $2 --steps 100 --warmup 10 --no-cpu-baseline --no-train-extra 2>gpurun_out/res_$1_$2.err | B=$1 F=$2 python -c "
import json,sys,os
t=sys.stdin.read(); B=int(os.environ['B']); F=int(os.environ['F']); n=B*F
try:
    d=json.loads(t)
    print('B',B,'F',F,'pairs',n,'ms/step',d['ms_per_step'],'us/pair',round(1e3*d['ms_per_step']/n,3),{k:round(1e3*v['ms']/n,3) for k,v in d['kernels'].items()})
except Exception as e:
    print('FAILED',B,F,repr(e),t[-300:])" | tee -a $out
done
