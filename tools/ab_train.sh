#!/bin/bash
# A/B on one GPU box, training window: default library vs variants (tools/variant.sh), as a hipGraph and eagerly
#   tools/ab_train.sh "VARIANT ..."
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in base $1; do
    if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
    g=$(python bench.py --mode train --graph --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    e=$(python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v graph $g eager $e"
  done
done
