#!/bin/bash
# A/B on one GPU box, whole training window as one hipGraph: default library vs variants (tools/build_variant.sh)
#   tools/ab_train.sh VARIANT [VARIANT ...]
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
    timeout 200 python bench.py --mode train --graph --steps 8 --warmup 2 2>/dev/null | V=$v python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['V'], d['ms_per_step'], d['loss'])"
  done
done
