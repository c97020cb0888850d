#!/bin/bash
# A/B on one GPU box: default library vs variant libraries (tools/build_variant.sh), alternating, loss-path bench.
#   tools/ab.sh VARIANT [VARIANT ...]
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
    timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train-extra 2>/dev/null | V=$v python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['V'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"
  done
done
