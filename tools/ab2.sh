#!/bin/bash
# A/B on one GPU box, loss-path bench: default library vs variant libraries under taming_event_flow_amd/build/variants
# (libtef_<name>.so), alternating; prints ms per step, the timed window's parity against the recorded reference result and
# the per-kernel times.   tools/ab2.sh [-n STEPS] VARIANT [VARIANT ...]
cd "$(dirname "$0")/.."
steps=50
if [ "$1" = "-n" ]; then steps=$2; shift 2; fi
for r in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
    timeout -k 10 240 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-train-extra 2>gpurun_out/ab2_$v.err | V=$v python -c "
import json,sys,os
t=sys.stdin.read()
try:
    d=json.loads(t); p=d.get('parity_vs_golden') or {}
    print(os.environ['V'], d['ms_per_step'], 'parity', p.get('loss_rel_err'), p.get('dflow_lattice_max_rel_err'), {k:round(v['ms'],5) for k,v in d['kernels'].items()})
except Exception as e:
    print(os.environ['V'], 'FAILED', repr(e), t[-300:])"
  done
done
