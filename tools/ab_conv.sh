#!/bin/bash
# A/B on one GPU box, conv path: default library vs variants (tools/build_variant.sh), per-layer forward / backward times
#   tools/ab_conv.sh "VARIANT ..." [conv_bench args]
cd "$(dirname "$0")/.."
vs=$1; shift
for r in 1 2; do
  for v in base $vs; do
    if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
    echo "== $v"
    timeout 200 python tools/conv_bench.py "$@" 2>/dev/null | grep -v "pred\|head\|layer"
  done
done
