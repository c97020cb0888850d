#!/bin/bash
# experiment: per-layer conv timings with the halo split factor forced (variant zenv: tools/build_variant.sh zenv -DTEF_HALO_SPLIT_ENV)
#   tools/zsweep.sh "0 2 3 4 6" [conv_bench args]
cd "$(dirname "$0")/.."
export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_zenv.so
zs=$1; shift
for z in $zs; do
  echo "== z=$z $*"
  TEF_HALO_Z=$z timeout 200 python tools/conv_bench.py "$@" 2>/dev/null | grep -v "pred\|head\|layer"
done
