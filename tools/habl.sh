#!/bin/bash
# experiment: halo-kernel ablations (variants hNOLOAD hNOMFMA hNOLDS hNOLOADLDS), forward time of the GRU layers
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-base hNOLOAD hNOMFMA hNOLDS hNOLOADLDS}; do
  if [ "$v" = base ]; then unset TEF_HIP_LIB; else export TEF_HIP_LIB=$PWD/taming_event_flow_amd/build/variants/libtef_$v.so; fi
  echo "== $v"
  timeout 200 python tools/conv_bench.py --only gru --reps 30 2>/dev/null | awk '{print $1,$2,$3,$4,$5,$6,$7}' | grep -v layer
done
