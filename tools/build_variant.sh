#!/bin/bash
# Experiment helper: build libtef_<name>.so with extra -D flags for one translation unit (A/B runs via TEF_HIP_LIB).
#   tools/build_variant.sh NAME [-DFLAG ...]            (SRC=tef_conv by default; SRC=tef_loss tools/build_variant.sh ...)
set -e
PKG=/root/repo/taming_event_flow_amd
SRC=${SRC:-tef_conv}
v=$1; shift
mkdir -p $PKG/build/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -munsafe-fp-atomics "$@" -I /root/repo/include -I $PKG/csrc -c $PKG/csrc/$SRC.hip -o $PKG/build/variants/$SRC.$v.o
objs=""
for f in tef_common tef_loss tef_smooth tef_encode tef_resize tef_val tef_collate tef_conv; do
  if [ "$f" != "$SRC" ]; then objs="$objs $PKG/build/$f.hip.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/variants/libtef_$v.so $PKG/build/variants/$SRC.$v.o $objs
echo built $v
