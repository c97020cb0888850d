#!/bin/bash
# Experiment helper: build libtef_<name>.so with extra -D flags for tef_conv.hip (A/B runs via TEF_HIP_LIB).
#   tools/build_variant.sh NAME [-DFLAG ...]
set -e
PKG=/root/repo/taming_event_flow_amd
v=$1; shift
mkdir -p $PKG/build/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -munsafe-fp-atomics "$@" -I /root/repo/include -I $PKG/csrc -c $PKG/csrc/tef_conv.hip -o $PKG/build/variants/tef_conv.$v.o
objs=""
for f in tef_common tef_loss tef_smooth tef_encode tef_resize tef_val tef_collate; do objs="$objs $PKG/build/$f.hip.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/variants/libtef_$v.so $PKG/build/variants/tef_conv.$v.o $objs
echo built $v
