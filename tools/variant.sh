#!/bin/bash
# Experiment helper: build libtef_<name>.so from a PATCHED COPY of one translation unit (the shipped sources carry no
# experiment hooks).  A/B runs pick the library through TEF_HIP_LIB (tools/ab.sh).  The variants must travel to the GPU box with
# the snapshot, so they live in the tree: delete taming_event_flow_amd/build/variants when the experiment is over.
#   SRC=tef_loss tools/variant.sh NAME 'sed -E expression' [-DFLAG ...]
set -e
PKG=/root/repo/taming_event_flow_amd
SRC=${SRC:-tef_loss}
v=$1; expr=$2; shift 2
mkdir -p $PKG/build/variants/src
sed -E "$expr" $PKG/csrc/$SRC.hip > $PKG/build/variants/src/$SRC.$v.hip
if cmp -s $PKG/csrc/$SRC.hip $PKG/build/variants/src/$SRC.$v.hip; then echo "variant $v: the expression changed nothing" >&2; exit 1; fi
UNSAFE=$(python3 -c "import sys; sys.path.insert(0, '/root/repo'); from taming_event_flow_amd import build as b; print('-munsafe-fp-atomics' if '$SRC' in b.UNSAFE_FP_ATOMICS else '')")
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $UNSAFE "$@" -I /root/repo/include -I $PKG/csrc -c $PKG/build/variants/src/$SRC.$v.hip -o $PKG/build/variants/$SRC.$v.o 2>/dev/null
objs=""
for f in tef_common tef_loss tef_smooth tef_encode tef_resize tef_val tef_collate tef_conv tef_cell tef_net tef_optim tef_prims; do
  if [ "$f" != "$SRC" ]; then objs="$objs $PKG/build/$f.hip.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/build/variants/libtef_$v.so $PKG/build/variants/$SRC.$v.o $objs
echo built $v
