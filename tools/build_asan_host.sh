#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libtef_hip.so: the HOST code (plan walking, workspace layout, launch
# bookkeeping — everything that touches the caller's heap) is instrumented, the device code is compiled as usual
# (-fno-gpu-sanitize: no GPU ASan, no xnack).  Output: ${TEF_ASAN_DIR:-/tmp/tef_asan}/libtef_hip_asan.so.
# Run with   tests/asan/run_host_asan.sh   (no GPU: dry-run HIP stub)
set -e
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
if [ "$1" = "--runtime" ]; then echo $RT; exit 0; fi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/taming_event_flow_amd
OUT=${TEF_ASAN_DIR:-/tmp/tef_asan}      # outside the tree: the instrumented build must not travel with gpurun snapshots
mkdir -p $OUT
objs=""
for src in $PKG/csrc/*.hip; do
  o=$OUT/$(basename $src).o
  hipcc --offload-arch=gfx950 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
        -ffp-contract=off -fPIC -std=c++17 -munsafe-fp-atomics -I $ROOT/include -I $PKG/csrc -c $src -o $o &
  objs="$objs $o"
  while [ $(jobs -r | wc -l) -ge 4 ]; do wait -n; done
done
wait
hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -shared -fPIC -o $OUT/libtef_hip_asan.so $objs
rm -f $OUT/*.o
echo $OUT/libtef_hip_asan.so
