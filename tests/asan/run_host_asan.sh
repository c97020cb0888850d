#!/bin/bash
# Host-side ASan + UBSan run of libtef_hip.so without a GPU: see tests/asan/dry_run_driver.py.   tests/asan/run_host_asan.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
# (outside the tree, keyed by the sources' hash: a stale instrumented build is never tested)
H=$(cat $ROOT/taming_event_flow_amd/csrc/* $ROOT/include/*.h $ROOT/tools/build_asan_host.sh | sha256sum | cut -c1-16)
export TEF_ASAN_DIR=${TEF_ASAN_DIR:-/tmp/tef_asan_$H}
DIR=$TEF_ASAN_DIR
LIB=$DIR/libtef_hip_asan.so
[ -f $LIB ] || $ROOT/tools/build_asan_host.sh
STUB=$DIR/libhipstub.so
gcc -O1 -g -fPIC -shared -o $STUB $ROOT/tests/asan/hip_stub.c
RT=$($ROOT/tools/build_asan_host.sh --runtime)
UB=$(dirname $RT)/libclang_rt.ubsan_standalone-x86_64.so
# PYTHONMALLOC=malloc: every Python object (the ctypes arrays and structs handed to the library among them) is a malloc block
# of its own, with redzones
LD_PRELOAD=$RT:$STUB TEF_HIP_LIB=$LIB TEF_HIP_STUB=$STUB PYTHONMALLOC=malloc \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_odr_violation=0:protect_shadow_gap=0 \
  UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python $ROOT/tests/asan/dry_run_driver.py "$@"
