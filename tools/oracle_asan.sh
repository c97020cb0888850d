#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU oracle (oracle/tef_oracle.c) over the golden tests — the one
# sanitizer run this image allows end to end (no GPU involved; the host-side ASan build of libtef_hip.so,
# tools/build_asan_host.sh, links but cannot run on the GPU box: DESIGN section 9d).
#   tools/oracle_asan.sh
set -e
cd "$(dirname "$0")/.."
OUT=$(mktemp -d)
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -mfma -fopenmp -fPIC -shared -ffp-contract=off -o $OUT/libtef_oracle_asan.so oracle/tef_oracle.c -lm
export TEF_ORACLE_LIB=$OUT/libtef_oracle_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  OMP_NUM_THREADS=4 python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider "$@"
rm -rf $OUT
