#!/bin/bash
# kernel-trace summary of one-stream training windows: tools/prof_train.sh OUTDIR [env assignments...]
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export TEF_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$out/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$out/err.txt
find $GRAFT_REPO_ROOT/gpurun_out/$out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/$out/kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/$out -name "*.csv" ! -name kernel_stats.csv -size +3M -delete
