#!/bin/bash
# kernel trace of the multi-stream training window (captured + eager) and its phase / overlap analysis
cd "$(dirname "$0")/.."
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pw
cd /tmp && export TMPDIR=/tmp
for m in graph eager; do
  flag=""; [ $m = graph ] && flag="--graph"
  rm -rf /tmp/pw_$m
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pw_$m -- python3 $R/bench.py --mode train $flag --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pw/bench_$m.json 2> $R/gpurun_out/pw/err_$m.txt
  f=$(find /tmp/pw_$m -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/window_phases.py $f > $R/gpurun_out/pw/phases_$m.txt 2>&1
  python3 $R/tools/trace_overlap.py $f --tail 0.1 > $R/gpurun_out/pw/overlap_$m.txt 2>&1
done
