#!/bin/bash
# Counter passes for one kernel of the loss step: tools/pmc_kernel.sh <kernel-regex> <outdir> "<counters pass 1>" ["<pass 2>" ...]
# (counters only: no trace domains; one rocprofv3 run per pass)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
K=$1; O=$2; shift 2
mkdir -p $O
i=1
for P in "$@"; do
  timeout -k 5 100 rocprofv3 --pmc $P --kernel-include-regex "$K" --output-format csv -d $O/p$i -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-train-extra > $O/p$i.json 2> $O/p$i.err
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{k:40s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
