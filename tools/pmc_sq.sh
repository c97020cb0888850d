#!/bin/bash
# SQ counter passes for one kernel of the loss step: tools/pmc_sq.sh <kernel-regex> <outdir>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
K=${1:-splat_kernel}
O=${2:-gpurun_out/pmc_sq}
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
i=1
for P in "$P1" "$P2"; do
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "$K" --output-format csv -d $O/p$i -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-train-extra > $O/p$i.json 2> $O/p$i.err
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{k:60s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
