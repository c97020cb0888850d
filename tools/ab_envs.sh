#!/bin/bash
# A/B of several environment settings on one box, training window (captured + eager):  tools/ab_envs.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then pre=""; else pre="$v"; fi
    g=$(env $pre python bench.py --mode train --graph --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    e=$(env $pre python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$v graph $g eager $e"
  done
done
