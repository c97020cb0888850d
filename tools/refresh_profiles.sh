#!/bin/bash
# Re-measure the end-of-round numbers on the GPU box into gpurun_out/final/ (copy what is judged into profiles/).
#   tools/refresh_profiles.sh [ROUND_TAG]        (default r02)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r06}
O=gpurun_out/final
mkdir -p $O
timeout 400 python bench.py > $O/${R}_loss_bench.json 2> $O/loss_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-train-extra --detached 5000 > $O/${R}_loss_bench_nd5000.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-train-extra --flow iid > $O/${R}_loss_bench_iid_flows.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-train-extra --warping Linear > $O/${R}_loss_bench_linear.json 2>/dev/null
timeout 300 python bench.py --mode train --steps 5 --warmup 2 > $O/${R}_train_bench_eager.json 2> $O/train_eager.err
timeout 300 python bench.py --mode train --graph --steps 10 --warmup 2 > $O/${R}_train_bench_graph.json 2> $O/train_graph.err
timeout 300 python bench.py --mode eval --steps 5 --warmup 2 > $O/${R}_eval_bench.json 2> $O/eval.err
timeout 300 python tools/conv_bench.py > $O/${R}_conv_bench.txt 2>/dev/null
# (kernel durations are a kernel's own only when nothing runs beside it: the profiled training run keeps to ONE stream)
TEF_TWO_STREAMS=0 TEF_WGRAD_GROUP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/${R}_train_bench_under_rocprof.json 2> $O/train_prof.err
TEF_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -- python3 bench.py --mode eval --steps 3 --warmup 1 > $O/${R}_eval_bench_under_rocprof.json 2> $O/eval_prof.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/loss_stats -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-extra > $O/${R}_loss_bench_under_rocprof.json 2> $O/loss_prof.err
# HBM traffic: one counter per pass (MI355X_MICROARCH.md, HBM section)
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-kernel-events > /dev/null 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-extra --no-kernel-events > /dev/null 2> $O/pmc_write.err
for d in train eval loss; do f=$(find $O/${d}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${R}_${d}_kernel_stats.csv; done
f=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/${R}_pmc_fetch_size.csv
f=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/${R}_pmc_write_size.csv
[ -f $O/${R}_pmc_fetch_size.csv ] && [ -f $O/${R}_pmc_write_size.csv ] && python tools/pmc_summary.py $O/${R}_pmc_fetch_size.csv $O/${R}_pmc_write_size.csv $O/${R}_pmc_traffic.json
timeout 300 python tools/conv_layer_attribution.py $O/${R}_conv_layer_attribution.csv > $O/${R}_conv_layer_attribution.txt 2>/dev/null
ls $O | head -40
cut -c1-300 $O/${R}_loss_bench.json; echo; cut -c1-200 $O/${R}_train_bench_graph.json; echo; cut -c1-200 $O/${R}_eval_bench.json
