#!/bin/bash
# Re-measure the end-of-round numbers on the GPU box into gpurun_out/final/ (copy what is judged into profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 300 python bench.py --steps 50 --warmup 5 > $O/loss_bench.json 2> $O/loss_bench.err
timeout 300 python bench.py --mode train --steps 5 --warmup 2 > $O/train_bench_eager.json 2> $O/train_eager.err
timeout 300 python bench.py --mode train --graph --steps 5 --warmup 2 > $O/train_bench_graph.json 2> $O/train_graph.err
timeout 300 python bench.py --mode eval --steps 5 --warmup 2 > $O/eval_bench.json 2> $O/eval.err
timeout 300 python tools/conv_bench.py > $O/conv_bench.txt 2>/dev/null
timeout 300 python tools/conv_bench.py --batch 1 --res 480 --width 640 > $O/conv_bench_480x640.txt 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/train_bench_under_rocprof.json 2> $O/train_prof.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval_stats -- python3 bench.py --mode eval --steps 3 --warmup 1 > $O/eval_bench_under_rocprof.json 2> $O/eval_prof.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/loss_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/loss_bench_under_rocprof.json 2> $O/loss_prof.err
find $O -name "*kernel_stats.csv" | head
tail -c 300 $O/loss_bench.json; echo; cut -c1-200 $O/train_bench_graph.json; cut -c1-200 $O/eval_bench.json
