cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
timeout 1200 python -m pytest tests/test_loss_gpu.py tests/test_prims_gpu.py tests/test_flow_val_gpu.py -m gpu -x -q > gpurun_out/c7/pytest.log 2>&1; tail -4 gpurun_out/c7/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-train-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['parity_vs_golden']['loss_rel_err'], d['parity_vs_golden']['dflow_lattice_max_rel_err'], {k:round(v['ms'],5) for k,v in d['kernels'].items()})"; done
