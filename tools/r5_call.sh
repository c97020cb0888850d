cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
timeout 600 python -m pytest tests/test_pack_gpu.py tests/test_loss_gpu.py tests/test_flow_val_gpu.py -m gpu -q > gpurun_out/c3/pytest.log 2>&1; tail -3 gpurun_out/c3/pytest.log
timeout 300 python tools/profile_update.py 2>&1 | grep "update x"
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-train-extra > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c3/bench.json'))
print(d['ms_per_step'], d['ms_update_per_window'], d['ms_update_per_window_device'], d['value_including_update'], d['parity_vs_golden'])
PY
