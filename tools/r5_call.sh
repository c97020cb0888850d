cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c4/pytest.log 2>&1; tail -5 gpurun_out/c4/pytest.log
timeout 600 python bench.py > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err
tail -3 gpurun_out/c4/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c4/bench.json'))
for k in ('value','ms_per_step','ms_update_per_window','ms_update_per_window_device','value_including_update','ms_deferred_update_per_window','ms_deferred_update_host','value_including_deferred_update'): print(k, d[k])
print(d['extra'])
PY
