cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print({k:e.get(k) for k in ('train_window_ms','train_window_eager_ms','dropin_window_ms','dropin_windows_timed','dropin_host_ms','dropin_fresh_process_window_ms','dropin_fresh_process_host_ms')})"
done
TEF_LAZY_FLOWS=0 timeout 900 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']
print('LAZY=0', {k:e.get(k) for k in ('train_window_eager_ms','dropin_window_ms','dropin_host_ms','dropin_fresh_process_window_ms')})"
