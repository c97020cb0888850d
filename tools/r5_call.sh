cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for pr in 0 1; do
TEF_STREAM_PRIO=$pr timeout 600 python bench.py --mode train --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PRIO=$pr eager', d['ms_per_step'])"
TEF_STREAM_PRIO=$pr timeout 600 python bench.py --mode train --graph --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PRIO=$pr graph', d['ms_per_step'])"
done
done
