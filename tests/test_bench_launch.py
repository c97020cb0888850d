"""`python bench.py --gpus N` must really start N ranks (one process per GPU, torch.distributed.run rendezvous on
127.0.0.1) and refuse a world size that contradicts --gpus.  TEF_BENCH_LAUNCH_ONLY=1 makes every rank report its
environment and exit before anything touches a GPU, so the launcher is testable here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "3"], {"TEF_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    import re

    lines = [json.loads(m) for m in re.findall(r'\{"launch_only".*?\}', r.stdout)]     # ranks share one stdout
    assert sorted((d["rank"], d["world"], d["local"]) for d in lines) == [(0, 2, 0), (1, 2, 1)]


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4"], {"TEF_BENCH_LAUNCH_ONLY": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "1"], {"TEF_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0 and json.loads(r.stdout.strip())["world"] == 1
