"""Host cost of the DP lock-step flag exchange without a GPU (round 6): WORLD CPU processes on a gloo group,
`parallel.any_rank` (one int, all-reduce MAX: what every rank does on every pass) and `parallel.any_rank_mask` (P ints, once
per window: Trainer.declare_fixed_sequences), microseconds per call.

    python tools/lockstep_timing.py [--world 8] [--calls 2000] [--passes 10]
"""
import argparse
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, calls, P, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taming_event_flow_amd import parallel

    out = {}
    for name, fn in (("any_rank", lambda k: parallel.any_rank(k % 7 == rank)),
                     ("any_rank_mask", lambda k: parallel.any_rank_mask(1 << (k % P) if k % 5 == rank % 5 else 0, P))):
        for k in range(50):
            fn(k)
        dist.barrier()
        ts = []
        for k in range(calls):
            t0 = time.perf_counter()
            fn(k)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = (1e6 * ts[len(ts) // 2], 1e6 * ts[int(len(ts) * 0.99)], 1e6 * sum(ts) / len(ts))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--passes", type=int, default=10)
    a = ap.parse_args()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, a.world, port, a.calls, a.passes, q)) for r in range(a.world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join()
    print(f"world {a.world}, {os.cpu_count()} host cores, {a.calls} calls; microseconds per call (median / p99 / mean), slowest rank:")
    for name in ("any_rank", "any_rank_mask"):
        worst = max((r[1][name] for r in res), key=lambda t: t[0])
        print(f"  {name:14s} {worst[0]:8.1f} / {worst[1]:8.1f} / {worst[2]:8.1f}")
