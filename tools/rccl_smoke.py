"""RCCL sanity on however many GPUs the box has (1 on the development box): the calls the DP path makes — process group
on "nccl" with a device id, the flat-gradient all-reduce, the gloo side group for the new_seq flag, barrier — so that a
missing library / a bad init argument shows up before the multi-GPU bench does.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from taming_event_flow_amd import parallel  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
flat = torch.full((31_365_352,), float(rank + 1), device=dev)          # RecEVFlowNet's gradient bucket
dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
want = world * (world + 1) / 2
assert float(flat[0]) == want and float(flat[-1]) == want, (float(flat[0]), want)
t0 = time.perf_counter()
for _ in range(10):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 100
flag = torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32)
dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=parallel._flag_group())
assert int(flag) == 1
dist.barrier()
if rank == 0:
    print(json.dumps({"rccl_smoke": "ok", "world": world, "allreduce_125MB_ms": round(ms, 3),
                      "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))}))
dist.destroy_process_group()
