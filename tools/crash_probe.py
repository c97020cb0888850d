#!/usr/bin/env python3
"""Bisection helper for an intermittent host-heap corruption seen at interpreter exit (glibc: "corrupted size vs.
prev_size" / "free(): invalid pointer") after short training runs: one Trainer, a few windows, then a burst of host
allocations that trips over a damaged heap if there is one.

    python tools/crash_probe.py JITTER MAX_GRAD STREAMS(0|1|2 = one Trainer of each kind, one after the other) [WARPING] [SCALES] [SMOOTHING 0|1]
"""
import copy
import gc
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as g

g.build()
from taming_event_flow_amd import train

jitter, max_grad, streams = int(sys.argv[1]), int(sys.argv[2]) or None, sys.argv[3] == "1"
warping = sys.argv[4] if len(sys.argv) > 4 else "Iterative"
scales = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=max_grad)
cfg["data"].update(passes_loss=4, scales_loss=scales)
cfg["loss"].update(warping=warping)
cfg["optimizer"]["lr"] = 0.0
smooth = len(sys.argv) > 6 and sys.argv[6] == "1"
if smooth:
    cfg["loss"].update(flow_spat_smooth_weight=0.001, flow_temp_smooth_weight=0.1)
for st in ([False, True] if sys.argv[3] == "2" else [False, True, False, True, False, True] if sys.argv[3] == "6" else [streams]):
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=st)
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=jitter)
    tr.reset()
    for _ in range(3):
        for _ in range(4):
            tr.step(src.next(), new_seq=False)
    print(st, float(tr.last_loss.item()), float(tr.last_grad_norm.item()))
    del tr, src
    gc.collect()
    torch.cuda.synchronize()
junk = [bytearray(n) for n in range(16, 4096, 8)] * 4      # walk the host heap
del junk
gc.collect()
import ast  # noqa: E402

for _ in range(20):
    ast.parse(open(__file__).read())
print("clean exit")
