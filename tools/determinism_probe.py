#!/usr/bin/env python3
"""Same training windows four ways — captured hipGraph / eager, one stream / the Trainer's streams — with lr = 0 (the
windows then differ through the recurrent state only): the printed (loss, gradient norm) sequences must agree to fp32
summation order.  With a learning rate they do not, from one PROCESS to the next, on any launch path: Adam's first steps
move a weight by +-lr whatever its gradient's size, and the next window's gradient norm lands on one of a few values
percents apart.

    python tools/determinism_probe.py graph|eager        (TEF_TWO_STREAMS=0|1)
"""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
g.build()
from taming_event_flow_amd import train
dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
cfg["data"].update(passes_loss=4, scales_loss=1)
cfg["optimizer"]["capturable"] = True
cfg["optimizer"]["lr"] = 0.0
torch.manual_seed(7)
tr = train.Trainer(cfg, dev)
src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
tr.reset()
mode = sys.argv[1]
if mode == "graph":
    win = tr.capture_window([src.next() for _ in range(4)], warmup=1)
    for r in range(8):
        win()
        print(mode, r, float(tr.last_loss.item()), float(tr.last_grad_norm.item()))
else:
    batches = [src.next() for _ in range(4)]
    for r in range(8):
        for b in batches:
            tr.step({k: v.clone() for k, v in b.items()}, new_seq=False)
        print(mode, r, float(tr.last_loss.item()), float(tr.last_grad_norm.item()))
