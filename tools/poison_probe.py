#!/usr/bin/env python3
"""Does any kernel read memory nobody wrote?  Every `torch.empty` buffer (arenas, workspaces, packed weights, loss
containers) is filled with NaN / 0xFF by torch (`torch.utils.deterministic.fill_uninitialized_memory` under
`use_deterministic_algorithms`) and the training windows are compared with the same windows run without the poison: a
kernel that reads a word it did not write — harmless while the allocator hands out zeros or the previous, identical
run's bytes — shows as NaN or as a different loss / gradient norm.

    python tools/poison_probe.py [--streams 0|1] [--warping Iterative|Linear] [--smooth 0|1] [--cut 0|1]
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=0)
ap.add_argument("--warping", default="Iterative")
ap.add_argument("--smooth", type=int, default=0)
ap.add_argument("--cut", type=int, default=1)
ap.add_argument("--scales", type=int, default=1)
a = ap.parse_args()
g.build()
from taming_event_flow_amd import train  # noqa: E402

dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
cfg["data"].update(passes_loss=4, scales_loss=a.scales)
cfg["loss"].update(warping=a.warping)
if a.smooth:
    cfg["loss"].update(flow_spat_smooth_weight=0.001, flow_temp_smooth_weight=0.1)
cfg["optimizer"]["lr"] = 0.0


def run(poison):
    torch.use_deterministic_algorithms(poison, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = poison
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=bool(a.streams))
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
    tr.reset()
    out = []
    for t in range((2 if a.cut else 0) + 8):
        if tr.step(src.next(), new_seq=(a.cut and t == 2)):
            out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
    tr.close()
    return np.array(out)


clean = run(False)
dirty = run(True)
clean2 = run(False)
print("clean :", clean)
print("poison:", dirty)
print("clean2:", clean2)
ok = np.isfinite(dirty).all() and np.allclose(dirty, clean, rtol=1e-4)
print("RESULT", "ok" if ok else "UNINITIALISED READ", vars(a))
sys.exit(0 if ok else 1)
