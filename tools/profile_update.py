"""Host time of Iterative.update() over one BASELINE window (10 passes), cProfile."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from taming_event_flow_amd import synth  # noqa: E402
from taming_event_flow_amd.loss.flow import Iterative  # noqa: E402

dev = torch.device("cuda:0")
B, H, W, P, F, N = 8, 128, 128, 10, 4, 10000
cfg = {"loader": {"resolution": [H, W], "batch_size": B},
       "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"},
       "data": {"passes_loss": P, "scales_loss": 1}}
win = synth.make_window(np.random.default_rng(0), B, H, W, P, F, N, 0, sigma=2.0)
flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]


def lists():
    return [(torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
             torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev)) for t in range(P)]


L = Iterative(cfg, dev)
for rep in range(4):
    evs = lists()
    L.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(P):
        L.update(flows[t], *evs[t])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"update x{P}: host {1e3 * (t1 - t0):.3f} ms, with sync {1e3 * (time.perf_counter() - t0):.3f} ms")
evs = lists()
L.reset()
pr = cProfile.Profile()
pr.enable()
for t in range(P):
    L.update(flows[t], *evs[t])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
