"""Where does the host time of an eager loss step go?  cProfile over 300 steps of `loss = L(); grad(loss, flows)` on the
BASELINE window (the GPU needs ~0.65 ms per step: anything above that on the host makes eager callers host-bound)."""
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
staged = []
for wi in range(2):
    fl = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
    L = Iterative(cfg, dev)
    for t in range(P):
        L.update(fl[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
                 torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
    staged.append((L, [f for row in fl for f in row]))
KK = [0]


def step():
    L, flat = staged[KK[0] % (2 if "--two" in sys.argv else 1)]
    KK[0] += 1
    loss = L()
    return torch.autograd.grad(loss, flat)


if "--graph" in sys.argv:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(); step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        out = step()
    torch.cuda.synchronize()


for _ in range(20):
    step()
torch.cuda.synchronize()
n = int(os.environ.get("N_STEPS", "300"))
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, with the final sync {1e3 * (t2 - t0) / n:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
