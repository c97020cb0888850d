"""Where does the host time of an EAGER training window go (bench.py --mode train without --graph)?  cProfile over a few
windows of Trainer.step on the BASELINE configs[2] workload; the GPU needs ~38 ms per window."""
import cProfile
import copy
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from taming_event_flow_amd import train  # noqa: E402

dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
torch.manual_seed(1234)
tr = train.Trainer(cfg, dev)
src = train.SyntheticSequences(cfg, dev, 10000, seq_len=10 ** 9, seed=100)
P = cfg["data"]["passes_loss"]
tr.reset()


def window():
    for _ in range(P):
        tr.step(src.next(), new_seq=False)


for _ in range(3):
    window()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    window()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/window, with the final sync {1e3 * (t2 - t0) / n:.2f} ms/window")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    window()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
