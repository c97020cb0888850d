"""Phase counters of the IWE scatter kernel K2 (experiment builds with -DTEF_K2_PROBE only: tools/variant.sh ... -DTEF_K2_PROBE).
   TEF_HIP_LIB=.../libtef_probe.so python tools/k2_probe.py [steps]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taming_event_flow_amd import _lib, synth  # noqa: E402
from taming_event_flow_amd.loss.flow import Iterative  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B, H, W, P, F, N = 8, 128, 128, 10, 4, 10000
cfg = {"loader": {"resolution": [H, W], "batch_size": B},
       "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"},
       "data": {"passes_loss": P, "scales_loss": 1}}
win = synth.make_window(np.random.default_rng(0), B, H, W, P, F, N, 0, sigma=2.0, kind="smooth")
flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
L = Iterative(cfg, dev)
for t in range(P):
    L.update(flows[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
             torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    L()
torch.cuda.synchronize()
lib.tef_k2_probe_read(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    L()
e1.record()
torch.cuda.synchronize()
lib.tef_k2_probe_read(out)
v = [int(x) for x in out]
names = ["items", "prologue (masks, list, zero)", "-", "-", "sweep", "barrier2 wait", "stats (all waves)",
         "final barrier (all)", "stats: halo", "stats: main loop", "-", "-", "-", "kernel ticks (per wg)", "wgs"]
items, wgs = max(v[0], 1), max(v[14], 1)
print("forward ms/step", e0.elapsed_time(e1) / steps, "items/step", items / steps, "wgs/step", wgs / steps)
print("kernel ticks per workgroup", v[13] / wgs)
for k, n in enumerate(names):
    if n == "-" or k in (0, 13, 14):
        continue
    per = 8
    print(f"{n:24s} per item {v[k] / items / per:10.1f} ticks   x items/wg {v[k] / per / wgs:10.1f}")
