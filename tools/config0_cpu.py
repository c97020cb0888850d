#!/usr/bin/env python3
"""BASELINE.json configs[0] ("synthetic 10k-event 128x128 stream, reference loss/flow.py EventWarping on CPU PyTorch, bs=1:
plumbing, no GPU"): the reference itself and the C port (oracle/) timed side by side in the build container, forward +
backward of one Iterative/two window (P = 10, F = 4), min and median of >= 5 repetitions.  Needs /root/reference.

    python tools/config0_cpu.py > profiles/r04_config0_cpu.json
    python tools/config0_cpu.py --batch 8 --events 10000 --threads 8 > profiles/r05_config1_cpu_b8.json
        (the BASELINE window bench.py times, B = 8: the reference itself against the port that serves as `cpu_baseline`)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from taming_event_flow_amd import synth  # noqa: E402

sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
from loss.flow import Iterative  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--events", type=int, nargs="*", default=[1000, 10000])
ap.add_argument("--threads", type=int, nargs="*", default=[1, 8])
args = ap.parse_args()
H = W = 128
P, F, B = 10, 4, args.batch
out = {"config": f"BASELINE configs[{0 if B == 1 else 1}]: Iterative/two fwd+bwd, 128x128, B={B}, P=10, F=4", "host_cores": os.cpu_count(),
       "rows": []}
for N in args.events:
    rng = np.random.default_rng(5)
    win = synth.make_window(rng, B, H, W, P, F, N, 0, sigma=2.0)
    cfg = {"loader": {"resolution": [H, W], "batch_size": B},
           "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"},
           "data": {"passes_loss": P, "scales_loss": 1}}

    def ref_once():
        L = Iterative(cfg, torch.device("cpu"))
        flows = [[torch.tensor(win["flows"][t][i], requires_grad=True) for i in range(F)] for t in range(P)]
        for t in range(P):
            L.update(flows[t], torch.tensor(win["ev"][t]).clone(), torch.tensor(win["pm"][t]).clone(),
                     torch.tensor(win["dev"][t]).clone(), torch.tensor(win["dpm"][t]).clone())
        t0 = time.perf_counter()
        loss = L()
        loss.backward()
        return time.perf_counter() - t0, float(loss.item())

    ow = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=1, mode="two")

    def port_once():
        t0 = time.perf_counter()
        l, _ = ow.iterative(backward=True)
        return time.perf_counter() - t0, float(l)

    for threads in args.threads:
        torch.set_num_threads(threads)
        oracle.threads(threads)
        for name, fn in (("reference (CPU PyTorch)", ref_once), ("port (oracle/tef_oracle.c)", port_once)):
            fn()
            ts, loss = [], None
            while len(ts) < 5 or (sum(ts) < 8.0 and len(ts) < 50):
                dt, loss = fn()
                ts.append(dt)
            ts.sort()
            ev = B * P * N
            out["rows"].append({"events_per_pass": N, "threads": threads, "impl": name, "reps": len(ts),
                                "ms_min": round(1e3 * ts[0], 2), "ms_median": round(1e3 * ts[len(ts) // 2], 2),
                                "events_per_s_median": round(ev / ts[len(ts) // 2], 1), "loss": loss})
            print(out["rows"][-1], file=sys.stderr)
print(json.dumps(out, indent=1))
