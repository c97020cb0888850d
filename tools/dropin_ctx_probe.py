#!/usr/bin/env python3
"""What in a process's history changes the speed of the literal train_flow.py loop on ONE stream (TEF_LAZY_FLOWS=0)?
Runs bench.dropin_extra after an optional prologue:  none | profiled (one small loss step with per-kernel HIP events) |
streams (create and use 8 torch streams) | graph (capture + replay one small hipGraph) | busy (2 s of matmuls)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
from taming_event_flow_amd import _lib, synth  # noqa: E402
from taming_event_flow_amd.loss.flow import Iterative  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prologue", default="none")
ns = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.lib()


def small_loss():
    B, H, W, P, F, N = 2, 64, 64, 4, 2, 2000
    cfg = {"loader": {"resolution": [H, W], "batch_size": B},
           "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"},
           "data": {"passes_loss": P, "scales_loss": 1}}
    win = synth.make_window(np.random.default_rng(0), B, H, W, P, F, N, 0, sigma=2.0)
    L = Iterative(cfg, dev)
    flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
    for t in range(P):
        L.update(flows[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
                 torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
    L().backward()
    torch.cuda.synchronize()


if ns.prologue == "profiled":
    lib.tef_profile_enable(1)
    small_loss()
    lib.tef_profile_collect()
    lib.tef_profile_enable(0)
elif ns.prologue == "streams":
    ss = [torch.cuda.Stream() for _ in range(8)]
    for s_ in ss:
        with torch.cuda.stream(s_):
            torch.zeros(1024, device=dev).add_(1)
    torch.cuda.synchronize()
elif ns.prologue == "graph":
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(1024, device=dev)
    s_ = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s_):
        x.add_(1)
    g.replay()
    torch.cuda.synchronize()
elif ns.prologue == "busy":
    a_ = torch.randn(4096, 4096, device=dev)
    import time
    t0 = time.time()
    while time.time() - t0 < 2.0:
        (a_ @ a_).sum().item()


class A:
    batch, res, events, detached, passes, steps = 8, [128, 128], 10000, 0, 10, 6


out = bench.dropin_extra(A, torch, dev, windows=6)
print(ns.prologue, os.environ.get("TEF_LAZY_FLOWS", "1"), out.get("dropin_window_ms"), out.get("dropin_host_ms"), out.get("error"))
