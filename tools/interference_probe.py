#!/usr/bin/env python3
"""In one process: other trainers (captured windows, lagging multi-stream windows, other losses) come and go WITHOUT any
tidy-up, then a one-stream trainer runs a fixed workload whose correct result is known; reports every deviation.
Looks for the interference seen once in ~25 pytest processes (tests/test_train_gpu.py without the harness's
collect-and-synchronize): the one-stream run of test_window_cut_short_by_new_seq returned a different loss / gradient.

    python tools/interference_probe.py [--iters 40] [--phases ABC] [--gc-threshold 0]
"""
import argparse
import copy
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--phases", default="ABC")
ap.add_argument("--gc-threshold", type=int, default=0)
a = ap.parse_args()
g.build()
from taming_event_flow_amd import train  # noqa: E402

dev = torch.device("cuda:0")
if a.gc_threshold:
    gc.set_threshold(a.gc_threshold, 2, 2)


def make_cfg(warping="Iterative", scales=1, smooth=False, capturable=False):
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4, scales_loss=scales)
    cfg["loss"].update(warping=warping)
    if smooth:
        cfg["loss"].update(flow_spat_smooth_weight=0.001, flow_temp_smooth_weight=0.1)
    cfg["optimizer"]["capturable"] = capturable
    cfg["optimizer"]["lr"] = 0.0
    return cfg


def target():
    """the one-stream half of test_window_cut_short_by_new_seq"""
    cfg = make_cfg()
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=False)
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
    tr.reset()
    out = []
    for t in range(2 + 8):
        if tr.step(src.next(), new_seq=(t == 2)):
            out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
    return np.array(out)


def phase_a():      # captured multi-stream window (test_multi_stream_window_matches_one_stream[..., graph=True])
    cfg = make_cfg(capturable=True)
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=True)
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
    tr.reset()
    win = tr.capture_window([src.next() for _ in range(4)], warmup=1)
    for _ in range(3):
        for b in win.inputs:
            for k, v in src.next().items():
                b[k].copy_(v)
        win()
    return float(tr.last_loss.item())


def phase_b():      # lagging streams (test_two_stream_window_has_no_race)
    cfg = make_cfg()
    for delay in ((0, 30_000_000, 0), (0, 0, 120_000_000)):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=True)
        tr.model.arch.engine.debug_delay = delay
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
        tr.reset()
        for _ in range(2):
            for _ in range(4):
                tr.step(src.next(), new_seq=False)
    return float(tr.last_loss.item())


def phase_c():      # Linear + smoothing + two scales, multi-stream
    cfg = make_cfg("Linear", 2, True)
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=True)
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
    tr.reset()
    for _ in range(3):
        for _ in range(4):
            tr.step(src.next(), new_seq=False)
    return float(tr.last_loss.item())


ref = target()
print("reference", ref, flush=True)
bad = 0
for it in range(a.iters):
    for ph in a.phases:
        {"A": phase_a, "B": phase_b, "C": phase_c}[ph]()
    got = target()
    err = np.abs(got - ref) / np.abs(ref)
    if not (err[0::2] <= 1e-6).all() or not (err[1::2] <= 1e-4).all():
        bad += 1
        print(f"iteration {it}: DEVIATION {got} (relative {err})", flush=True)
print(f"phases {a.phases}: {bad} deviations in {a.iters} iterations", flush=True)
sys.exit(1 if bad else 0)
