#!/usr/bin/env python3
"""Stress test for the intermittent host-heap corruption of round 3 (glibc "corrupted size vs. prev_size" /
"free(): invalid pointer" in ~1 pytest process in 10, only with multi-stream trainers and captured windows).

Creates and drops trainers + captured windows WITHOUT the test harness's collect-and-synchronize: objects are released
by reference counting or by the cycle collector at whatever allocation it wakes up on.  `--gc-threshold N` lowers the
collector's threshold so that it fires at arbitrary points inside the training steps (the crash needs the collector to run
while work is in flight); `--sync-before-drop` is the control: wait for the device before references are dropped.

    python tools/teardown_stress.py [--iters 12] [--gc-threshold 20] [--graphs 1] [--sync-before-drop 0]

Exit code 0 and the line "clean exit" = no abort.  Run it in a loop (tools/teardown_stress.sh) and count.
"""
import argparse
import copy
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--gc-threshold", type=int, default=20)
ap.add_argument("--graphs", type=int, default=1)
ap.add_argument("--sync-before-drop", type=int, default=0)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--close", type=int, default=0, help="call Trainer.close() / CapturedWindow.close() before dropping")
a = ap.parse_args()

g.build()
from taming_event_flow_amd import train  # noqa: E402

dev = torch.device("cuda:0")
cfg = copy.deepcopy(train.DEFAULT_CONFIG)
cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
cfg["data"].update(passes_loss=4)
cfg["optimizer"]["lr"] = 1e-5
P = 4
if a.gc_threshold > 0:
    gc.set_threshold(a.gc_threshold, 2, 2)      # the cycle collector wakes up every few allocations, in any thread

for it in range(a.iters):
    torch.manual_seed(7 + it)
    tr = train.Trainer(cfg, dev, streams=bool(a.streams))
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3 + it, jitter=50)
    tr.reset()
    for _ in range(2):
        for _ in range(P):
            tr.step(src.next(), new_seq=False)
    cw = None
    if a.graphs:
        cw = tr.capture_window([src.next() for _ in range(P)], warmup=1)
        for _ in range(3):
            cw.replay()
    loss = tr.last_loss
    if a.sync_before_drop:
        torch.cuda.synchronize()
    if a.close:
        if cw is not None:
            cw.close()
        tr.close()
    # references dropped with the last replay / window possibly still running; no gc.collect(), no synchronize
    del tr, src, cw
    junk = [bytearray(n) for n in range(16, 2048, 16)]      # walk the host heap
    del junk
    print(it, float(loss.item()), flush=True)
print("clean exit", flush=True)
