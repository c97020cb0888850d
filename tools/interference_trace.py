#!/usr/bin/env python3
"""Follow-up of tools/pytest_sequence_probe.py: the captured multi-stream window test followed by the one-stream cut-short
workload, with checksums of everything the workload consumes and produces (weights, every batch tensor, the four flows of
every pass, the recurrent states, per-window loss and gradient norm), so that a deviating iteration can be compared with a
good one line by line.

    python tools/interference_trace.py [--iters 40]
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TEF_TEST_NO_COLLECT"] = "1"
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()

import test_train_gpu as T  # noqa: E402
from taming_event_flow_amd import train  # noqa: E402

dev = torch.device("cuda:0")
EXPECTED = None


def cs(t):
    return float(t.detach().double().abs().sum().item())


def target():
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4)
    cfg["optimizer"]["lr"] = 0.0
    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=False)
    log = [("weights", sum(cs(p) for p in tr.model.parameters()))]
    global EXPECTED
    snap = [p.detach().clone() for p in tr.model.parameters()]
    if EXPECTED is None:
        EXPECTED = snap
    else:
        import time
        names = [n for n, _ in tr.model.named_parameters()]
        for n, e, g_ in zip(names, EXPECTED, snap):
            d = (e != g_).reshape(-1)
            if bool(d.any()):
                idx = d.nonzero().reshape(-1)
                print(f"   parameter {n}: {int(d.sum())} of {d.numel()} elements differ, first at {int(idx[0])}..last at {int(idx[-1])}; "
                      f"expected {e.reshape(-1)[idx[:4]].tolist()} got {g_.reshape(-1)[idx[:4]].tolist()}", flush=True)
        time.sleep(0.05)
        torch.cuda.synchronize()
        later = sum(cs(p) for p in tr.model.parameters())
        if later != log[0][1]:
            print(f"   weights kept changing after construction: {log[0][1]} -> {later}", flush=True)
    src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
    tr.reset()
    out = []
    for t in range(2 + 8):
        batch = src.next()
        log.append((f"batch{t}", tuple(round(cs(v), 6) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor))))
        if t == 2:
            tr.reset()
        arch = tr.model.arch
        arch.flow_scale = float(cfg["loss"]["flow_scaling"])
        flows = tr.model(batch["net_input"])["flow"]
        arch.flow_scale = 1.0
        log.append((f"flows{t}", tuple(round(cs(f), 5) for f in flows)))
        log.append((f"states{t}", tuple(round(cs(s), 5) for s in arch.states)))
        tr.loss_function.update(flows, batch["event_list"], batch["event_list_pol_mask"], batch["d_event_list"],
                                batch["d_event_list_pol_mask"])
        if tr.loss_function.num_passes >= 4:
            tr._backward_window()
            log.append((f"gradsum{t}", round(cs(tr.bucket.flat), 6)))
            tr.all_reduce_gradients()
            tr._apply_update()
            out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
            log.append((f"window{t}", tuple(out[-2:])))
    tr.close()
    return np.array(out), log


ref, ref_log = target()
print("reference", ref, flush=True)
bad = 0
for it in range(a.iters):
    T.test_multi_stream_window_matches_one_stream("Iterative", 1, False, True)
    got, log = target()
    if not np.allclose(got, ref, rtol=1e-6):
        bad += 1
        print(f"iteration {it}: DEVIATION {got}", flush=True)
        for (k0, v0), (k1, v1) in zip(ref_log, log):
            if v0 != v1:
                print("   first difference:", k0, v0, "->", v1, flush=True)
                break
        for (k0, v0), (k1, v1) in zip(ref_log, log):
            if v0 != v1:
                print("   differs:", k0, flush=True)
print(f"{bad} deviations in {a.iters} iterations", flush=True)
