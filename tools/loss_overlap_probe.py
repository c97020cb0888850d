#!/usr/bin/env python3
"""Do the loss kernels co-run?  The warp / chain kernels are bound by the L1 pipeline, the two scatters by LDS atomics and
vector issue: if kernels of different kinds shared the chip well, pipelining the heads (or two windows) over two streams
would raise the rate.  Timing probe: two independent BASELINE loss windows (forward + backward, as hipGraphs), replayed
A, B, A, B ... on ONE stream, and A on one stream / B on another with B's start delayed by about half a step (so that
unlike kernels meet).

    python tools/loss_overlap_probe.py [--reps 200]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import synth
    from taming_event_flow_amd.loss.flow import Iterative

    dev = torch.device("cuda:0")
    B, H, W, P, F, N = 8, 128, 128, 10, 4, 10000
    cfg = {"loader": {"batch_size": B, "resolution": [H, W]}, "data": {"passes_loss": P, "scales_loss": 1},
           "loss": {"iterative_mode": "two", "round_ts": False, "flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None}}
    wins = []
    for wi in range(2):
        rng = np.random.default_rng(wi)
        win = synth.make_window(rng, B, H, W, P, F, N, 0, sigma=2.0, kind="smooth")
        flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
        L = Iterative(cfg, dev)
        for t in range(P):
            L.update(flows[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
                     torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
        wins.append((L, [f for row in flows for f in row]))

    def step(k):
        L, fl = wins[k]
        return torch.autograd.grad(L(), fl)

    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    graphs = []
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            for _ in range(3):
                step(k)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[k]):
                keep = step(k)
            graphs.append((gr, keep))
    torch.cuda.synchronize()

    def timed(fn):
        fn(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(a.reps)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / (2 * a.reps)

    def serial(n):
        with torch.cuda.stream(streams[0]):
            for _ in range(n):
                graphs[0][0].replay()
                graphs[1][0].replay()

    def forked(n, delay):
        with torch.cuda.stream(streams[1]):
            if delay:
                torch.cuda._sleep(delay)
        for _ in range(n):
            with torch.cuda.stream(streams[0]):
                graphs[0][0].replay()
            with torch.cuda.stream(streams[1]):
                graphs[1][0].replay()

    for rnd in range(2):
        print(f"one stream            {timed(serial):.4f} ms per window")
        for delay in (0, 400_000, 800_000):
            print(f"two streams, skew {delay:7d} cycles  {timed(lambda n: forked(n, delay)):.4f} ms per window")


if __name__ == "__main__":
    main()
