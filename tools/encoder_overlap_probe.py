#!/usr/bin/env python3
"""Probe for DESIGN section 9f ("what a halo launch costs"): how much of the encoder side's sequential-launch cost would
a second, independent chain of launches hide?  Two networks' encoder halves (10 passes each, forward only), first one after
the other on one stream, then side by side on two streams — the shape a level-pipelined encoder walker would have.

    python tools/encoder_overlap_probe.py [CHAINS]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train
    from taming_event_flow_amd.models.model import RecEVFlowNet

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nets = []
    for k in range(N):
        torch.manual_seed(k)
        m = RecEVFlowNet(cfg["model"].copy() if isinstance(cfg["model"], dict) else cfg["model"]).to(dev)
        nets.append(m)
    P, B = 10, 8
    xs = [torch.rand(B, 2, 128, 128, device=dev) for _ in range(P)]
    streams = [torch.cuda.Stream() for _ in range(N)]

    def chain(m, n=P):
        eng = m.arch.engine
        states = [None] * eng.plan.levels
        for t in range(n):
            _, states, _ = eng.forward(xs[t], states, keep=False, part=1)

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    def one_after_the_other():
        with torch.no_grad():
            for m in nets:
                chain(m)

    def side_by_side():
        with torch.no_grad():
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            # interleave the host's launches pass by pass so that both chains are fed
            engs = [m.arch.engine for m in nets]
            states = [[None] * engs[0].plan.levels for _ in nets]
            for t in range(P):
                for k in range(N):
                    with torch.cuda.stream(streams[k]):
                        _, states[k], _ = engs[k].forward(xs[t], states[k], keep=False, part=1)
            for s in streams:
                cur.wait_stream(s)

    # as hipGraphs (the host out of the picture)
    def graph_of(fn):
        gph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(gph):
            fn()
        return gph

    a = timed(one_after_the_other)
    b = timed(side_by_side)
    print(f"eager: {N} encoder chains of {P} passes one after the other {a:.3f} ms, side by side on {N} streams {b:.3f} ms ({b / a:.3f})")
    ga, gb = graph_of(one_after_the_other), graph_of(side_by_side)
    a = timed(ga.replay)
    b = timed(gb.replay)
    print(f"hipGraph: one after the other {a:.3f} ms, side by side {b:.3f} ms ({b / a:.3f})")


if __name__ == "__main__":
    main()
