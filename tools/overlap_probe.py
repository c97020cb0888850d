#!/usr/bin/env python3
"""Would two streams help the network?  The encoders of pass t + 1 do not depend on the residual blocks / decoders of
pass t, so the two halves of consecutive passes could run side by side and fill each other's launch ramps and tails.
Timing probe only: the forward convolutions of the encoder half and of the decoder half of one pass (B = 8, 128 x 128), as
one hipGraph on ONE stream (A then B) and as a forked hipGraph (A on one stream, B on another, joined).  The halves share
the convolution workspace here, so the numbers mean something and the outputs do not.

    python tools/overlap_probe.py [--reps 30]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.conv_bench import layers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.models import submodules as sm

    dev = torch.device("cuda:0")
    enc, dec = [], []
    for name, c0, c1, n, k, s, res, calls in layers(a.batch, (128, 128)):
        if "pred" in name:
            continue
        x0 = torch.randn(a.batch, c0, res.h, res.w, device=dev)
        x1 = torch.randn(a.batch, c1, res.h, res.w, device=dev) if c1 else None
        w = torch.randn(n, c0 + c1, k, k, device=dev) * 0.05
        b = torch.zeros(n, device=dev)
        pk = sm.PackedWeights()
        fn = (lambda pk=pk, x0=x0, w=w, b=b, s=s, x1=x1: sm.conv2d(pk, x0, w, b, stride=s, act="relu", x1=x1))
        (enc if name.startswith("enc") else dec).extend([fn] * calls)
    with torch.no_grad():
        for f in enc + dec:
            f()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

        def run_serial():
            for f in enc:
                f()
            for f in dec:
                f()

        def run_forked():
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                for f in dec:
                    f()
            for f in enc:
                f()
            cur.wait_stream(s2)

        graphs = {}
        for name, fn in (("one stream", run_serial), ("two streams", run_forked), ("encoder half", lambda: [f() for f in enc]),
                         ("decoder half", lambda: [f() for f in dec])):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s1):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s1):
                    fn()
            graphs[name] = gr
        for rnd in range(2):
            for name, gr in graphs.items():
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                print(f"{name:14s} {e0.elapsed_time(e1) / a.reps:8.3f} ms")


if __name__ == "__main__":
    main()
