#!/usr/bin/env python3
"""Per-layer timing of the RecEVFlowNet convolutions on the HIP implicit-GEMM kernels (tef_conv.hip): forward and
backward (input + weight gradient) of every distinct shape at B=8, 128x128, with the MFMA rate each reaches.

    python tools/conv_bench.py [--reps 20] [--batch 8] [--res 128]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def layers(B, R):
    """(name, C0, C1, N, k, stride, input resolution (h, w), calls per forward pass)"""
    L = []
    c_in, res = 2, Res(*R)
    for i, c in enumerate((64, 128, 256, 512)):
        L.append((f"enc{i}.head {c_in}->{c} s2", c_in, 0, c, 3, 2, res, 1))
        res = res // 2
        L.append((f"enc{i}.gru gates {c}+{c}->{2 * c}", c, c, 2 * c, 3, 1, res, 1))
        L.append((f"enc{i}.gru out {c}+{c}->{c}", c, c, c, 3, 1, res, 1))
        c_in = c
    L.append(("resblock 512->512", 512, 0, 512, 3, 1, res, 4))
    for i, (ci, co) in enumerate(((512, 256), (258, 128), (130, 64), (66, 32))):
        res = res * 2
        L.append((f"dec{i} {ci}->{co}", ci, 0, co, 3, 1, res, 1))
        L.append((f"pred{i} {co}->2 1x1", co, 0, 2, 1, 1, res, 1))
    return L


class Res:
    """input resolution of a layer; halves / doubles along the UNet"""

    def __init__(self, h, w):
        self.h, self.w = h, w

    def __floordiv__(self, s):
        return Res(self.h // s, self.w // s)

    def __mul__(self, s):
        return Res(self.h * s, self.w * s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--width", type=int, default=0, help="rectangular inputs: --res is the height (eval shape: 480 x 640)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.models import submodules as sm

    dev = torch.device("cuda:0")
    tot_f = tot_b = fl_tot = 0.0
    print(f"{'layer':44s} {'GFLOP':>7s} {'fwd ms':>8s} {'TF/s':>6s} {'bwd ms':>8s} {'TF/s':>6s}")
    for name, c0, c1, n, k, s, res, calls in layers(a.batch, (a.res, a.width or a.res)):
        name = f"{name} @{res.h}x{res.w}"
        if a.only and a.only not in name:
            continue
        x0 = torch.randn(a.batch, c0, res.h, res.w, device=dev, requires_grad=True)
        x1 = torch.randn(a.batch, c1, res.h, res.w, device=dev, requires_grad=True) if c1 else None
        w = (torch.randn(n, c0 + c1, k, k, device=dev) * 0.05).requires_grad_()
        b = torch.zeros(n, device=dev, requires_grad=True)
        pk = sm.PackedWeights()
        ro = res // s
        gflop = 2.0 * a.batch * ro.h * ro.w * n * (c0 + c1) * k * k / 1e9

        def fwd():
            return sm.conv2d(pk, x0, w, b, stride=s, act="relu", x1=x1)

        out = fwd()
        dout = torch.randn_like(out)
        ins = [x0, w, b] + ([x1] if x1 is not None else [])
        torch.autograd.grad(out, ins, dout, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(a.reps):
            fwd()
        e1.record()
        for _ in range(a.reps):
            torch.autograd.grad(out, ins, dout, retain_graph=True)
        e2.record()
        torch.cuda.synchronize()
        tf, tb = e0.elapsed_time(e1) / a.reps, e1.elapsed_time(e2) / a.reps
        print(f"{name:44s} {gflop:7.2f} {tf:8.3f} {gflop / tf:6.1f} {tb:8.3f} {2 * gflop / tb:6.1f}   x{calls}")
        tot_f += tf * calls
        tot_b += tb * calls
        fl_tot += gflop * calls
    print(f"{'per pass':44s} {fl_tot:7.2f} {tot_f:8.3f} {fl_tot / tot_f:6.1f} {tot_b:8.3f} {2 * fl_tot / tot_b:6.1f}")


if __name__ == "__main__":
    main()
