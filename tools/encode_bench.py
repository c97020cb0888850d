#!/usr/bin/env python3
"""Timing of the count / voxel encoder alone (back-to-back launches): python tools/encode_bench.py [--events 100000] [--res 480 640]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=100000)
    ap.add_argument("--res", type=int, nargs=2, default=[480, 640])
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.dataloader import encodings as enc

    H, W = a.res
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(0)
    N, B = a.events, a.batch
    ev = torch.stack([torch.rand(B, N, generator=gen), torch.randint(0, H, (B, N), generator=gen).float(),
                      torch.randint(0, W, (B, N), generator=gen).float(),
                      torch.randint(0, 2, (B, N), generator=gen).float() * 2 - 1], dim=2).to(dev)
    for name, fn in (("channels", lambda: enc.event_list_to_channels(ev, (H, W))),
                     ("voxel5", lambda: enc.event_list_to_voxel(ev, 5, (H, W)))):
        try:
            fn()
        except Exception as e:      # helper signature differs: report and stop
            print(name, "skipped:", e)
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, B={B} N={N} {H}x{W}")


if __name__ == "__main__":
    main()
