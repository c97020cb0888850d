#!/usr/bin/env python3
"""Golden vectors for the loader stage, recorded by RUNNING the reference's own code (build container
only): BaseDataLoader.event_formatting / augment_events / create_list_encoding / create_polarity_mask /
split_event_list / custom_collate (dataloader/base.py:147-222, 252-278, 348-377, 392-434).

The reference module cannot be imported as a module here: its first lines import OpenCV (base.py:3), which this image
does not have, although none of the functions above touch it (only the rectification remaps do).  This script therefore
parses /root/reference/dataloader/base.py with `ast`, takes the BaseDataLoader class definition AS WRITTEN and executes
that definition with the names its methods really use (torch, numpy, random, abstractmethod and the reference's own
dataloader.encodings, which imports normally).  No library is substituted, no reference text is stored: the fixture
holds inputs, the random draws and outputs.
"""
import ast
import os
import random
import sys
from abc import abstractmethod

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_loader_class():
    sys.path.insert(0, REF)
    from dataloader.encodings import events_to_channels, events_to_voxel      # the reference's (imports torch only)

    with open(os.path.join(REF, "dataloader", "base.py")) as f:
        tree = ast.parse(f.read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BaseDataLoader"]
    assert len(cls) == 1
    env = {"torch": torch, "np": np, "random": random, "abstractmethod": abstractmethod,
           "events_to_channels": events_to_channels, "events_to_voxel": events_to_voxel}
    exec(compile(ast.Module(body=cls, type_ignores=[]), "reference:dataloader/base.py", "exec"), env)
    return env["BaseDataLoader"]


def main():
    Base = reference_loader_class()

    class Loader(Base):              # the abstract methods are never called
        def __getitem__(self, index):
            raise NotImplementedError

        def get_events(self, history):
            raise NotImplementedError

    H, W, B, G = 20, 26, 4, 30
    aug = ["Horizontal", "Vertical", "Polarity"]
    np.random.seed(0)
    ld = Loader({"loader": {"device": "cpu", "resolution": [H, W], "batch_size": B, "augment": aug,
                            "augment_prob": [0.5, 0.5, 0.5]}})
    # fix the augmentation state explicitly (all eight combinations appear over the samples / cases)
    combos = [(False, False, False), (True, False, True), (False, True, False), (True, True, True)]
    for b, c in enumerate(combos):
        for m, v in zip(aug, c):
            ld.batch_augmentation[m][b] = v
    rng = np.random.default_rng(7)
    counts = [45, 8, 30, 61]         # > G (split), <= 10 (emptied by the caller, h5.py:340-345), == G (no split), > G
    out = {"H": H, "W": W, "B": B, "G": G, "counts": np.array(counts),
           "flags": np.array([sum(bit for bit, on in zip((1, 2, 4), c) if on) for c in combos])}
    items = []
    for b, n in enumerate(counts):
        xs = rng.integers(0, W, n).astype(np.int64)
        ys = rng.integers(0, H, n).astype(np.int64)
        ts = np.sort(rng.random(n) * 0.01 + 3.0 + b)             # raw timestamps (seconds, float64); fp32 rounding matters
        ps = rng.integers(0, 2, n).astype(np.int64)
        out.update({f"xs{b}": xs, f"ys{b}": ys, f"ts{b}": ts, f"ps{b}": ps})
        if n <= 10:                                              # h5.py:340-345
            xs = ys = ts = ps = np.empty([0])
        fx, fy, ft, fp = ld.event_formatting(xs, ys, ts, ps)
        out.update({f"fmt_ts{b}": ft.numpy().copy(), f"fmt_ps{b}": fp.numpy().copy()})
        ax, ay, ap, _, _ = ld.augment_events(fx, fy, fp, None, None, b)
        ev = ld.create_list_encoding(ax, ay, ft, ap)
        mk = ld.create_polarity_mask(ap)
        out.update({f"list{b}": ev.numpy().copy(), f"mask{b}": mk.numpy().copy()})
        torch.manual_seed(100 + b)
        if ev.shape[1] > G:          # the same draw split_event_list makes (base.py:363-366), recorded for the replay
            probs = torch.ones(ev.shape[1], dtype=torch.float32) / ev.shape[1]
            out[f"sampled{b}"] = probs.multinomial(G, replacement=False).numpy().copy()
        torch.manual_seed(100 + b)
        g, gm, d, dm = ld.split_event_list(ev, mk, G)
        out.update({f"g{b}": g.numpy().copy(), f"gm{b}": gm.numpy().copy(), f"d{b}": d.numpy().copy(),
                    f"dm{b}": dm.numpy().copy()})
        items.append({"event_list": g, "event_list_pol_mask": gm, "d_event_list": d, "d_event_list_pol_mask": dm,
                      "net_input": torch.full((2, 3, 3), float(b)), "gtflow": None})
    col = ld.custom_collate(items)
    assert col["gtflow"] is None
    for k in ("event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask", "net_input"):
        out["col_" + k] = col[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **out)
    print("loader.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and k.startswith("col_")})


if __name__ == "__main__":
    main()
