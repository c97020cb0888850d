#!/usr/bin/env python3
"""Diagnosis helper for a failing case of tests/fuzz_loss.py: re-draw the sweep up to the case, then look for the smallest
set of events that still shows the gradient mismatch between the HIP path and the CPU oracle (test infrastructure).

    python tools/fuzz_case_bisect.py --seed 700 --case 1197
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def draw(seed, case, ragged_windows=False, big_frac=0.05):
    """The generator of tests/fuzz_loss.py::sweep, drawn up to `case` (kept in step with it by hand)."""
    from taming_event_flow_amd import synth

    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 1)
    for c in range(case + 1):
        kind = "Iterative" if rng.random() < 0.7 else "Linear"
        S = int(rng.integers(1, 4))
        P = int(rng.integers(2, 7)) * (1 << (S - 1))
        if kind == "Iterative" and P // (1 << (S - 1)) < 2:
            P = 2 << (S - 1)
        if ragged_windows and S > 1:
            P += int(rng2.integers(0, 1 << (S - 1)))
        mode = "one" if rng.random() < 0.3 else "two"
        B, F = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        H, W = int(rng.integers(6, 70)), int(rng.integers(6, 90))
        if rng.random() < 0.1:
            H, W = int(rng.integers(150, 260)), int(rng.integers(180, 330))
            B, F, P, S = 1, 1, min(P, 4), 1
        nmax = int(rng.integers(1, 400))
        if rng.random() < big_frac:
            nmax = int(rng.integers(4096, 9000))
            B, F, P, S = 1, int(rng.integers(1, 3)), min(P, 4), 1
        ng = [int(rng.integers(0, nmax + 1)) if rng.random() < 0.5 else nmax for _ in range(P)]
        if sum(ng) == 0:
            ng[0] = 5
        nd = [int(rng.integers(0, nmax // 2 + 1)) if rng.random() < 0.5 else 0 for _ in range(P)]
        sigma = float(rng.choice([0.0, 0.5, 2.0, 6.0]))
        fk = "smooth" if rng.random() < 0.7 else "iid"
        win = synth.make_window(rng, B, H, W, P, F, ng, nd, sigma=sigma, kind=fk, ragged=rng.random() < 0.5,
                                integer_coords=rng.random() < 0.7)
        spat = float(rng.choice([0.001, 0.1])) if rng.random() < 0.25 else None
        temp = float(rng.choice([0.001, 0.1])) if rng.random() < 0.25 else None
        rts = rng.random() < 0.15 and min(ng) > 0 and min(nd) > 0
        comp = rng.random() >= 0.2
    meta = dict(H=H, W=W, B=B, P=P, S=S, mode=mode, spat=spat, temp=temp, round_ts=bool(rts), border_compensation=comp)
    return kind, meta, win


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--ragged-windows", action="store_true")
    ap.add_argument("--no-smooth", action="store_true", help="drop the smoothing terms")
    ap.add_argument("--frac", type=float, default=0.1, help="a subset is kept while it shows this share of the whole case's mismatch")
    ap.add_argument("--keep", default=None, help="events to keep instead of searching, e.g. pm:4:0:41,dpm:3:0:7")
    ap.add_argument("--set", default=None, help="overrides of the case's settings, e.g. S=1,mode=two")
    a = ap.parse_args()
    import __graft_entry__ as g

    g.build()
    from oracle import oracle
    from test_loss_gpu import make_cfg, run_hip

    dev = torch.device("cuda:0")
    kind, meta, win = draw(a.seed, a.case, a.ragged_windows)
    if a.no_smooth:
        meta = dict(meta, spat=None, temp=None)
    for kv in (a.set.split(",") if a.set else []):
        k_, v_ = kv.split("=")
        meta[k_] = (v_ == "True") if isinstance(meta[k_], bool) else type(meta[k_])(v_)
    print(kind, meta)

    def err(w, show=0):
        l, gr, _ = run_hip(kind, make_cfg(meta), {k: [np.array(x_, copy=True) for x_ in v] if k != "flows" else v for k, v in w.items()},
                           dev, border_compensation=meta["border_compensation"])
        ow = oracle.Window(w["flows"], [np.array(x_, copy=True) for x_ in w["ev"]], w["pm"], [np.array(x_, copy=True) for x_ in w["dev"]],
                           w["dpm"], S=meta["S"], mode=meta["mode"], round_ts=meta["round_ts"],
                           border_compensation=meta["border_compensation"])
        ol, od = ow.loss(kind, meta["spat"], meta["temp"])
        d = np.abs(gr - od)
        if show:
            for idx in np.argsort(d.ravel())[::-1][:show]:
                ii = np.unravel_index(idx, d.shape)
                print("   ", tuple(int(v) for v in ii), "hip", repr(float(gr[ii])), "oracle", repr(float(od[ii])))
            per = d.reshape(d.shape[0], -1).max(-1) / max(np.abs(od).max(), 1e-30)
            print("    per pass:", np.array2string(per, precision=1, max_line_width=250))
            own = d.reshape(d.shape[0], -1).max(-1) / np.maximum(np.abs(od).reshape(d.shape[0], -1).max(-1), 1e-30)
            print("    per pass, relative to the pass's own largest gradient:", np.array2string(own, precision=1, max_line_width=250))
        return float(d.max() / max(np.abs(od).max(), 1e-30)), l, float(ol), np.unravel_index(d.argmax(), d.shape)

    print("whole case:", err(win))
    # active events: (list, pass, sample, index)
    act = [(k, t, b, j) for k, m in (("pm", "pm"), ("dpm", "dpm")) for t in range(meta["P"]) for b in range(meta["B"])
           for j in range(win[m][t].shape[1]) if win[m][t][b, j].any()]
    print(len(act), "active events")

    def masked(keep):
        w = dict(win)
        w["pm"] = [np.zeros_like(x_) for x_ in win["pm"]]
        w["dpm"] = [np.zeros_like(x_) for x_ in win["dpm"]]
        for k, t, b, j in keep:
            w[k][t][b, j] = win[k][t][b, j]
        return w

    if a.keep:
        keep = [(k_, int(t_), int(b_), int(j_)) for k_, t_, b_, j_ in (e_.split(":") for e_ in a.keep.split(","))]
        print("kept events:", err(masked(keep), show=12))
        return
    keep = list(act)
    base = err(masked(keep))[0]
    print("all re-masked:", base)
    # greedy halving: drop chunks while the mismatch stays above a tenth of the original
    chunk = max(1, len(keep) // 2)
    while chunk >= 1:
        i = 0
        while i < len(keep) and len(keep) > 1:
            trial = keep[:i] + keep[i + chunk:]
            if trial and err(masked(trial))[0] > a.frac * base:
                keep = trial
            else:
                i += chunk
        chunk //= 2
    print(len(keep), "events left:", keep[:40], "..." if len(keep) > 40 else "")
    print("their mismatch:", err(masked(keep), show=8))
    for k, t, b, j in keep:
        lst = win["ev" if k == "pm" else "dev"][t][b, j]
        print("  ", k, t, b, j, "event (ts, y, x, p) =", [repr(float(v)) for v in lst], "mask", win[k][t][b, j])


if __name__ == "__main__":
    main()
