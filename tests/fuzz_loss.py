#!/usr/bin/env python3
"""Randomised parity sweep of the HIP loss path against the CPU oracle (test infrastructure): random resolutions, batch,
window length, heads, temporal scales, modes, ragged / empty passes, detached lists, float coordinates, flow kinds,
smoothing weights, round_ts, border compensation on / off.  Prints every case whose error exceeds the 1e-4 bar and the worst case seen.

    python tests/fuzz_loss.py [--cases 200] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # conftest / test_loss_gpu helpers when run as a script


def sweep(cases, seed, verbose=True, big_frac=0.05, ragged_windows=False, only=None):
    """-> (number of cases over the 1e-4 bar or raising, worst relative error).  ragged_windows: passes_loss need not be a
    multiple of 2^(scales_loss - 1) — the trailing passes then belong to no window of the finer scales (drawn from a
    second generator so that the cases of the plain sweeps keep their numbers).  only: run just these case numbers of the
    sweep (the others are drawn and skipped) and say where their gradients differ, with and without the smoothing terms."""
    import __graft_entry__ as g

    g.build()
    from oracle import oracle
    from taming_event_flow_amd import synth
    from test_loss_gpu import make_cfg, run_hip
    from conftest import rel_err

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 1)
    worst, bad, t0 = (0.0, None), 0, time.time()
    for c in range(cases):
        kind = "Iterative" if rng.random() < 0.7 else "Linear"
        S = int(rng.integers(1, 4))
        P = int(rng.integers(2, 7)) * (1 << (S - 1))          # every scale divides the window
        if kind == "Iterative" and P // (1 << (S - 1)) < 2:
            P = 2 << (S - 1)
        if ragged_windows and S > 1:
            P += int(rng2.integers(0, 1 << (S - 1)))
        mode = "one" if rng.random() < 0.3 else "two"
        B, F = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        H, W = int(rng.integers(6, 70)), int(rng.integers(6, 90))
        if rng.random() < 0.1:
            H, W = int(rng.integers(150, 260)), int(rng.integers(180, 330))      # several LDS row bands
            B, F, P, S = 1, 1, min(P, 4), 1
        nmax = int(rng.integers(1, 400))
        if rng.random() < big_frac:       # long passes: several pack workgroups per sample, many rows per scatter item
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
        # round_ts on an empty list raises in the reference (:461 for the gradient list, :463 for the detached one)
        rts = rng.random() < 0.15 and min(ng) > 0 and min(nd) > 0
        comp = rng.random() >= 0.2                                   # border_compensation=False in a fifth of the cases
        meta = dict(H=H, W=W, B=B, P=P, S=S, mode=mode, spat=spat, temp=temp, round_ts=bool(rts), border_compensation=comp)
        if only is not None:
            if c not in only:
                continue
            for sp_, tp_ in ((spat, temp), (None, None)):
                m2 = dict(meta, spat=sp_, temp=tp_)
                l, gr, _ = run_hip(kind, make_cfg(m2), win, dev, border_compensation=comp)
                ow = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=S, mode=mode, round_ts=bool(rts),
                                   border_compensation=comp)
                ol, od = ow.loss(kind, sp_, tp_)
                d = np.abs(gr - od)
                print(f"case {c} spat={sp_} temp={tp_}: loss {l!r} vs {float(ol)!r}; max |grad| {np.abs(od).max():.3e}, max diff {d.max():.3e} "
                      f"at (pass, head, sample, channel, y, x) = {np.unravel_index(d.argmax(), d.shape)}")
                per = d.reshape(d.shape[0], d.shape[1], -1).max(-1) / max(np.abs(od).max(), 1e-30)
                print("   relative difference per (pass, head):", np.array2string(per, precision=1, max_line_width=200))
        try:
            l, gr, _ = run_hip(kind, make_cfg(meta), win, dev, border_compensation=comp)
            ow = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=S, mode=mode, round_ts=bool(rts),
                               border_compensation=comp)
            ol, od = ow.loss(kind, spat, temp)
        except Exception as e:                                # noqa: BLE001
            print("EXC", c, kind, meta, ng, nd, repr(e)[:200], flush=True)
            bad += 1
            continue
        el = abs(l - ol) / max(abs(ol), 1e-12)
        eg = rel_err(gr, od) if np.abs(od).max() > 0 else float(np.abs(gr).max())
        e = max(el, eg)
        if e > worst[0]:
            worst = (e, (c, kind, meta, fk, sigma))
        if e > 1e-4 or not np.isfinite(e):
            # Is the case itself ill-conditioned in fp32?  (Isolated events: d loss / d weight is a difference of two nearly
            # equal terms, and flows whose Jacobians are large amplify its rounding noise along the chain — 20 steps at
            # sigma 6 reach 1e5.)  The oracle on the same window with every flow value moved by one unit in the last place:
            # when ITS gradient moves by a comparable amount, the distance above is the problem's conditioning, not the path.
            prng = np.random.default_rng(c)
            fl1 = [[np.nextafter(f_, np.where(prng.random(f_.shape) < 0.5, -np.inf, np.inf).astype(np.float32)) for f_ in row]
                   for row in win["flows"]]
            ow1 = oracle.Window(fl1, win["ev"], win["pm"], win["dev"], win["dpm"], S=S, mode=mode, round_ts=bool(rts),
                                border_compensation=comp)
            _, od1 = ow1.loss(kind, spat, temp)
            own = rel_err(od1, od) if np.abs(od).max() > 0 else float(np.abs(od1).max())
            # ... and with the events of every pass in another ORDER: the loss does not depend on it, the fp32 sums of the
            # oracle (the reference's accumulation order) do — passes of thousands of events with nearly equal time stamps
            # leave tau - A / (C + eps) as the difference of two nearly equal sums (the HIP path accumulates in fixed point)
            def shuffled(lst, msk):
                out_l, out_m = [], []
                for l_, m_ in zip(lst, msk):
                    o_ = prng.permutation(l_.shape[1])
                    out_l.append(np.ascontiguousarray(l_[:, o_])), out_m.append(np.ascontiguousarray(m_[:, o_]))
                return out_l, out_m
            ev2, pm2 = shuffled(win["ev"], win["pm"])
            dv2, dpm2 = shuffled(win["dev"], win["dpm"])
            ow2 = oracle.Window(win["flows"], ev2, pm2, dv2, dpm2, S=S, mode=mode, round_ts=bool(rts), border_compensation=comp)
            _, od2 = ow2.loss(kind, spat, temp)
            own2 = rel_err(od2, od) if np.abs(od).max() > 0 else float(np.abs(od2).max())
            # ... and with every event's coordinates moved by one unit in the last place (zero flows have no last place to
            # move; an event 7e-6 px off a pixel row leaves that row a weight one ulp changes by 14 %)
            def nudged(lst):
                out_l = []
                for l_ in lst:
                    l2 = np.array(l_, copy=True)
                    # (with flows that are exactly zero an event ON a pixel stays on it at every reference time: that tie is a
                    # kink of the loss, which the path must reproduce, not noise — those coordinates stay.  With any other
                    # flows the warped positions of such an event are ordinary numbers, and one that lands 1e-6 px from a
                    # pixel row — a flow component of 1e-5 does it — is exactly the small-weight case)
                    keep_ties = sigma == 0.0
                    for col in (1, 2):
                        v_ = l2[:, :, col]
                        to = np.where(prng.random(v_.shape) < 0.5, -np.inf, np.inf).astype(np.float32)
                        l2[:, :, col] = np.where(keep_ties & (v_ == np.floor(v_)), v_, np.nextafter(v_, to))
                    out_l.append(l2)
                return out_l
            ow3 = oracle.Window(win["flows"], nudged(win["ev"]), win["pm"], nudged(win["dev"]), win["dpm"], S=S, mode=mode,
                                round_ts=bool(rts), border_compensation=comp)
            _, od3 = ow3.loss(kind, spat, temp)
            own3 = rel_err(od3, od) if np.abs(od).max() > 0 else float(np.abs(od3).max())
            if np.isfinite(e) and el <= 1e-4 and max(own, own2, own3) > 0.25 * eg:
                print(f"ill-conditioned case {c}: {kind} {meta} flows={fk}/{sigma} grad rel {eg:.2e}; the oracle moves by {own:.2e} "
                      f"under one-ulp noise on the flows, by {own2:.2e} with its events in another order, by {own3:.2e} under one-ulp "
                      "noise on the event coordinates", flush=True)
                continue
            bad += 1
            print(f"FAIL case {c}: {kind} {meta} ng={ng} nd={nd} flows={fk}/{sigma} loss {l} vs {float(ol)} (rel {el:.2e}) "
                  f"grad rel {eg:.2e} (oracle under one-ulp noise: {own:.2e}, with its events in another order: {own2:.2e}, under one-ulp noise on the coordinates: {own3:.2e})", flush=True)
    if verbose:
        print(f"{cases} cases in {time.time() - t0:.0f} s, {bad} over the 1e-4 bar; worst {worst[0]:.2e} at {worst[1]}")
    return bad, worst[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big-frac", type=float, default=0.05, help="fraction of cases with 4096..9000 events per pass")
    ap.add_argument("--ragged-windows", action="store_true", help="passes_loss not a multiple of 2^(scales_loss - 1)")
    ap.add_argument("--only", default=None, help="comma-separated case numbers: run these alone, with a report of where the gradients differ")
    a = ap.parse_args()
    only = None if a.only is None else {int(x_) for x_ in a.only.split(",")}
    bad, _ = sweep(a.cases, a.seed, big_frac=a.big_frac, ragged_windows=a.ragged_windows, only=only)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
