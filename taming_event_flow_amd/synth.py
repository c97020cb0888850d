"""Synthetic DSEC-shaped inputs (numpy only, deterministic across machines).

Restates the *shape* of what the reference data loader hands to the training
loop (reference dataloader/h5.py:413-431 batch dict, dataloader/base.py:252-278
list encoding + polarity mask, base.py:392-434 zero-padded collate), without any
HDF5 / OpenCV dependency.  Used by tests, bench.py and the golden generator.

Event list layout is the reference's AoS ``[B, N, 4] = (ts, y, x, p)`` with
``ts`` normalised to [0, 1] inside the pass, integer pixel coordinates stored as
fp32, ``p`` in {-1, +1}; polarity mask ``[B, N, 2] = (p > 0, p < 0)``.
Padding rows (collate) are all-zero with a (0, 0) mask.
"""

import numpy as np


def make_event_pass(rng, B, N, H, W, n_valid=None, integer_coords=True):
    """One pass worth of events for B samples.

    n_valid: optional per-sample number of real events (rest is zero padding,
    as produced by the reference's custom_collate, base.py:414-421).
    Returns (event_list [B,N,4] f32, pol_mask [B,N,2] f32).
    """
    ev = np.zeros((B, N, 4), np.float32)
    pm = np.zeros((B, N, 2), np.float32)
    for b in range(B):
        n = N if n_valid is None else int(n_valid[b])
        if n == 0:
            continue
        ts = np.sort(rng.random(n).astype(np.float32))
        if n > 1:
            # the reference normalises the slice to [0, 1] (base.py:168-169)
            ts = (ts - ts[0]) / max(ts[-1] - ts[0], 1e-9)
        if integer_coords:
            y = rng.integers(0, H, n).astype(np.float32)
            x = rng.integers(0, W, n).astype(np.float32)
        else:
            y = (rng.random(n) * (H - 1)).astype(np.float32)
            x = (rng.random(n) * (W - 1)).astype(np.float32)
        p = np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
        ev[b, :n, 0] = ts
        ev[b, :n, 1] = y
        ev[b, :n, 2] = x
        ev[b, :n, 3] = p
        pm[b, :n, 0] = p > 0
        pm[b, :n, 1] = p < 0
    return ev, pm


def _bilinear_upsample(grid, H, W):
    """grid [..., gh, gw] -> [..., H, W] (align_corners=True style, numpy)."""
    gh, gw = grid.shape[-2:]
    ys = np.linspace(0, gh - 1, H)
    xs = np.linspace(0, gw - 1, W)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    g = grid
    a = g[..., y0[:, None], x0[None, :]]
    b = g[..., y0[:, None], x0[None, :] + 1]
    c = g[..., y0[:, None] + 1, x0[None, :]]
    d = g[..., y0[:, None] + 1, x0[None, :] + 1]
    # (C-contiguous like a network output: fancy indexing leaves a permuted memory layout behind, and a flow map in that
    # layout makes update() take its converting, call-by-call path)
    return np.ascontiguousarray((a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx).astype(np.float32))


def make_flow(rng, B, H, W, sigma=2.0, kind="smooth", grid=8):
    """One flow map [B, 2, H, W] (channel 0 = x, channel 1 = y; px / pass).

    kind="smooth": bilinear up-sampling of a grid x grid field of N(0, sigma^2)
    (most events stay in bounds); kind="iid": i.i.d. N(0, sigma^2) per pixel
    (stress set, many purged events); kind="zero": exact zeros (tie cases).
    """
    if kind == "zero":
        return np.zeros((B, 2, H, W), np.float32)
    if kind == "iid":
        return (rng.standard_normal((B, 2, H, W)) * sigma).astype(np.float32)
    g = rng.standard_normal((B, 2, grid, grid)) * sigma
    return _bilinear_upsample(g, H, W)


def make_window(rng, B, H, W, P, F, n_grad, n_det=0, sigma=2.0, kind="smooth", ragged=False,
                integer_coords=True):
    """A full loss window: P passes of events + F flow heads per pass.

    n_grad / n_det: int or list of P ints (events per pass, per sample).
    Returns dict with lists over passes: flows[t][i] [B,2,H,W], ev[t], pm[t], dev[t], dpm[t].
    """
    ng = [n_grad] * P if np.isscalar(n_grad) else list(n_grad)
    nd = [n_det] * P if np.isscalar(n_det) else list(n_det)
    out = {"flows": [], "ev": [], "pm": [], "dev": [], "dpm": []}
    for t in range(P):
        out["flows"].append([make_flow(rng, B, H, W, sigma, kind) for _ in range(F)])
        nv = None
        if ragged and ng[t] > 4:
            nv = rng.integers(ng[t] // 2, ng[t] + 1, B)
            nv[rng.integers(0, B)] = ng[t]
        e, m = make_event_pass(rng, B, ng[t], H, W, nv, integer_coords)
        out["ev"].append(e)
        out["pm"].append(m)
        e, m = make_event_pass(rng, B, nd[t], H, W, None, integer_coords)
        out["dev"].append(e)
        out["dpm"].append(m)
    return out


def make_eval_window(seed, H, W, passes, N, sigma=2.0):
    """Inputs of one evaluation window for the validation metrics (loss/flow_val.py; batch 1 as eval_flow.py:30 hard-wires):
    per pass the events + polarity mask, the flow map that counts, a low-resolution decoy (flow_list[-1] is the one the
    reference uses) and the event mask image -> list of dicts."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(passes):
        ev, pm = make_event_pass(rng, 1, N, H, W)
        flow = make_flow(rng, 1, H, W, sigma=sigma, grid=12)
        low = make_flow(rng, 1, H, W, sigma=sigma, grid=12)
        mask = np.zeros((1, 1, H, W), np.float32)
        mask[0, 0, ev[0, :, 1].astype(int), ev[0, :, 2].astype(int)] = 1.0
        out.append(dict(ev=ev, pm=pm, flow=flow, low=low, mask=mask))
    return out


def make_model_weights(shapes, seed):
    """Deterministic (numpy PCG64) parameter values for a list of (name, shape): weights ~ N(0, 1/fan_in),
    biases ~ N(0, 0.05^2).  Used instead of storing 31 M parameters in the golden fixtures."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) == 1:
            out[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    return out
