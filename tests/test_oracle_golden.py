"""Pin the CPU oracle (oracle/tef_oracle.c) against golden vectors recorded from the reference.

Reference under test (by recorded outputs): loss/flow.py Iterative/Linear, utils/iwe.py primitives,
dataloader/encodings.py.  Tolerances: 1e-6 relative in max-norm, 1e-5 element by element (the north-star bar for the HIP path is 1e-4).
"""
import os

import numpy as np
import pytest

from conftest import (BENCH_WINDOW_CASES, GOLDEN, FULL_RES_CASES, ITERATIVE_CASES, LINEAR_CASES, check_bench_window,
                      elementwise_error, load_bench_window, load_case, rel_err)
from oracle import oracle

TOL = 1e-6


@pytest.mark.parametrize("name", ITERATIVE_CASES + LINEAR_CASES + FULL_RES_CASES)
def test_loss_cases(name):
    meta, win, loss, dflows = load_case(name)
    w = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=meta["S"], mode=meta["mode"],
                      round_ts=meta["round_ts"], loss_scaling=meta.get("loss_scaling", True),
                      border_compensation=meta.get("border_compensation", True))
    l, d = w.loss(meta["kind"], meta["spat"], meta["temp"])
    assert abs(l - loss) <= TOL * abs(loss), (l, loss)
    assert d.shape == dflows.shape
    assert rel_err(d, dflows) <= TOL
    # per-map check so that a small map cannot hide behind a large one
    for t in range(meta["P"]):
        for i in range(meta["F"]):
            if np.abs(dflows[t, i]).max() > 0:
                assert rel_err(d[t, i], dflows[t, i]) <= 1e-5, (t, i)
    # element by element, the plain form at a tenth of the HIP path's bar: |oracle - reference| <= 1e-5 |reference| +
    # 1e-7 max|reference| at every pixel (measured: <= 0.25 of that)
    assert elementwise_error(d, dflows, rtol=1e-5, floor=1e-7)[0] <= 1.0

@pytest.mark.parametrize("name", BENCH_WINDOW_CASES)
def test_bench_windows(name):
    """The windows bench.py times, at full size (B = 8, F = 4, P = 10, 10 000 events per pass and sample): the oracle
    against the reference's recorded loss and gradient — the oracle is what bench.py's `cpu_baseline` leg runs."""
    meta, win, gold = load_bench_window(name)
    w = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=1, mode="two")
    l, d = w.loss("Iterative", None, None)
    print(name, check_bench_window(meta, gold, l, d, TOL))


def test_primitives():
    z = np.load(os.path.join(GOLDEN, "primitives.npz"))
    H, W = int(z["H"]), int(z["W"])
    out, dfx, dfy, dloc = oracle.get_event_flow(z["gef_fx"], z["gef_fy"], z["gef_loc"], z["gef_w"])
    assert rel_err(out, z["gef_out"]) < 1e-6
    assert rel_err(dfx, z["gef_dfx"]) < 1e-6
    assert rel_err(dfy, z["gef_dfy"]) < 1e-6
    assert rel_err(dloc, z["gef_dloc"]) < 1e-5
    idx, w, dpos = oracle.get_interpolation(z["gef_loc"], H, W, z["gi_r"])
    np.testing.assert_array_equal(idx, z["gi_idx"])
    assert rel_err(w, z["gi_w"]) < 1e-6
    assert rel_err(dpos, z["gi_dpos"]) < 1e-6
    iwe, iwe_ts = oracle.iwe_formatting(z["gef_loc"], z["purge_pm"], z["fmt_ts"][..., 0], H, W, 2.0, 2.0)
    assert rel_err(iwe, z["fmt_iwe"]) < 1e-6
    assert rel_err(iwe_ts, z["fmt_iwe_ts"]) < 1e-6
    fl = oracle.focus_loss(iwe, iwe_ts)
    assert abs(fl - z["focus"]) < 1e-6 * abs(z["focus"])


def test_encodings():
    z = np.load(os.path.join(GOLDEN, "encodings.npz"))
    H, W = int(z["H"]), int(z["W"])
    np.testing.assert_array_equal(oracle.events_to_channels(z["xs"], z["ys"], z["ps"], H, W), z["cnt"])
    np.testing.assert_array_equal(oracle.events_to_image(z["xs"], z["ys"], z["ps"], H, W), z["image"])
    for bins in (2, 5, 9):
        v = oracle.events_to_voxel(z["xs"], z["ys"], z["ts"], z["ps"], bins, H, W)
        assert rel_err(v, z[f"voxel{bins}"]) < 1e-6


def test_primitives_chain_gradients():
    """A focus loss assembled from the primitives one by one (tests/golden/make_golden.py --primitives: lookup ->
    event_propagation -> purge_unfeasible -> corners -> per-polarity images -> loss/flow.py:112-129) — the oracle's
    gradient primitives chained by hand against the gradients the reference's autograd recorded: d loss / d both flow maps
    and d loss / d event locations."""
    z = np.load(os.path.join(GOLDEN, "primitives.npz"))
    H, W = int(z["H"]), int(z["W"])
    fx, fy, loc, ts, pm = z["chain_fx"], z["chain_fy"], z["chain_loc"], z["chain_ts"], z["chain_pm"]
    B, N = loc.shape[:2]
    flow = oracle.get_event_flow(fx, fy, loc)
    dt = (1.0 - ts).astype(np.float32)                                        # event_propagation to tref = 1
    warped = (loc + dt * flow).astype(np.float32)
    inside = ((warped[..., 0:1] >= 0) & (warped[..., 0:1] <= H - 1.0) & (warped[..., 1:2] >= 0) & (warped[..., 1:2] <= W - 1.0))
    inside = inside.astype(np.float32)
    warped, wpm = warped * inside, pm * inside                                 # purge_unfeasible
    idx, w = oracle.get_interpolation(warped, H, W)
    ii = idx[:, :, 0].astype(np.int64)
    tau = np.concatenate([ts] * 4, 1)                                          # 1 - |1 - ts| / 1
    C = np.zeros((B, 2, H * W), np.float64)
    T = np.zeros((B, 2, H * W), np.float64)
    for c in range(2):
        m4 = np.concatenate([wpm[:, :, c:c + 1]] * 4, 1)
        for b in range(B):
            np.add.at(C[b, c], ii[b], (w[b, :, 0] * m4[b, :, 0]).astype(np.float64))
            np.add.at(T[b, c], ii[b], (w[b, :, 0] * tau[b, :, 0] * m4[b, :, 0]).astype(np.float64))
    assert rel_err(C.reshape(B, 2, H, W), z["chain_iwe"]) < 1e-5 and rel_err(T.reshape(B, 2, H, W), z["chain_iwe_ts"]) < 1e-5
    A = T / (C + 1e-9)
    n = ((C[:, 0] + C[:, 1]) != 0).sum(1) + 1e-9                               # pixels with events (no gradient through it)
    loss = ((A ** 2).sum((1, 2)) / n).sum()
    assert abs(loss - float(z["chain_loss"])) <= 1e-5 * abs(float(z["chain_loss"]))
    dT = 2.0 * A / ((C + 1e-9) * n[:, None, None])
    dC = -2.0 * A ** 2 / ((C + 1e-9) * n[:, None, None])
    dw = np.zeros((B, 4 * N, 1), np.float64)
    for c in range(2):
        m4 = np.concatenate([wpm[:, :, c:c + 1]] * 4, 1)
        for b in range(B):
            dw[b, :, 0] += m4[b, :, 0] * (dC[b, c][ii[b]] + tau[b, :, 0] * dT[b, c][ii[b]])
    _, _, dwarped = oracle.get_interpolation(warped, H, W, dw.astype(np.float32))
    dwarped = dwarped * inside                                                 # loc * mask
    _, dfx, dfy, dloc_lookup = oracle.get_event_flow(fx, fy, loc, (dwarped * dt).astype(np.float32))
    dloc = dwarped + dloc_lookup
    assert rel_err(dfx, z["chain_dfx"]) < 2e-5 and rel_err(dfy, z["chain_dfy"]) < 2e-5
    assert rel_err(dloc, z["chain_dloc"]) < 2e-5
