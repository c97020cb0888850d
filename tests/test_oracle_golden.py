"""Pin the CPU oracle (oracle/tef_oracle.c) against golden vectors recorded from the reference.

Reference under test (by recorded outputs): loss/flow.py Iterative/Linear, utils/iwe.py primitives,
dataloader/encodings.py.  Tolerances: 1e-5 relative (the north-star bar for the HIP path is 1e-4).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, FULL_RES_CASES, ITERATIVE_CASES, LINEAR_CASES, elementwise_excess, load_case, rel_err
from oracle import oracle

TOL = 1e-5


@pytest.mark.parametrize("name", ITERATIVE_CASES + LINEAR_CASES + FULL_RES_CASES)
def test_loss_cases(name):
    meta, win, loss, dflows = load_case(name)
    w = oracle.Window(win["flows"], win["ev"], win["pm"], win["dev"], win["dpm"], S=meta["S"], mode=meta["mode"],
                      round_ts=meta["round_ts"], loss_scaling=meta.get("loss_scaling", True),
                      border_compensation=meta.get("border_compensation", True))
    l, d = w.loss(meta["kind"], meta["spat"], meta["temp"])
    assert abs(l - loss) <= TOL * abs(loss), (l, loss)
    assert d.shape == dflows.shape
    assert rel_err(d, dflows) <= 5 * TOL
    # per-map check so that a small map cannot hide behind a large one
    for t in range(meta["P"]):
        for i in range(meta["F"]):
            if np.abs(dflows[t, i]).max() > 0:
                assert rel_err(d[t, i], dflows[t, i]) <= 1e-3, (t, i)
    # element by element against each pixel's own scale (conftest.elementwise_excess; calibrates the bar the HIP path is
    # held to in tests/test_loss_gpu.py: the worst of these cases is 2.5)
    mass = w.gradient_mass(meta["kind"])
    assert (np.abs(d - (w.smoothing(meta["spat"], meta["temp"])[1] if (meta["spat"] is not None or meta["temp"] is not None) else 0))
            <= mass * (1 + 1e-5) + 1e-30).all()          # |signed sum| <= sum of magnitudes
    if meta["spat"] is not None or meta["temp"] is not None:
        mass = mass + np.abs(w.smoothing(meta["spat"], meta["temp"])[1])
    assert elementwise_excess(d, dflows, mass)[0] <= 3.0


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
