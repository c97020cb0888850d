"""GPU parity of the validation metrics (loss/flow_val.py mirror on tef_val.hip) vs golden vectors recorded from the
reference: RSAT / FWL after every pass, windowed event / IWE / flow images, AEE, in-place time shift."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", ["val_linear_24x30", "val_iterative_24x30", "val_iterative_20x26"])
def test_validation_golden(name):
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.loss import flow_val

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kind, H, W, passes = str(z["kind"]), int(z["H"]), int(z["W"]), int(z["passes"])
    cfg = {"loader": {"resolution": [H, W]}, "loss": {"round_ts": bool(z["round_ts"])}, "vis": {"mask_output": True},
           "metrics": {}}
    V = getattr(flow_val, kind)(cfg, dev)
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    for p in range(passes):
        ev = t(z[f"ev{p}"])
        V.update([t(z[f"low{p}"]), t(z[f"flow{p}"])], ev, t(z[f"pm{p}"]), t(z[f"mask{p}"]))
        np.testing.assert_array_equal(ev.cpu().numpy(), z[f"ev_after{p}"])       # in-place shift (flow_val.py:88)
        assert V.num_passes == p + 1
        assert abs(float(V.rsat().item()) - float(z[f"rsat{p}"])) <= TOL * float(z[f"rsat{p}"]), p
        assert abs(float(V.fwl().item()) - float(z[f"fwl{p}"])) <= TOL * float(z[f"fwl{p}"]), p
    np.testing.assert_array_equal(V.window_events(round_idx=True).cpu().numpy(), z["events_round"])
    assert rel_err(V.window_events(round_idx=False).cpu().numpy(), z["events_bilinear"]) <= 1e-6
    if kind == "Iterative":
        for mode in ("forward", "backward"):
            assert rel_err(V.window_iwe(mode=mode, round_idx=True).cpu().numpy(), z[f"iwe_{mode}_round"]) <= 1e-6, mode
            assert rel_err(V.window_iwe(mode=mode, round_idx=False).cpu().numpy(), z[f"iwe_{mode}"]) <= TOL, mode
            got, ref = V.window_flow(mode=mode, mask=True).cpu().numpy(), z[f"flow_{mode}"]
            assert rel_err(np.nan_to_num(got), np.nan_to_num(ref)) <= TOL, mode
            got, ref = V.window_flow(mode=mode, mask=False).cpu().numpy(), z[f"flow_{mode}_nomask"]
            assert np.array_equal(np.isnan(got), np.isnan(ref)), mode
            assert rel_err(np.nan_to_num(got, posinf=0, neginf=0), np.nan_to_num(ref, posinf=0, neginf=0)) <= TOL, mode
        assert rel_err(V.window_flow(mode=None, mask=True).cpu().numpy(), z["flow_none"]) <= TOL
    else:
        assert rel_err(V.window_iwe(round_idx=True).cpu().numpy(), z["iwe_round"]) <= 1e-6
        assert rel_err(V.window_iwe(round_idx=False).cpu().numpy(), z["iwe"]) <= TOL
        assert rel_err(V.window_flow(mask=True).cpu().numpy(), z["flow_mask"]) <= TOL
        assert rel_err(V.window_flow(mask=False).cpu().numpy(), z["flow_nomask"]) <= TOL
    aee = V.compute_aee(t(z["aee_pred"]), t(z["aee_gt"]))
    assert abs(float(aee.item()) - float(z["aee_nomask"])) <= TOL * float(z["aee_nomask"])
    aee = V.compute_aee(t(z["aee_pred"]), t(z["aee_gt"]), mask=V._event_mask.unsqueeze(0))
    assert abs(float(aee.item()) - float(z["aee_mask"])) <= TOL * float(z["aee_mask"])
    V.reset()
    assert V.num_passes == 0


def test_validation_dsec_shape():
    """BASELINE configs[4] at its real size against the reference: flow_val.Iterative at 480x640, 10 passes x 100 000
    events (tests/golden/make_golden_val.py --dsec-shape; inputs regenerated from the seed and checked against the
    recorded digest): FWL / RSAT after every pass, every window image on a stride-4 lattice and by float64 sums over all
    pixels."""
    import hashlib

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import synth
    from taming_event_flow_amd.loss import flow_val

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "val_iterative_480x640.npz"))
    H, W, passes, N, stride = int(z["H"]), int(z["W"]), int(z["passes"]), int(z["N"]), int(z["stride"])
    inp = synth.make_eval_window(int(z["seed"]), H, W, passes, N)
    h = hashlib.sha256()
    for d in inp:
        for k in ("ev", "pm", "flow", "low", "mask"):
            h.update(np.ascontiguousarray(d[k]).tobytes())
    assert h.hexdigest() == str(z["digest"]), "regenerated inputs differ from the ones the reference was run on"
    cfg = {"loader": {"resolution": [H, W]}, "loss": {"round_ts": False}, "vis": {"mask_output": True}, "metrics": {}}
    V = flow_val.Iterative(cfg, dev)
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    worst = 0.0
    for p, d in enumerate(inp):
        V.update([t(d["low"]), t(d["flow"])], t(d["ev"]), t(d["pm"]), t(d["mask"]))
        for name, got in (("rsat", V.rsat()), ("fwl", V.fwl())):
            e = abs(float(got.item()) - float(z[f"{name}{p}"])) / float(z[f"{name}{p}"])
            worst = max(worst, e)
            assert e <= TOL, (name, p, e)
    images = {"events_round": V.window_events(round_idx=True), "events_bilinear": V.window_events(round_idx=False),
              "flow_none": V.window_flow(mode=None, mask=True)}
    for mode in ("forward", "backward"):
        images[f"iwe_{mode}_round"] = V.window_iwe(mode=mode, round_idx=True)
        images[f"iwe_{mode}"] = V.window_iwe(mode=mode, round_idx=False)
        images[f"flow_{mode}"] = V.window_flow(mode=mode, mask=True)
    for name, img in images.items():
        a = img.cpu().numpy()
        fin = np.isfinite(a)
        assert int((~fin).sum()) == int(z[f"{name}.nonfinite"]), name
        a0 = np.where(fin, a, 0)
        e_lat = rel_err(a0[..., ::stride, ::stride], np.nan_to_num(z[f"{name}.lattice"], posinf=0, neginf=0))
        a64 = a0.astype(np.float64)
        den = np.maximum(z[f"{name}.abs_sum"], 1e-30)
        e_sum = float((np.abs(a64.sum(axis=(-1, -2)) - z[f"{name}.sum"]) / den).max())
        e_abs = float((np.abs(np.abs(a64).sum(axis=(-1, -2)) - z[f"{name}.abs_sum"]) / den).max())
        e_sq = float((np.abs((a64 * a64).sum(axis=(-1, -2)) - z[f"{name}.sq_sum"]) / np.maximum(z[f"{name}.sq_sum"], 1e-30)).max())
        print(f"{name}: lattice {e_lat:.2e} sum {e_sum:.2e} abs-sum {e_abs:.2e} sq-sum {e_sq:.2e}")
        assert e_lat <= TOL and e_sum <= TOL and e_abs <= TOL and e_sq <= 2 * TOL, name
    print(f"validation window 480x640: worst FWL / RSAT relative error {worst:.2e}")


def test_compute_pol_iwe():
    """utils/iwe.py compute_pol_iwe / deblur_events (one-shot IWE for visualisation) vs the reference, all 4 modes."""
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.utils import iwe

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "pol_iwe.npz"))
    H, W = int(z["H"]), int(z["W"])
    flow, ev, evf, pm = (torch.tensor(z[k], device=dev) for k in ("flow", "ev", "evf", "pm"))
    for ri in (True, False):
        for rf in (True, False):       # the reference's nearest-flow gather assumes integer pixel coordinates
            got = iwe.compute_pol_iwe(flow, ev if rf else evf, (H, W), pm, round_idx=ri, round_flow=rf).cpu().numpy()
            assert rel_err(got, z[f"iwe_{int(ri)}{int(rf)}"]) <= 1e-5, (ri, rf)
    one = iwe.deblur_events(flow, ev, (H, W), polarity_mask=pm[:, :, 0:1]).cpu().numpy()
    assert rel_err(one, z["iwe_11"][:, 0:1]) <= 1e-5


def test_iwe_primitives_standalone():
    """utils.iwe's training-time primitives as importable functions (reference utils/iwe.py:5-136), forward values against
    the reference's recorded outputs (tests/golden/primitives.npz)."""
    import os

    from conftest import GOLDEN, rel_err
    from taming_event_flow_amd.utils import iwe

    z = np.load(os.path.join(GOLDEN, "primitives.npz"))
    H, W = int(z["H"]), int(z["W"])
    dev = torch.device("cuda:0")
    t = lambda a: torch.tensor(a, device=dev)      # noqa: E731
    ef = iwe.get_event_flow(t(z["gef_fx"]), t(z["gef_fy"]), t(z["gef_loc"]))
    assert rel_err(ef.cpu().numpy(), z["gef_out"]) < 1e-6
    prop = iwe.event_propagation(t(z["prop_ts"]), t(z["gef_loc"]), t(z["gef_out"]), 1.0)
    assert rel_err(prop.cpu().numpy(), z["prop_out"]) < 1e-6
    ploc, ppm = iwe.purge_unfeasible(prop, t(z["purge_pm"]), (H, W))
    assert np.array_equal(ploc.cpu().numpy(), z["purge_loc"]) and np.array_equal(ppm.cpu().numpy(), z["purge_mask"])
    idx, w = iwe.get_interpolation(t(z["gef_loc"]), (H, W))
    assert np.array_equal(idx.cpu().numpy(), z["gi_idx"]) and rel_err(w.cpu().numpy(), z["gi_w"]) < 1e-6
    pm4 = torch.cat([t(z["purge_pm"])[:, :, 0:1]] * 4, 1)
    img = iwe.interpolate(idx, w, (H, W), polarity_mask=pm4)
    assert img.shape == z["interp_img"].shape and rel_err(img.cpu().numpy(), z["interp_img"]) < 1e-6
    idx_r, w_r = iwe.get_interpolation(t(z["gef_loc"]), (H, W), round_idx=True)
    assert np.array_equal(idx_r.cpu().numpy(), z["gi_round_idx"]) and np.array_equal(w_r.cpu().numpy(), z["gi_round_w"])


def test_visualisation_functions_refuse_gradients():
    """compute_pol_iwe / deblur_events are forward-only here (eval_flow.py:104-111 calls them for images): a call whose
    result could be differentiated raises instead of returning a constant.  (The training-time primitives are
    differentiable: tests/test_prims_gpu.py.)"""
    from taming_event_flow_amd.utils import iwe

    dev = torch.device("cuda:0")
    with pytest.raises(RuntimeError, match="forward-only"):
        iwe.compute_pol_iwe(torch.randn(1, 2, 8, 9, device=dev, requires_grad=True), torch.rand(1, 5, 4, device=dev),
                            (8, 9), torch.ones(1, 5, 2, device=dev))
    with torch.no_grad():                                   # a constant is what the caller asked for: fine
        out = iwe.compute_pol_iwe(torch.randn(1, 2, 8, 9, device=dev, requires_grad=True), torch.rand(1, 5, 4, device=dev),
                                  (8, 9), torch.ones(1, 5, 2, device=dev))
    assert out.shape == (1, 2, 8, 9)
