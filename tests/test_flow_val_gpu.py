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
