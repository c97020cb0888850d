"""utils/iwe.py's training-time primitives as stand-alone DIFFERENTIABLE operators (csrc/tef_prims.hip behind
taming_event_flow_amd/utils/iwe.py) against values AND gradients recorded from the reference's own torch functions
(tests/golden/primitives.npz, made by tests/golden/make_golden.py --primitives):
get_event_flow (utils/iwe.py:17-40), get_interpolation (:63-113), interpolate (:116-136), and a focus loss assembled from
the primitives one by one the way a caller of that module would (lookup -> event_propagation -> purge_unfeasible ->
corners -> per-polarity images -> loss/flow.py:112-129), gradients to both flow maps and the event locations.
Tolerance 1e-4 relative (the north-star bar), 1e-5 on single primitives.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prim():
    import __graft_entry__ as ge

    ge.build()
    from taming_event_flow_amd.utils import iwe

    z = np.load(os.path.join(GOLDEN, "primitives.npz"))
    dev = torch.device("cuda:0")
    return iwe, z, dev


def test_event_flow_value_and_gradients(prim):
    iwe, z, dev = prim
    fx = torch.tensor(z["gef_fx"], device=dev, requires_grad=True)
    fy = torch.tensor(z["gef_fy"], device=dev, requires_grad=True)
    loc = torch.tensor(z["gef_loc"], device=dev, requires_grad=True)
    out = iwe.get_event_flow(fx, fy, loc)
    assert rel_err(out.detach().cpu().numpy(), z["gef_out"]) < 1e-6
    (out * torch.tensor(z["gef_w"], device=dev)).sum().backward()
    assert rel_err(fx.grad.cpu().numpy(), z["gef_dfx"]) < 1e-5
    assert rel_err(fy.grad.cpu().numpy(), z["gef_dfy"]) < 1e-5
    assert rel_err(loc.grad.cpu().numpy(), z["gef_dloc"]) < 1e-5
    # only the locations need a gradient: the maps' scatter is skipped, the result is the same
    loc2 = torch.tensor(z["gef_loc"], device=dev, requires_grad=True)
    (iwe.get_event_flow(fx.detach(), fy.detach(), loc2) * torch.tensor(z["gef_w"], device=dev)).sum().backward()
    assert np.array_equal(loc2.grad.cpu().numpy(), loc.grad.cpu().numpy())


def test_interpolation_weights_gradient(prim):
    iwe, z, dev = prim
    H, W = int(z["H"]), int(z["W"])
    pos = torch.tensor(z["gef_loc"], device=dev, requires_grad=True)
    idx, w = iwe.get_interpolation(pos, (H, W))
    assert not idx.requires_grad and w.requires_grad
    assert np.array_equal(idx.cpu().numpy(), z["gi_idx"]) and rel_err(w.detach().cpu().numpy(), z["gi_w"]) < 1e-6
    (w * torch.tensor(z["gi_r"], device=dev)).sum().backward()
    assert rel_err(pos.grad.cpu().numpy(), z["gi_dpos"]) < 1e-6
    # nearest-pixel branch: weights are ones * mask, constant in the locations
    pos_r = torch.tensor(z["gef_loc"], device=dev, requires_grad=True)
    idx_r, w_r = iwe.get_interpolation(pos_r, (H, W), round_idx=True)
    assert np.array_equal(idx_r.cpu().numpy(), z["gi_round_idx"])
    w_r.sum().backward()
    assert pos_r.grad is None or float(pos_r.grad.abs().max()) == 0.0


def test_interpolate_gradients(prim):
    iwe, z, dev = prim
    H, W = int(z["H"]), int(z["W"])
    idx = torch.tensor(z["gi_idx"], device=dev)
    w = torch.tensor(z["gi_w"], device=dev, requires_grad=True)
    pm4 = torch.cat([torch.tensor(z["purge_pm"], device=dev)[:, :, 0:1]] * 4, 1).clone().requires_grad_()
    zeros = torch.zeros(idx.shape[0], H * W, 1, device=dev, requires_grad=True)
    img = iwe.interpolate(idx, w, (H, W), polarity_mask=pm4, zeros=zeros)
    assert rel_err(img.detach().cpu().numpy(), z["interp_img"]) < 1e-6
    g = torch.randn(img.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    (img * g).sum().backward()
    # scatter_add_'s backward is a gather of the image gradient (reference :134): d w = g[idx] * mask, d mask = g[idx] * w
    gi = g.view(idx.shape[0], -1).cpu().numpy()
    at = np.take_along_axis(gi, z["gi_idx"][:, :, 0].astype(np.int64), axis=1)[:, :, None]
    assert np.array_equal(w.grad.cpu().numpy(), at * pm4.detach().cpu().numpy())
    assert np.array_equal(pm4.grad.cpu().numpy(), at * z["gi_w"])
    assert np.array_equal(zeros.grad.cpu().numpy().reshape(gi.shape), gi)


def test_loss_assembled_from_the_primitives(prim):
    iwe, z, dev = prim
    H, W = int(z["H"]), int(z["W"])
    fx = torch.tensor(z["chain_fx"], device=dev, requires_grad=True)
    fy = torch.tensor(z["chain_fy"], device=dev, requires_grad=True)
    loc = torch.tensor(z["chain_loc"], device=dev, requires_grad=True)
    ts, pm = torch.tensor(z["chain_ts"], device=dev), torch.tensor(z["chain_pm"], device=dev)
    flow = iwe.get_event_flow(fx, fy, loc)
    warped = iwe.event_propagation(ts, loc, flow, 1.0)
    warped, wpm = iwe.purge_unfeasible(warped, pm, (H, W))
    idx, w = iwe.get_interpolation(warped, (H, W))
    tau = torch.cat([1.0 - (1.0 - ts)] * 4, 1)
    imgs, timgs = [], []
    for c in range(2):
        m4 = torch.cat([wpm[:, :, c:c + 1]] * 4, 1)
        imgs.append(iwe.interpolate(idx, w, (H, W), polarity_mask=m4))
        timgs.append(iwe.interpolate(idx, w * tau, (H, W), polarity_mask=m4))
    img, timg = torch.cat(imgs, 1), torch.cat(timgs, 1)
    assert rel_err(img.detach().cpu().numpy(), z["chain_iwe"]) < 1e-5
    assert rel_err(timg.detach().cpu().numpy(), z["chain_iwe_ts"]) < 1e-5
    # loss/flow.py:112-129 focus_loss with loss_scaling: sum of squared per-pixel mean timestamps / pixels with events
    a = timg / (img + 1e-9)
    B = img.shape[0]
    nz = ((img[:, 0:1] + img[:, 1:2]).view(B, -1) != 0).sum(1).float() + 1e-9
    loss = (((a[:, 0].reshape(B, -1) ** 2).sum(1) + (a[:, 1].reshape(B, -1) ** 2).sum(1)) / nz).sum()
    assert abs(float(loss.detach()) - float(z["chain_loss"])) <= 1e-5 * abs(float(z["chain_loss"]))
    loss.backward()
    assert rel_err(fx.grad.cpu().numpy(), z["chain_dfx"]) < 1e-4
    assert rel_err(fy.grad.cpu().numpy(), z["chain_dfy"]) < 1e-4
    assert rel_err(loc.grad.cpu().numpy(), z["chain_dloc"]) < 1e-4
