"""Bilinear up-sampling kernels (tef_upsample_bilinear_crop / _backward) against torch's CPU interpolate
(align_corners=False; reference call sites models/submodules.py:264 and models/model.py:79): every scale the network
uses, the x2 fast paths (even / odd widths, the summed second input) and the generic path, with and without the
top / left crop of the flow heads; the backward against autograd of the same expression."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # planes, H, W, scale, crop_top, crop_left, second input, mul
    (6, 8, 8, 2, 0, 0, True, 1.0),       # x2 fast paths, W % 4 == 0
    (5, 7, 6, 2, 0, 0, False, 1.0),      # x2, W even but not a multiple of 4
    (4, 5, 7, 2, 0, 0, True, 1.0),       # x2, odd width: generic forward, per-pixel backward
    (3, 6, 10, 2, 3, 5, False, 2.0),     # x2 with crop (flow head at half resolution)
    (2, 4, 5, 8, 5, 3, False, 8.0),      # x8 with crop
    (2, 6, 6, 4, 0, 2, False, 4.0),      # x4
    (2, 9, 11, 1, 1, 2, False, 1.0),     # x1: a crop only
    (64, 32, 32, 2, 0, 0, True, 1.0),    # a decoder-sized plane count
]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("planes,H,W,s,ct,cl,second,mul", CASES)
def test_upsample_against_torch(planes, H, W, s, ct, cl, second, mul):
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import _lib

    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(planes * 1000 + H * 10 + W)
    x = torch.tensor(rng.standard_normal((1, planes, H, W)).astype(np.float32), requires_grad=True)
    x2 = torch.tensor(rng.standard_normal((1, planes, H, W)).astype(np.float32)) if second else None
    ref = mul * F.interpolate(x + x2 if second else x, scale_factor=s, mode="bilinear", align_corners=False)[:, :, ct:, cl:]
    r = torch.tensor(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * r).sum().backward()
    xd = x.detach().to(dev).contiguous()
    x2d = x2.to(dev).contiguous() if second else None
    y = torch.empty((planes, H * s - ct, W * s - cl), device=dev)
    rc = lib.tef_upsample_bilinear_crop(_ptr(xd), _ptr(x2d), planes, H, W, s, s, mul, ct, cl, _ptr(y), st)
    assert rc == 0, lib.tef_last_error()
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy()[0], rtol=1e-5, atol=1e-6)
    dy = r[0].to(dev).contiguous()
    dx = torch.empty((planes, H, W), device=dev)
    rc = lib.tef_upsample_bilinear_crop_backward(_ptr(dy), planes, H, W, s, s, mul, ct, cl, _ptr(dx), st)
    assert rc == 0, lib.tef_last_error()
    np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy()[0], rtol=1e-5, atol=2e-6)
