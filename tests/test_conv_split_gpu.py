"""Split reductions of the deep layers (tef_conv.hip: conv3x3_halo_kernel EPI_SLAB + splitk_reduce_kernel /
splitk_reduce_s2d_kernel): layers whose few output tiles are split over the input channels, at the geometries the
network's 8x8 .. 32x32 levels produce.  Outputs and input gradients against torch's float64 convolution at the 1e-4 bar,
and bit-identical from run to run (the slabs are added in slab order; a reduction that depended on arrival order would
differ between repeats).

Reference layers these geometries come from: models/model.py RecEVFlowNet encoders / ConvGRU gates / residual blocks
(stride-1, stride-2 heads and their input gradients, gated two-source gates).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# B, C0, C1, N, H, W, stride, act, gated
CASES = [
    (2, 256, 0, 256, 8, 8, 1, "relu", False),           # 8 x 8: two images per tile
    (4, 128, 128, 256, 8, 8, 1, "sigmoid", True),       # ConvGRU gate pair on the concatenated, gated source
    (1, 128, 0, 128, 16, 16, 1, "tanh", False),
    (2, 96, 32, 100, 16, 16, 1, None, False),           # ragged row tile (100 rows of 128)
    (1, 256, 0, 64, 32, 32, 1, "relu", False),
    (1, 64, 0, 128, 32, 32, 2, "relu", False),          # stride-2 head: forward split and the S2D input gradient
    (2, 128, 0, 256, 16, 16, 2, "relu", False),         # S2D on the 8 x 8 gradient grid
    (1, 200, 0, 72, 12, 40, 1, "relu", False),          # general rectangles with ragged edges
]
REPEATS = 6


def _inputs(ci):
    B, C0, C1, N, H, W, stride, act, gated = CASES[ci]
    g = torch.Generator().manual_seed(100 + ci)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    gate = torch.rand(B, C1, H, W, generator=g) if gated else None
    w = torch.randn(N, C0 + C1, 3, 3, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ho, wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    dout = torch.randn(B, N, ho, wo, generator=g)
    return x0, x1, gate, w, b, dout


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_split_layers_are_repeatable_and_match_float64(ci):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    ge.build()
    from taming_event_flow_amd.models import submodules as sm

    dev = torch.device("cuda:0")
    B, C0, C1, N, H, W, stride, act, gated = CASES[ci]
    x0, x1, gate, w, b, dout = _inputs(ci)
    leaves = [t for t in (x0, x1, gate, w, b) if t is not None]
    names = ["x0"] + (["x1"] if x1 is not None else []) + (["gate"] if gate is not None else []) + ["w", "b"]
    first = {}
    for rep in range(REPEATS):
        ds = dict(zip(names, [t.to(dev).requires_grad_() for t in leaves]))
        y = sm.conv2d(sm.PackedWeights(), ds["x0"], ds["w"], ds["b"], stride=stride, act=act, x1=ds.get("x1"),
                      gate1=ds.get("gate"))
        grads = torch.autograd.grad(y, list(ds.values()), dout.to(dev))
        got = {"y": y.detach().cpu().numpy()}
        got.update({"d" + n: gr.cpu().numpy() for n, gr in zip(names, grads)})
        for k in ("y", "dx0", "dx1", "dgate"):          # weight / bias gradients accumulate with float atomics
            if k not in got:
                continue
            if rep == 0:
                first[k] = got[k]
            else:
                assert np.array_equal(first[k], got[k]), f"{k}: repeat {rep} differs from repeat 0"
    rx0 = x0.double().requires_grad_()
    xin = rx0
    if x1 is not None:
        xin = torch.cat([rx0, x1.double() * gate.double() if gated else x1.double()], dim=1)
    yr = torch.nn.functional.conv2d(xin, w.double(), b.double(), stride=stride, padding=1)
    if act is not None:
        yr = getattr(torch, act)(yr)
    (gx0,) = torch.autograd.grad(yr, [rx0], dout.double())
    for got, want in ((first["y"], yr.detach().numpy()), (first["dx0"], gx0.numpy())):
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= 1e-4, err
