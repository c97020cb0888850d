"""Round-6 fused launches of the network backward against the launches they replace, through the C ABI:
tef_dec_head_backward (= tef_grad_act + tef_conv_backward on the 1x1 head + tef_grad_act), tef_conv_backward_post (= input
gradient + tef_grad_act, inside the split-K reduction and as the in-place sweep), tef_convgru_cell_bwd_head (= tef_convgru_cell_bwd
+ tef_grad_act), tef_conv_pack_weights (= one tef_conv_pack_weight per part).  Element-wise results are the same operations in
the same order: bit-identical; per-channel sums (bias / 1x1 weight gradients) go through float atomics: 1e-5 relative.

Reference code these belong to: models/arch.py:238-240 (decoder tail), models/submodules.py:207-227 (ResidualBlock),
:95-152 (RecurrentConvLayer = head + ConvGRU)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
ACT = {None: 0, "relu": 1, "tanh": 2, "sigmoid": 3}


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available()
    import __graft_entry__ as ge

    ge.build()
    from taming_event_flow_amd import _lib

    return _lib, _lib.lib(), torch.device("cuda:0")


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _pack(_lib, lib, d, weights):
    np_, nt = ctypes.c_size_t(), ctypes.c_size_t()
    lib.tef_conv_packed_weight_floats(ctypes.byref(d), ctypes.byref(np_), ctypes.byref(nt))
    dev = weights[0].device
    wp, w2 = torch.empty(np_.value, device=dev), torch.empty(nt.value, device=dev)
    row0 = 0
    for w in weights:
        _lib.check(lib.tef_conv_pack_weight(ctypes.byref(d), w.data_ptr(), w.shape[0], row0, wp.data_ptr(), w2.data_ptr(), None), "pack")
        row0 += w.shape[0]
    return wp, w2


@pytest.mark.parametrize("B,C,H,W,nsrc,with_feat", [(2, 32, 16, 16, 2, True), (3, 72, 8, 20, 1, False), (1, 256, 6, 5, 2, True)])
def test_decoder_tail_backward(env, B, C, H, W, nsrc, with_feat):
    _lib, lib, dev = env
    g = torch.Generator().manual_seed(B * 100 + C)
    N, HW = 2, H * W
    dec = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    pred = torch.tanh(torch.randn(B, N, H, W, generator=g)).to(dev)
    srcs = [torch.randn(B, N, H, W, generator=g).to(dev) for _ in range(nsrc)]
    dfeat = torch.randn(B, C, H, W, generator=g).to(dev) if with_feat else None
    w = (torch.randn(N, C, 1, 1, generator=g) * 0.1).to(dev)
    head = _lib.ConvDesc(B, C, 0, H, W, N, 1, 1, ACT["tanh"])
    lin = _lib.ConvDesc(B, C, 0, H, W, N, 1, 1, ACT[None])
    _, w2 = _pack(_lib, lib, head, [w])
    ws = torch.empty(max(lib.tef_conv_workspace_bytes(ctypes.byref(head)), 1 << 20), dtype=torch.uint8, device=dev)
    # the launches it replaces
    gp0, dd0, gd0 = torch.empty_like(pred), torch.empty_like(dec), torch.empty_like(dec)
    dbp0, dwp0, dbd0 = torch.zeros(N, device=dev), torch.zeros(N, C, device=dev), torch.zeros(C, device=dev)
    _lib.check(lib.tef_grad_act(_ptrs(srcs), nsrc, pred.data_ptr(), ACT["tanh"], B, N, HW, gp0.data_ptr(), dbp0.data_ptr(), None), "grad_act")
    _lib.check(lib.tef_conv_backward_keep(ctypes.byref(lin), dec.data_ptr(), None, None, w2.data_ptr(), None, None, gp0.data_ptr(), None, N,
                                          dd0.data_ptr(), None, dwp0.data_ptr(), None, None, None, N, None, ws.data_ptr(), ws.numel(), None),
               "conv_backward_keep")
    feat = [dd0] + ([dfeat] if with_feat else [])
    _lib.check(lib.tef_grad_act(_ptrs(feat), len(feat), dec.data_ptr(), ACT["relu"], B, C, HW, gd0.data_ptr(), dbd0.data_ptr(), None), "grad_act")
    # the fused launch
    gp1, gd1 = torch.empty_like(pred), torch.empty_like(dec)
    dbp1, dwp1, dbd1 = torch.zeros(N, device=dev), torch.zeros(N, C, device=dev), torch.zeros(C, device=dev)
    _lib.check(lib.tef_dec_head_backward(ctypes.byref(head), _ptrs(srcs), nsrc, pred.data_ptr(), w2.data_ptr(), dec.data_ptr(), ACT["relu"],
                                         None if dfeat is None else dfeat.data_ptr(), gp1.data_ptr(), gd1.data_ptr(), dbp1.data_ptr(),
                                         dwp1.data_ptr(), dbd1.data_ptr(), None), "tef_dec_head_backward")
    torch.cuda.synchronize()
    assert torch.equal(gp0, gp1) and torch.equal(gd0, gd1)
    for a, b in ((dbp0, dbp1), (dwp0, dwp1), (dbd0, dbd1)):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-6)


@pytest.mark.parametrize("B,C,N,H,W,addend", [(8, 64, 64, 8, 8, True), (2, 32, 48, 8, 8, False), (1, 24, 40, 12, 20, True)])
def test_conv_backward_post(env, B, C, N, H, W, addend):
    """Input gradient delivered as the previous layer's pre-activation gradient: the split 8 x 8 geometry (inside the
    reduction) and an unsplit one (the in-place sweep)."""
    _lib, lib, dev = env
    g = torch.Generator().manual_seed(N)
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    gg = torch.randn(B, N, H, W, generator=g).to(dev)
    add = torch.randn(B, C, H, W, generator=g).to(dev) if addend else None
    w = (torch.randn(N, C, 3, 3, generator=g) * 0.05).to(dev)
    d = _lib.ConvDesc(B, C, 0, H, W, N, 3, 1, ACT[None])
    _, w2 = _pack(_lib, lib, d, [w])
    ws = torch.empty(max(lib.tef_conv_workspace_bytes(ctypes.byref(d)), 1 << 20), dtype=torch.uint8, device=dev)
    dx0, g0, db0 = torch.empty_like(x), torch.empty_like(x), torch.zeros(C, device=dev)
    _lib.check(lib.tef_conv_backward_keep(ctypes.byref(d), x.data_ptr(), None, None, w2.data_ptr(), None, None, gg.data_ptr(), None, N,
                                          dx0.data_ptr(), None, None, None, None, None, N, None, ws.data_ptr(), ws.numel(), None), "backward")
    srcs = [dx0] + ([add] if addend else [])
    _lib.check(lib.tef_grad_act(_ptrs(srcs), len(srcs), x.data_ptr(), ACT["relu"], B, C, H * W, g0.data_ptr(), db0.data_ptr(), None), "grad_act")

    class Post(ctypes.Structure):
        _fields_ = [("mask", ctypes.c_void_p), ("act", ctypes.c_int), ("addend", ctypes.c_void_p), ("g_out", ctypes.c_void_p),
                    ("dbias", ctypes.c_void_p)]

    g1, db1 = torch.empty_like(x), torch.zeros(C, device=dev)
    post = Post(x.data_ptr(), ACT["relu"], None if add is None else add.data_ptr(), g1.data_ptr(), db1.data_ptr())
    _lib.check(lib.tef_conv_backward_post(ctypes.byref(d), x.data_ptr(), w2.data_ptr(), gg.data_ptr(), None, ctypes.byref(post), ws.data_ptr(),
                                          ws.numel(), None), "tef_conv_backward_post")
    torch.cuda.synchronize()
    assert torch.equal(g0, g1)
    assert float((db0 - db1).abs().max()) <= 1e-5 * max(float(db0.abs().max()), 1e-6)


def test_pack_weights_in_one_launch(env):
    _lib, lib, dev = env
    g = torch.Generator().manual_seed(4)
    layers = [(_lib.ConvDesc(2, 64, 0, 64, 64, 128, 3, 2, 1), [(128, 64)]),           # stride-2 head (S2D rows too)
              (_lib.ConvDesc(2, 32, 32, 16, 16, 64, 3, 1, 3), [(32, 64), (32, 64)]),  # update | reset gates: two parts
              (_lib.ConvDesc(2, 66, 0, 32, 32, 32, 3, 1, 1), [(32, 66)]),             # ragged channel chunks
              (_lib.ConvDesc(2, 32, 0, 32, 32, 2, 1, 1, 2), [(2, 32)])]               # 1x1 head
    jobs, singles, batched = [], [], []
    for d, parts in layers:
        k = d.ksize
        ws_ = [(torch.randn(r, c, k, k, generator=g) * 0.1).to(dev) for r, c in parts]
        singles.append(_pack(_lib, lib, d, ws_))
        np_, nt = ctypes.c_size_t(), ctypes.c_size_t()
        lib.tef_conv_packed_weight_floats(ctypes.byref(d), ctypes.byref(np_), ctypes.byref(nt))
        wp, w2 = torch.full((np_.value,), float("nan"), device=dev), torch.full((nt.value,), float("nan"), device=dev)
        batched.append((wp, w2))
        row0 = 0
        for w in ws_:
            jobs.append((d, w, w.shape[0], row0, wp, w2))
            row0 += w.shape[0]
    arr = (_lib.PackJob * len(jobs))()
    for a, (d, w, rows, row0, wp, w2) in zip(arr, jobs):
        a.desc = d
        a.weight, a.rows, a.row0, a.wp, a.w2 = w.data_ptr(), rows, row0, wp.data_ptr(), w2.data_ptr()
    _lib.check(lib.tef_conv_pack_weights(arr, len(jobs), None), "tef_conv_pack_weights")
    torch.cuda.synchronize()
    for (a0, b0), (a1, b1) in zip(singles, batched):
        assert torch.equal(a0, a1) and torch.equal(b0, b1)


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 16, 16), (1, 24, 12, 20)])
def test_cell_backward_with_the_head_folded_in(env, B, C, H, W):
    _lib, lib, dev = env
    g = torch.Generator().manual_seed(C)
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).to(dev)
    h = torch.tanh(torch.randn(B, C, H, W, generator=g)).to(dev)
    w_u, w_r, w_o = [(torch.randn(C, 2 * C, 3, 3, generator=g) * 0.05).to(dev) for _ in range(3)]
    gd = _lib.GruDesc(B, C, H, W)
    ur = _lib.ConvDesc(B, C, C, H, W, 2 * C, 3, 1, ACT["sigmoid"])
    og = _lib.ConvDesc(B, C, C, H, W, C, 3, 1, ACT["tanh"])
    wp_ur, w2_ur = _pack(_lib, lib, ur, [w_u, w_r])
    wp_o, w2_o = _pack(_lib, lib, og, [w_o])
    ws = torch.empty(max(lib.tef_convgru_workspace_bytes(ctypes.byref(gd)), 1 << 20), dtype=torch.uint8, device=dev)
    u, r, o, hn = (torch.empty_like(x) for _ in range(4))
    _lib.check(lib.tef_convgru_cell_fwd(ctypes.byref(gd), x.data_ptr(), h.data_ptr(), wp_ur.data_ptr(), wp_o.data_ptr(), None, None,
                                        u.data_ptr(), r.data_ptr(), o.data_ptr(), hn.data_ptr(), ws.data_ptr(), ws.numel(), None), "cell_fwd")
    dhn = [torch.randn(B, C, H, W, generator=g).to(dev) for _ in range(2)]

    def run(fold):
        g_ur, g_o = torch.empty(B, 2 * C, H, W, device=dev), torch.empty_like(x)
        dx, dh, g_x, db = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.zeros(C, device=dev)
        if fold:
            _lib.check(lib.tef_convgru_cell_bwd_head(ctypes.byref(gd), x.data_ptr(), h.data_ptr(), u.data_ptr(), r.data_ptr(), o.data_ptr(),
                                                     _ptrs(dhn), 2, w2_ur.data_ptr(), w2_o.data_ptr(), g_ur.data_ptr(), g_o.data_ptr(), None,
                                                     dh.data_ptr(), None, None, None, None, None, None, ACT["relu"], g_x.data_ptr(),
                                                     db.data_ptr(), ws.data_ptr(), ws.numel(), None), "cell_bwd_head")
        else:
            _lib.check(lib.tef_convgru_cell_bwd(ctypes.byref(gd), x.data_ptr(), h.data_ptr(), u.data_ptr(), r.data_ptr(), o.data_ptr(),
                                                _ptrs(dhn), 2, w2_ur.data_ptr(), w2_o.data_ptr(), g_ur.data_ptr(), g_o.data_ptr(), dx.data_ptr(),
                                                dh.data_ptr(), None, None, None, None, None, None, ws.data_ptr(), ws.numel(), None), "cell_bwd")
            _lib.check(lib.tef_grad_act(_ptrs([dx]), 1, x.data_ptr(), ACT["relu"], B, C, H * W, g_x.data_ptr(), db.data_ptr(), None), "grad_act")
        torch.cuda.synchronize()
        return g_ur, g_o, dh, g_x, db

    a, b = run(False), run(True)
    for k in range(4):
        assert torch.equal(a[k], b[k]), k
    assert float((a[4] - b[4]).abs().max()) <= 1e-5 * max(float(a[4].abs().max()), 1e-6)
