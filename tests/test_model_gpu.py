"""GPU parity of the network rows (models/*.py on the MFMA conv kernels) vs golden vectors recorded from the
reference's nn.Conv2d implementation (tests/golden/make_golden_model.py).  Tolerance 1e-4 relative (max-norm)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    return torch.device("cuda:0")


def load_weights(module, seed, dev):
    from taming_event_flow_amd import synth

    sd = module.state_dict()
    w = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], seed)
    module.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return module.to(dev)


def digest(module):
    norms, heads = [], []
    for _, p in module.named_parameters():
        g = p.grad.detach().cpu().numpy().ravel()
        norms.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        h = np.zeros(32, np.float32)
        h[: min(32, g.size)] = g[:32]
        heads.append(h)
    return np.array(norms), np.stack(heads)


def check_digest(module, gnorm, ghead, tol=TOL):
    norms, heads = digest(module)
    assert np.abs(norms - gnorm).max() <= tol * gnorm.max()
    for i in range(len(norms)):
        assert abs(norms[i] - gnorm[i]) <= 10 * tol * max(gnorm[i], 1e-12), i
        assert np.abs(heads[i] - ghead[i]).max() <= 10 * tol * max(np.abs(ghead[i]).max(), 1e-3 * gnorm[i]), i


def test_conv_layers(dev):
    from taming_event_flow_amd.models.submodules import ConvLayer

    z = np.load(os.path.join(GOLDEN, "model_layers.npz"))
    for tag, (cin, cout, k, s, act) in {"conv_s2": (5, 12, 3, 2, "relu"), "conv_1x1": (7, 2, 1, 1, "tanh"),
                                        "conv_s1": (6, 40, 3, 1, None)}.items():
        layer = load_weights(ConvLayer(cin, cout, k, s, act), int(z[f"{tag}_seed"]), dev)
        x = torch.tensor(z[f"{tag}_x"], device=dev, requires_grad=True)
        y = layer(x)
        assert rel_err(y.detach().cpu().numpy(), z[f"{tag}_y"]) <= TOL, tag
        (y * torch.tensor(z[f"{tag}_r"], device=dev)).sum().backward()
        assert rel_err(x.grad.cpu().numpy(), z[f"{tag}_dx"]) <= TOL, tag
        assert rel_err(layer.conv2d.weight.grad.cpu().numpy(), z[f"{tag}_dw"]) <= TOL, tag
        assert rel_err(layer.conv2d.bias.grad.cpu().numpy(), z[f"{tag}_db"]) <= TOL, tag


def test_convgru_two_steps(dev):
    from taming_event_flow_amd.models.submodules import ConvGRU

    z = np.load(os.path.join(GOLDEN, "model_layers.npz"))
    gru = load_weights(ConvGRU(8, 8, 3), int(z["gru_seed"]), dev)
    x1 = torch.tensor(z["gru_x1"], device=dev, requires_grad=True)
    x2 = torch.tensor(z["gru_x2"], device=dev, requires_grad=True)
    h1, _ = gru(x1, None)
    h2, s2 = gru(x2, h1)
    assert s2 is h2
    assert rel_err(h1.detach().cpu().numpy(), z["gru_h1"]) <= TOL
    assert rel_err(h2.detach().cpu().numpy(), z["gru_h2"]) <= TOL
    (h2 * torch.tensor(z["gru_r"], device=dev)).sum().backward()
    assert rel_err(x1.grad.cpu().numpy(), z["gru_dx1"]) <= TOL
    assert rel_err(x2.grad.cpu().numpy(), z["gru_dx2"]) <= TOL
    check_digest(gru, z["gru_gnorm"], z["gru_ghead"])


def test_resblock_and_upsample(dev):
    from taming_event_flow_amd.models.submodules import ResidualBlock, UpsampleConvLayer

    z = np.load(os.path.join(GOLDEN, "model_layers.npz"))
    rb = load_weights(ResidualBlock(6, 6), int(z["rb_seed"]), dev)
    x = torch.tensor(z["rb_x"], device=dev, requires_grad=True)
    y2, y1 = rb(x)
    assert rel_err(y2.detach().cpu().numpy(), z["rb_y2"]) <= TOL
    assert rel_err(y1.detach().cpu().numpy(), z["rb_y1"]) <= TOL
    (y2 * torch.tensor(z["rb_r"], device=dev)).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), z["rb_dx"]) <= TOL
    check_digest(rb, z["rb_gnorm"], z["rb_ghead"])
    up = load_weights(UpsampleConvLayer(5, 7, 3), int(z["up_seed"]), dev)
    x = torch.tensor(z["up_x"], device=dev, requires_grad=True)
    y = up(x)
    assert rel_err(y.detach().cpu().numpy(), z["up_y"]) <= TOL
    (y * torch.tensor(z["up_r"], device=dev)).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), z["up_dx"]) <= TOL
    check_digest(up, z["up_gnorm"], z["up_ghead"])


@pytest.mark.parametrize("name", ["model_32x32", "model_40x52_pad"])
def test_recevflownet(name, dev):
    from taming_event_flow_amd.models.model import RecEVFlowNet

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), int(z["seed"]), dev)
    net.train()
    assert [k for k, _ in net.named_parameters()] == list(z["keys"])
    passes = int(z["passes"])
    loss = 0
    for t in range(passes):
        flows = net(torch.tensor(z[f"x{t}"], device=dev))["flow"]
        assert len(flows) == 4
        for i, fl in enumerate(flows):
            ref = z[f"flow{t}_{i}"]
            assert tuple(fl.shape) == ref.shape
            assert rel_err(fl.detach().cpu().numpy(), ref) <= TOL, (t, i)
            loss = loss + (fl * torch.tensor(z[f"r{t}_{i}"], device=dev)).sum()
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    for li, st in enumerate(net.states):
        assert rel_err(st.detach().cpu().numpy(), z[f"state{li}"]) <= TOL
    check_digest(net, z["gnorm"], z["ghead"])
    # state API of the reference (models/model.py:42-63)
    net.detach_states()
    assert all(not s.requires_grad for s in net.arch.states)
    net.reset_states()
    assert net.arch.states == [None] * 4


def test_fused_pass_matches_layer_by_layer(dev):
    """The fused pass (one autograd node, hand-written backward: models/engine.py) against the same network run layer by
    layer through the modules' own autograd nodes (MultiResUNetRecurrent.forward): flows, states, input-state gradients and
    every parameter gradient, over two recurrent passes with a gradient arriving through the carried state."""
    from taming_event_flow_amd.models.model import RecEVFlowNet
    from taming_event_flow_amd.models.submodules import upsample_bilinear

    rng = np.random.default_rng(3)
    xs = [torch.tensor(rng.poisson(0.4, (2, 2, 32, 48)).astype(np.float32), device=dev) for _ in range(2)]
    rs = [[torch.tensor(rng.standard_normal((2, 2, 32, 48)).astype(np.float32), device=dev) for _ in range(4)] for _ in range(2)]

    def run(fused):
        net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), 9, dev)
        net.train()
        loss, flows_all = 0, []
        for t in range(2):
            if fused:
                flows = net(xs[t])["flow"]
            else:
                preds = net.arch(xs[t])
                flows = [upsample_bilinear(p, 2 ** (3 - i), 2 ** (3 - i), mul=float(2 ** (3 - i))) for i, p in enumerate(preds)]
            flows_all.append([f.detach().clone() for f in flows])
            loss = loss + sum((f * r).sum() for f, r in zip(flows[1:], rs[t][1:]))     # head 0 of each pass gets no gradient
        loss.backward()
        return flows_all, [s.detach().clone() for s in net.arch.states], [p.grad.clone() for p in net.parameters()]

    fa, sa, ga = run(True)
    fb, sb, gb = run(False)
    for t in range(2):
        for i in range(4):
            assert rel_err(fa[t][i].cpu().numpy(), fb[t][i].cpu().numpy()) <= 1e-5, (t, i)
    for a, b in zip(sa, sb):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5
    for k, (a, b) in enumerate(zip(ga, gb)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 5e-5, k


@pytest.mark.parametrize("nres", [2, 0])
def test_two_node_pass_matches_fused(dev, nres):
    """The pass as two autograd nodes on two streams (models/engine.py `_EncFn` / `_DecFn` on `tef_net_pass_*_part`; what
    train.Trainer runs) against the one-node pass: three recurrent passes, a head of each pass without gradient, the
    input's gradient wanted; flows and states bit-identical, gradients to fp32 summation order (the decoder half hands
    autograd one tensor per level instead of adding in the cell kernel).  With and without residual blocks (without, the
    deepest state enters decoder 0 twice)."""
    from taming_event_flow_amd import parallel
    from taming_event_flow_amd.models import submodules
    from taming_event_flow_amd.models.model import RecEVFlowNet

    rng = np.random.default_rng(23 + nres)
    xs_np = [rng.poisson(0.4, (2, 2, 32, 48)).astype(np.float32) for _ in range(3)]
    rs = [[torch.tensor(rng.standard_normal((2, 2, 32, 48)).astype(np.float32), device=dev) for _ in range(4)] for _ in range(3)]

    def run(two):
        net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01, "num_residual_blocks": nres}, 2), 9, dev)
        net.train()
        bucket = parallel.FlatGradBucket(net.parameters())         # in-place parameter gradients (what the split needs)
        submodules.enable_direct_grads(net)
        net.arch.engine.side_stream = torch.cuda.Stream() if two else None
        xs = [torch.tensor(x, device=dev, requires_grad=True) for x in xs_np]
        loss, flows_all = 0, []
        for t in range(3):
            flows = net(xs[t])["flow"]                               # (joined: defer_join is off)
            flows_all.append([f.detach().clone() for f in flows])
            loss = loss + sum((f * r).sum() for f, r in zip(flows[1:], rs[t][1:]))
        loss.backward()
        torch.cuda.synchronize()
        return flows_all, [s.detach().clone() for s in net.arch.states], bucket.flat.clone(), [x.grad.clone() for x in xs]

    fa, sa, ga, xa = run(True)
    fb, sb, gb, xb = run(False)
    for t in range(3):
        for i in range(4):
            assert torch.equal(fa[t][i], fb[t][i]), (t, i)
    for a, b in zip(sa, sb):
        assert torch.equal(a, b)
    assert rel_err(ga.cpu().numpy(), gb.cpu().numpy()) <= 2e-5
    for a, b in zip(xa, xb):
        assert float(b.abs().max()) > 0 and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5


@pytest.mark.parametrize("nres", [2, 0])
def test_window_mode_and_level_pipeline_match_fused(dev, nres):
    """Window mode at the level of the model (arch.encode per pass, arch.decode_window once: the decoder halves of all passes
    as one batch, `_DecWinFn`) and its encoder halves pipelined by level over two streams (`_EncLowFn` / `_EncHighFn` on
    tef_net_pass_forward_levels / _backward_levels; the first pass — no states yet — takes the single call) against the
    one-node pass: four recurrent passes, a head of each pass without gradient, the input's gradient wanted.  States
    bit-identical (the encoder halves are the same launches); flows to fp32 summation order against the one-node pass (a
    batch of P x B samples is tiled and split over k differently) and bit-identical between the two window forms; gradients
    to fp32 summation order."""
    from taming_event_flow_amd import parallel
    from taming_event_flow_amd.models import submodules
    from taming_event_flow_amd.models.model import RecEVFlowNet

    P = 4
    rng = np.random.default_rng(41 + nres)
    xs_np = [rng.poisson(0.4, (2, 2, 32, 48)).astype(np.float32) for _ in range(P)]
    rs = [[torch.tensor(rng.standard_normal((2, 2, 32, 48)).astype(np.float32), device=dev) for _ in range(4)] for _ in range(P)]

    def run(mode):
        net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01, "num_residual_blocks": nres}, 2), 9, dev)
        net.train()
        bucket = parallel.FlatGradBucket(net.parameters())
        submodules.enable_direct_grads(net)
        eng = net.arch.engine
        if mode == "pipeline":
            eng.enc_streams = (torch.cuda.Stream(), torch.cuda.Stream())
        xs = [torch.tensor(x, device=dev, requires_grad=True) for x in xs_np]
        if mode == "fused":
            flows_all = [net(xs[t])["flow"] for t in range(P)]
        else:
            for t in range(P):
                net.arch.encode(xs[t])
            flows_all = net.arch.decode_window()
        loss = sum((f * r).sum() for t in range(P) for f, r in zip(flows_all[t][1:], rs[t][1:]))
        loss.backward()
        eng.join()
        eng.flush_window()
        torch.cuda.synchronize()
        return ([[f.detach().clone() for f in fl] for fl in flows_all], [s.detach().clone() for s in net.arch.states],
                bucket.flat.clone(), [x.grad.clone() for x in xs])

    ref = run("fused")
    first = None
    for mode in ("window", "pipeline"):
        got = run(mode)
        for t in range(P):
            for i in range(4):
                assert rel_err(got[0][t][i].cpu().numpy(), ref[0][t][i].cpu().numpy()) <= 1e-5, (mode, t, i)
                assert first is None or torch.equal(got[0][t][i], first[0][t][i]), (mode, t, i)
        first = first or got
        for a, b in zip(got[1], ref[1]):
            assert torch.equal(a, b), mode
        assert rel_err(got[2].cpu().numpy(), ref[2].cpu().numpy()) <= 2e-5, mode
        for a, b in zip(got[3], ref[3]):
            assert float(b.abs().max()) > 0 and rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5, mode


@pytest.mark.parametrize("two", [False, True])
def test_backward_twice_is_refused(dev, two):
    """The fused pass releases its activation arena after its backward (a BPTT window holds ten of them): a second
    backward through the same pass (retain_graph=True) is refused with an explanation, not answered from freed memory."""
    from taming_event_flow_amd import parallel
    from taming_event_flow_amd.models import submodules
    from taming_event_flow_amd.models.model import RecEVFlowNet

    net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), 9, dev)
    net.train()
    parallel.FlatGradBucket(net.parameters())
    submodules.enable_direct_grads(net)
    net.arch.engine.side_stream = torch.cuda.Stream() if two else None
    x = torch.rand(2, 2, 32, 32, device=dev)
    loss = sum(f.sum() for f in net(x)["flow"])
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


@pytest.mark.parametrize("nres", [2, 0])
def test_input_gradient_and_no_residual_blocks(dev, nres):
    """d loss / d network input through the fused pass (the reference's autograd delivers it, models/arch.py:217-227;
    round 2 returned None silently) against the layer-by-layer path, and the architecture without residual blocks, where
    the deepest state enters decoder 0 twice (features + skip): its gradient must count both."""
    from taming_event_flow_amd.models.model import RecEVFlowNet
    from taming_event_flow_amd.models.submodules import upsample_bilinear

    rng = np.random.default_rng(17 + nres)
    x_np = rng.poisson(0.4, (2, 2, 32, 48)).astype(np.float32)
    rs = [torch.tensor(rng.standard_normal((2, 2, 32, 48)).astype(np.float32), device=dev) for _ in range(4)]

    def run(fused):
        net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01, "num_residual_blocks": nres}, 2), 9, dev)
        net.train()
        assert len(net.arch.resblocks) == nres
        x = torch.tensor(x_np, device=dev, requires_grad=True)
        if fused:
            flows = net(x)["flow"]
        else:
            preds = net.arch(x)
            flows = [upsample_bilinear(p, 2 ** (3 - i), 2 ** (3 - i), mul=float(2 ** (3 - i))) for i, p in enumerate(preds)]
        sum((f * r).sum() for f, r in zip(flows, rs)).backward()
        assert x.grad is not None and tuple(x.grad.shape) == x_np.shape
        return x.grad.clone(), [p.grad.clone() for p in net.parameters()]

    dxa, ga = run(True)
    dxb, gb = run(False)
    assert float(dxb.abs().max()) > 0
    assert rel_err(dxa.cpu().numpy(), dxb.cpu().numpy()) <= 5e-5
    for k, (a, b) in enumerate(zip(ga, gb)):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 5e-5, k


def test_input_gradient_padded_shape(dev):
    """Input sides that are not multiples of 16 are padded at the top / left inside the pass: the input gradient is the
    crop of the padded input's gradient."""
    from taming_event_flow_amd.models.model import RecEVFlowNet

    rng = np.random.default_rng(23)
    x_np = rng.poisson(0.4, (1, 2, 40, 52)).astype(np.float32)
    r_np = rng.standard_normal((4, 1, 2, 40, 52)).astype(np.float32)

    def run(pad):
        net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), 4, dev)
        net.train()
        if pad:
            xp = np.zeros((1, 2, 48, 64), np.float32)
            xp[:, :, 8:, 12:] = x_np
            x = torch.tensor(xp, device=dev, requires_grad=True)
            flows = [f[:, :, 8:, 12:] for f in net(x)["flow"]]
        else:
            x = torch.tensor(x_np, device=dev, requires_grad=True)
            flows = net(x)["flow"]
        sum((f * torch.tensor(r, device=dev)).sum() for f, r in zip(flows, r_np)).backward()
        return x.grad[:, :, 8:, 12:] if pad else x.grad

    a, b = run(False), run(True)
    assert tuple(a.shape) == (1, 2, 40, 52)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5


def test_dsec_eval_shape_forward(dev):
    """BASELINE config 5 shape: 480x640 inference (no grad), states carried over two passes; cross-checked against the
    same network evaluated on a crop-free 2x-downsampled... no reference is stored at this size, so the check is
    structural: shapes, finiteness, state shapes, and determinism of a repeated pass from the same state."""
    from taming_event_flow_amd.models.model import RecEVFlowNet

    net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), 5, dev)
    net.eval()
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.poisson(0.2, (1, 2, 480, 640)).astype(np.float32), device=dev)
    with torch.no_grad():
        f1 = net(x)["flow"]
        s1 = net.states
        f2 = net(x)["flow"]
        net.states = s1
        f2b = net(x)["flow"]
    assert [tuple(f.shape) for f in f1] == [(1, 2, 480, 640)] * 4
    assert all(torch.isfinite(f).all() for f in f1 + f2)
    assert [tuple(s.shape) for s in s1] == [(1, 64, 240, 320), (1, 128, 120, 160), (1, 256, 60, 80), (1, 512, 30, 40)]
    for a, b in zip(f2, f2b):
        assert torch.equal(a, b)                 # forward is deterministic (no atomics on the forward path)
    assert not torch.equal(f1[-1], f2[-1])       # the recurrent state matters


def test_dsec_eval_shape_against_reference(dev):
    """480x640 inference (BASELINE configs[4]) against the reference's recorded flows (tests/golden/
    model_480x640_eval.npz: two recurrent passes; every flow map on a stride-4 lattice plus float64 sums of the full maps
    and of the final states)."""
    from taming_event_flow_amd.models.model import RecEVFlowNet

    z = np.load(os.path.join(GOLDEN, "model_480x640_eval.npz"))
    st = int(z["stride"])
    net = load_weights(RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2), int(z["seed"]), dev)
    net.eval()
    with torch.no_grad():
        for t in range(int(z["passes"])):
            flows = net(torch.tensor(z[f"x{t}"].astype(np.float32), device=dev))["flow"]
            for i, fl in enumerate(flows):
                f = fl.cpu().numpy()
                assert f.shape == (1, 2, 480, 640)
                assert rel_err(f[:, :, ::st, ::st], z[f"flow{t}_{i}"]) <= TOL, (t, i)
                ref_sum, ref_abs = z[f"sum{t}_{i}"]
                assert abs(np.abs(f.astype(np.float64)).sum() - ref_abs) <= 1e-4 * ref_abs, (t, i)
                assert abs(f.astype(np.float64).sum() - ref_sum) <= 1e-4 * ref_abs, (t, i)
        for li, s_ in enumerate(net.states):
            a = s_.cpu().numpy().astype(np.float64)
            ref_sum, ref_abs = z[f"state_sum{li}"]
            assert abs(np.abs(a).sum() - ref_abs) <= 1e-4 * ref_abs and abs(a.sum() - ref_sum) <= 1e-4 * ref_abs, li


def test_conv_against_torch_reference_large(dev):
    """A full-size layer (ConvGRU level 1 at B=8: 128 ch @ 32x32) against torch's own fp32 conv on the same device."""
    from taming_event_flow_amd.models.submodules import ConvGRU

    torch.manual_seed(0)
    gru = ConvGRU(128, 128, 3).to(dev)
    x = torch.randn(8, 128, 32, 32, device=dev, requires_grad=True)
    h = torch.randn(8, 128, 32, 32, device=dev, requires_grad=True)
    out, _ = gru(x, h)
    r = torch.randn_like(out)
    (out * r).sum().backward()
    got = [out.detach().clone(), x.grad.clone(), h.grad.clone()] + [p.grad.clone() for p in gru.parameters()]
    for p in gru.parameters():
        p.grad = None
    x.grad = h.grad = None
    F = torch.nn.functional
    s = torch.cat([x, h], 1)
    u = torch.sigmoid(F.conv2d(s, gru.update_gate.weight, gru.update_gate.bias, padding=1))
    rr = torch.sigmoid(F.conv2d(s, gru.reset_gate.weight, gru.reset_gate.bias, padding=1))
    o = torch.tanh(F.conv2d(torch.cat([x, h * rr], 1), gru.out_gate.weight, gru.out_gate.bias, padding=1))
    ref_out = h * (1 - u) + o * u
    (ref_out * r).sum().backward()
    ref = [ref_out.detach(), x.grad, h.grad] + [p.grad for p in gru.parameters()]
    for a, b in zip(got, ref):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-4


@pytest.mark.parametrize("B,C0,C1,N,H,W,k,stride,act,gated", [
    (3, 5, 0, 7, 5, 7, 3, 1, "relu", False),        # nothing a multiple of 4: scalar gathers everywhere
    (2, 3, 4, 33, 9, 6, 3, 1, "tanh", True),        # concat + gate, ragged rows, 2 row tiles of 32
    (2, 6, 0, 70, 11, 13, 3, 2, "sigmoid", False),  # stride 2 on odd sizes, 64-row tile
    (1, 9, 0, 2, 6, 10, 1, 1, None, False),         # 1x1 prediction head shape
    (2, 4, 4, 130, 8, 12, 3, 1, "relu", True),      # quad-vector gathers (W % 4 == 0), 128-row tile + remainder, gate
    (1, 2, 0, 5, 4, 4, 3, 1, None, False),          # a single 16-pixel image: every quad touches the tensor's ends
    (1, 12, 0, 20, 30, 40, 3, 1, "relu", False),    # 8 x 16 halo rectangles (deepest eval level), <= 32 output rows
    (2, 5, 6, 70, 33, 24, 3, 1, "tanh", True),      # 8 x 16 rectangles, ragged bottom edge, concat + gate
    (1, 64, 8, 40, 30, 40, 3, 1, "relu", True),     # 8 x 16 rectangles with the reduction split over slabs
    (2, 20, 0, 40, 16, 32, 3, 2, "relu", False),    # stride-2 input gradient by parity classes (8 x 16 gradient grid)
    (1, 64, 0, 24, 24, 48, 3, 2, None, False),      # the same on general 8 x 16 rectangles (12 x 24 gradient grid)
    (2, 130, 0, 9, 16, 16, 3, 2, "tanh", False),    # class boundaries inside a row tile (130 channels), two 8 x 8 grids per tile
    (4, 48, 0, 200, 32, 32, 3, 2, "relu", False),   # the same with the reduction split over slabs (25 chunks)
])
def test_conv_ragged_geometries(dev, B, C0, C1, N, H, W, k, stride, act, gated):
    """tef_conv_forward / backward against torch's CPU fp32 convolution on shapes that exercise every staging path."""
    from taming_event_flow_amd.models import submodules as sm

    g = torch.Generator().manual_seed(B * 1000 + N)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    gate = torch.rand(B, C1, H, W, generator=g) if gated else None
    w = torch.randn(N, C0 + C1, k, k, generator=g) * 0.3
    b = torch.randn(N, generator=g)
    leaves = [t for t in (x0, x1, gate, w, b) if t is not None]

    def ref():
        ts = [t.clone().requires_grad_() for t in leaves]
        it = iter(ts)
        rx0 = next(it)
        rx1 = next(it) if x1 is not None else None
        rg = next(it) if gate is not None else None
        rw, rb = next(it), next(it)
        xin = rx0 if rx1 is None else torch.cat([rx0, rx1 * rg if rg is not None else rx1], dim=1)
        y = torch.nn.functional.conv2d(xin, rw, rb, stride=stride, padding=k // 2)
        if act is not None:
            y = getattr(torch, act)(y)
        return y, ts

    y_ref, ts_ref = ref()
    dout = torch.randn(y_ref.shape, generator=g)
    g_ref = torch.autograd.grad(y_ref, ts_ref, dout)

    ts = [t.to(dev).requires_grad_() for t in leaves]
    it = iter(ts)
    dx0 = next(it)
    dx1 = next(it) if x1 is not None else None
    dg = next(it) if gate is not None else None
    dw, db = next(it), next(it)
    y = sm.conv2d(sm.PackedWeights(), dx0, dw, db, stride=stride, act=act, x1=dx1, gate1=dg)
    assert y.shape == y_ref.shape
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) <= 1e-5
    grads = torch.autograd.grad(y, ts, dout.to(dev))
    for got, want, name in zip(grads, g_ref, ["x0", "x1", "gate", "w", "b"] if gated else
                               (["x0", "x1", "w", "b"] if x1 is not None else ["x0", "w", "b"])):
        assert rel_err(got.cpu().numpy(), want.numpy()) <= 2e-5, name


def test_conv_randomised_sweep(dev):
    """150 random convolution geometries (tools/fuzz_conv.py) against torch's CPU convolution in float64, forward and
    every gradient; the long form of this sweep ran 3000 cases (seed 2026), worst 4.4e-5, plus one single-channel bias
    gradient (a sum of 60 cancelling terms) at 2.3e-4 where torch's own fp32 result is 1.3e-4 from float64."""
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "fuzz_conv", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_conv.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad, worst = fuzz.sweep(150, seed=99, verbose=False)
    assert bad == 0 and worst <= TOL
