"""GPU parity of the training-window semantics (reference train_flow.py:80-156) on the full HIP path:
HIP encoder -> RecEVFlowNet (MFMA convs) -> Iterative loss (HIP) -> backward -> clip -> Adam, two windows."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# Measured (round 4, tests/golden/train_trace_f64.npz): over all parameters the HIP gradient is 8.7e-5 from the reference's
# float64 gradient, the reference's own float32 run 2.35e-4 — the HIP path is 2.7x NEARER to the truth; per parameter the
# ratio is 0.35 ... 0.7 for the 52 encoder / residual / deepest-decoder tensors and 1.0 ... 1.9 for the three shallow
# decoders and heads (2.0e-4 against 1.06e-4).  The distance between the two float32 runs (3.2e-4, what
# test_two_window_trace sees) is therefore mostly the reference's own distance from the truth.
TOL_ANCHOR_GLOBAL, TOL_ANCHOR_PARAM = 1.0, 2.5


def _digest_errors(params_grad_flat, params, gnorm, ghead):
    """-> (worst error of a parameter's gradient norm relative to the LARGEST norm, worst relative to its own norm, worst
    error of a 32-element head relative to max(|head|, 1e-3 x the parameter's norm)) — check_digest's three criteria."""
    e_glob = e_own = e_head = 0.0
    o = 0
    for i, p in enumerate(params):
        g = params_grad_flat[o:o + p.numel()].double().cpu().numpy()
        o += p.numel()
        n = float(np.sqrt((g ** 2).sum()))
        e_glob = max(e_glob, abs(n - gnorm[i]) / gnorm.max())
        e_own = max(e_own, abs(n - gnorm[i]) / max(gnorm[i], 1e-12))
        h = np.zeros(32)
        h[: min(32, g.size)] = g[:32]
        e_head = max(e_head, np.abs(h - ghead[i]).max() / max(np.abs(ghead[i]).max(), 1e-3 * gnorm[i], 1e-30))
    return e_glob, e_own, e_head


def _trace_config(z):
    H, W, B, P = int(z["H"]), int(z["W"]), int(z["B"]), int(z["P"])
    return {
        "data": {"passes_loss": P, "scales_loss": 1, "voxel": None},
        "model": {"name": "RecEVFlowNet", "final_w_scale": 0.01},
        "loss": {"warping": "Iterative", "iterative_mode": "two", "round_ts": False, "flow_scaling": 32,
                 "flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "clip_grad": float(z["clip"])},
        "optimizer": {"name": "Adam", "lr": float(z["lr"])},
        "loader": {"batch_size": B, "resolution": [H, W], "max_num_grad_events": None, "seed": 0},
    }


def _load_trace_weights(tr, z, dev):
    from taming_event_flow_amd import synth

    sd = tr.model.state_dict()
    w = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], int(z["seed"]))
    tr.model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})


@pytest.mark.parametrize("trace", ["train_trace", "train_trace_lr1e-5"])
def test_two_window_trace(trace):
    """Two consecutive loss windows against the reference's recorded trace: loss, pre-clip gradient (global norm and, per
    parameter, norm + first 32 elements), Adam update norms.  `train_trace_lr1e-5` runs at the reference's own learning
    rate: Adam's first steps move every weight by +-lr whatever the gradient's size, so the second window's distance
    between two fp32 implementations scales with lr and can be held to 1e-3 there (lr = 1e-3: percent level)."""
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import synth, train
    from taming_event_flow_amd.dataloader import encodings

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, trace + ".npz"))
    H, W, B, P = int(z["H"]), int(z["W"]), int(z["B"]), int(z["P"])
    cfg = {
        "data": {"passes_loss": P, "scales_loss": 1, "voxel": None},
        "model": {"name": "RecEVFlowNet", "final_w_scale": 0.01},
        "loss": {"warping": "Iterative", "iterative_mode": "two", "round_ts": False, "flow_scaling": 32,
                 "flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "clip_grad": float(z["clip"])},
        "optimizer": {"name": "Adam", "lr": float(z["lr"])},
        "loader": {"batch_size": B, "resolution": [H, W], "max_num_grad_events": None, "seed": 0},
    }
    tr = train.Trainer(cfg, dev)
    sd = tr.model.state_dict()
    w = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], int(z["seed"]))
    tr.model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    small_lr = float(z["lr"]) <= 1e-4
    for win in range(int(z["windows"])):
        before = [p.detach().clone() for p in tr.model.parameters()]
        for t in range(P):
            ev, pm = torch.tensor(z[f"ev{win}_{t}"], device=dev), torch.tensor(z[f"pm{win}_{t}"], device=dev)
            dv, dpm = torch.tensor(z[f"dev{win}_{t}"], device=dev), torch.tensor(z[f"dpm{win}_{t}"], device=dev)
            net_input = encodings.event_list_to_channels(torch.cat([ev, dv], 1), (H, W))
            np.testing.assert_array_equal(net_input.cpu().numpy(), z[f"inp{win}_{t}"])     # counts: bit-exact
            batch = {"net_input": net_input, "event_list": ev, "event_list_pol_mask": pm, "d_event_list": dv,
                     "d_event_list_pol_mask": dpm}
            if t < P - 1:
                assert not tr.step(batch, new_seq=(win == 0 and t == 0))
            else:
                # the last pass of the window taken apart (what Trainer.step does, train.py::_pass) so that the gradient can
                # be read BEFORE clipping and the optimiser step
                assert tr._forward_update(batch)
                tr._backward_window()
                tr.bucket.all_reduce_sum()
                flat = tr.bucket.flat.detach().clone()
                tr._apply_update()
        loss, gn = float(tr.last_loss.item()), float(tr.last_grad_norm.item())
        assert abs(float(flat.double().norm()) - gn) <= 1e-5 * gn            # what was clipped is what was read
        delta = np.array([float((p.detach() - b0).double().norm()) for p, b0 in zip(tr.model.parameters(), before)])
        e_glob, e_own, e_head = _digest_errors(flat, tr.bucket.params, z[f"pgnorm{win}"], z[f"pghead{win}"])
        print(f"{trace} window {win}: loss rel {abs(loss - float(z[f'loss{win}'])) / abs(float(z[f'loss{win}'])):.2e} "
              f"gnorm rel {abs(gn - float(z[f'gnorm{win}'])) / float(z[f'gnorm{win}']):.2e} per-parameter norm {e_glob:.2e} "
              f"(own {e_own:.2e}) heads {e_head:.2e}")
        # window 0: the same weights as the reference.  The loss meets the north-star bar; the PARAMETER gradient went through
        # P passes of BPTT in fp32 on both sides: the reference's own float32 run is 2.35e-4 from its float64 run, the HIP
        # path 0.87e-4 (test_bptt_gradient_accuracy_anchor), the two are 3.2e-4 apart: 5e-4.  window 1: weights after one
        # Adam step (lr 1e-5: 3.7e-4 measured).
        tol = 1e-4 if win == 0 else (1e-3 if small_lr else 2e-2)
        gtol = 5e-4 if win == 0 else (1e-3 if small_lr else 5e-2)
        assert abs(loss - float(z[f"loss{win}"])) <= tol * abs(float(z[f"loss{win}"])), (win, loss)
        assert abs(gn - float(z[f"gnorm{win}"])) <= gtol * float(z[f"gnorm{win}"]), (win, gn)
        assert e_glob <= gtol and e_own <= 5 * gtol and e_head <= 5 * gtol, (win, e_glob, e_own, e_head)
        ref = z[f"delta{win}"]
        assert np.abs(delta - ref).max() <= 5e-2 * ref.max(), win
        assert tr.loss_function.num_passes == 0
        assert all(s is not None and not s.requires_grad for s in tr.model.arch.states)


def test_bptt_gradient_accuracy_anchor():
    """Which side of the 3e-4 between the HIP parameter gradients and the reference's fp32 CPU run is nearer to the truth?
    tests/golden/train_trace_f64.npz holds window 0 of the trace run through the reference in FLOAT64 and in float32
    (a seeded subset of <= 2048 elements per parameter, both precisions).  The HIP path's distance to the float64 gradient
    must not exceed 1.5 x the distance of the reference's own float32 run — per parameter (with a floor of 1e-5 of the
    parameter's gradient norm for parameters both runs resolve almost exactly) and over all parameters together."""
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train
    from taming_event_flow_amd.dataloader import encodings

    z = np.load(os.path.join(GOLDEN, "train_trace.npz"))
    a = np.load(os.path.join(GOLDEN, "train_trace_f64.npz"))
    dev = torch.device("cuda:0")
    H, W, B, P = int(z["H"]), int(z["W"]), int(z["B"]), int(z["P"])
    cfg = _trace_config(z)
    tr = train.Trainer(cfg, dev)
    _load_trace_weights(tr, z, dev)
    tr.reset()
    for t in range(P):
        ev, pm = torch.tensor(z[f"ev0_{t}"], device=dev), torch.tensor(z[f"pm0_{t}"], device=dev)
        dv, dpm = torch.tensor(z[f"dev0_{t}"], device=dev), torch.tensor(z[f"dpm0_{t}"], device=dev)
        batch = {"net_input": encodings.event_list_to_channels(torch.cat([ev, dv], 1), (H, W)), "event_list": ev,
                 "event_list_pol_mask": pm, "d_event_list": dv, "d_event_list_pol_mask": dpm}
        if t < P - 1:
            tr.step(batch, new_seq=(t == 0))
        else:
            assert tr._forward_update(batch)
            tr._backward_window()
    names = [n for n, _ in tr.model.named_parameters()]
    assert names == [str(n) for n in a["names"]]
    off, idx = a["offsets"], a["index"]
    e_hip_all = e_ref_all = n_all = 0.0
    rows = []
    for k, p in enumerate(tr.model.parameters()):
        sel = idx[off[k]:off[k + 1]]
        hip = p.grad.detach().reshape(-1)[torch.tensor(sel, device=dev)].double().cpu().numpy()
        g64, g32 = a["g64"][off[k]:off[k + 1]], a["g32"][off[k]:off[k + 1]].astype(np.float64)
        e_hip, e_ref, nrm = np.linalg.norm(hip - g64), np.linalg.norm(g32 - g64), np.linalg.norm(g64)
        e_hip_all, e_ref_all, n_all = e_hip_all + e_hip ** 2, e_ref_all + e_ref ** 2, n_all + nrm ** 2
        rows.append((e_hip / max(e_ref, 1e-5 * nrm, 1e-30), names[k], e_hip / max(nrm, 1e-30), e_ref / max(nrm, 1e-30),
                     np.linalg.norm(hip - g32) / max(nrm, 1e-30)))
    e_hip_all, e_ref_all, n_all = np.sqrt(e_hip_all), np.sqrt(e_ref_all), np.sqrt(n_all)
    print(f"distance to the float64 gradient over all parameters: HIP {e_hip_all / n_all:.2e}, reference fp32 {e_ref_all / n_all:.2e}")
    for r in sorted(rows, reverse=True):
        print(f"  {r[1]:44s} hip-f64 {r[2]:.2e}  ref32-f64 {r[3]:.2e}  hip-ref32 {r[4]:.2e}  ratio {r[0]:.2f}")
    assert e_hip_all <= TOL_ANCHOR_GLOBAL * e_ref_all
    assert max(r[0] for r in rows) <= TOL_ANCHOR_PARAM, max(rows)


def test_graph_replay_matches_eager():
    """A loss window captured into a hipGraph (Trainer.capture_window) reproduces the eager window."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[32, 32], max_num_grad_events=300)
    cfg["data"]["passes_loss"] = 3
    cfg["optimizer"].update(lr=1e-4, capturable=True)

    def fresh():
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev)
        src = train.SyntheticSequences(cfg, dev, 380, seq_len=10 ** 9, seed=3)
        tr.reset()
        return tr, [src.next() for _ in range(3)], [src.next() for _ in range(3)]

    def clone(batches):
        return [{k: v.clone() for k, v in b.items()} for b in batches]

    def eager(order):
        tr_e, win_a, win_b = fresh()
        out = []
        for name in order:
            for b in clone({"A": win_a, "B": win_b}[name]):
                tr_e.step(b, new_seq=False)
            out.append(float(tr_e.last_loss.item()))
        return out

    losses_e = eager("AAB")
    l3_stale = eager("AAA")[2]                     # what a replay that ignored the refreshed inputs would give

    tr_g, win_a, win_b = fresh()
    keep = clone(win_a)
    cw = tr_g.capture_window(win_a, warmup=1)      # window 1 (A) runs eagerly, then the window is captured
    l1 = tr_g.warmup_losses[0]
    for b, k in zip(win_a, keep):                  # the caller's tensors are not touched by capture / replay
        for name in b:
            assert torch.equal(b[name], k[name]), name
    cw.replay()                                    # A again
    torch.cuda.synchronize()
    l2 = float(tr_g.last_loss.item())
    for dst, src in zip(cw.inputs, win_b):         # the caller refreshes the static input buffers: window B
        for name in dst:
            dst[name].copy_(src[name])
    cw()
    torch.cuda.synchronize()
    l3 = float(tr_g.last_loss.item())
    assert abs(l1 - losses_e[0]) <= 1e-5 * abs(losses_e[0]), (l1, losses_e)
    assert abs(l2 - losses_e[1]) <= 1e-3 * abs(losses_e[1]), (l2, losses_e)      # state + weights carried over
    assert abs(l3 - losses_e[2]) <= 2e-2 * abs(losses_e[2]), (l3, losses_e)      # fp32 atomics order + Adam amplify
    assert abs(l3 - losses_e[2]) < abs(l3 - l3_stale)                            # the refreshed inputs were used
    # a sequence change at the window boundary: the static recurrent state is cleared before the replay
    cw.replay(new_seq=True)
    torch.cuda.synchronize()
    assert np.isfinite(float(tr_g.last_loss.item()))


def test_parked_windows_only_where_nothing_is_baked_in(monkeypatch):
    """Round-5 advisory: a closed CapturedWindow is parked (and handed out again by the next capture_window of the same
    shapes) only with the fused optimiser, whose hyper-parameters live in device memory; a torch.optim optimiser bakes lr into
    the captured kernels, so its windows are retired and a new capture is a NEW capture.  The park's key holds the clip: a
    trainer whose loss.clip_grad changed captures afresh; and a replay uses the clip of ITS capture whatever an eager step
    in between used."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[32, 32], max_num_grad_events=300)
    cfg["data"]["passes_loss"] = 2
    cfg["optimizer"].update(lr=1e-4, capturable=True)

    def batches(tr):
        src = train.SyntheticSequences(cfg, dev, 380, seq_len=10 ** 9, seed=3)
        tr.reset()
        return [src.next() for _ in range(2)]

    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev)
    assert tr.fused_opt is not None
    win = batches(tr)
    cw = tr.capture_window(win, warmup=1)
    graph = cw.graph
    cw.close()
    assert len(tr._parked) == 1
    cw2 = tr.capture_window(win, warmup=1)
    assert cw2.graph is graph                       # handed out again
    # the clip of the capture (100) stays the captured window's, whatever an eager step used in between
    assert cw2.max_norm == cfg["loss"]["clip_grad"]
    tr.fused_opt.step(max_norm=1e-9)                # an eager step with another clip (gradient is zero here: a no-op update)
    cw2.replay()
    torch.cuda.synchronize()
    assert tr.fused_opt._hp_uploaded[4] == float(cfg["loss"]["clip_grad"])
    cw2.close()
    tr.cfg["loss"]["clip_grad"] = 50.0              # another clip: not the parked window's signature any more
    cw3 = tr.capture_window(win, warmup=1)
    assert cw3.graph is not graph and cw3.max_norm == 50.0
    cw3.close()
    tr.close()
    # torch.optim.Adam: lr is a constant of the captured kernels -> never parked
    monkeypatch.setenv("TEF_TORCH_ADAM", "1")
    torch.manual_seed(7)
    tr2 = train.Trainer(cfg, dev)
    assert tr2.fused_opt is None
    win2 = batches(tr2)
    n0 = train.retired_graph_count()
    cwt = tr2.capture_window(win2, warmup=1)
    cwt.close()
    assert len(tr2._parked) == 0 and train.retired_graph_count() > n0
    tr2.close()


@pytest.mark.parametrize("B,R,passes", [(2, 32, 3), (4, 64, 4)])
def test_deferred_weight_gradients_match_immediate(B, R, passes):
    """Weight gradients of a BPTT window computed per layer in one long reduction (flush_deferred_wgrads) equal the per-pass
    accumulation; the two differ only in the order of the fp32 atomics."""
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd.models import submodules as sm
    from taming_event_flow_amd.models.model import RecEVFlowNet

    dev = torch.device("cuda:0")
    grads = []
    for deferred in (False, True):
        torch.manual_seed(3)
        net = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2).to(dev)
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        sm.enable_direct_grads(net)
        sm.enable_deferred_wgrad(net, deferred)
        gen = torch.Generator().manual_seed(5)
        total = 0.0
        for _ in range(passes):                 # passes with recurrent state, one backward through all of them
            x = torch.rand(B, 2, R, R, generator=gen).to(dev)
            total = total + sum((f * f).sum() for f in net(x)["flow"])
        total.backward()
        if deferred:
            assert any(eng._pending for eng in sm._DEFERRED_ENGINES)      # the fused passes queue their backward calls
            sm.flush_deferred_wgrads()
            assert not sm._DEFERRED and not sm._DEFERRED_ENGINES
        grads.append([p.grad.detach().clone() for p in net.parameters()])
    for a, b in zip(*grads):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-12)


def _trace_run(two_streams, delay, window_decode=False, level_delay=None):
    """Two eager windows of the golden trace's inputs (fresh input tensors every pass, dropped right after the call — what
    a data loader does) -> [loss, pre-clip gradient norm] per window + a parameter checksum."""
    from taming_event_flow_amd import synth, train
    from taming_event_flow_amd.dataloader import encodings

    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, "train_trace_lr1e-5.npz"))
    H, W, B, P = int(z["H"]), int(z["W"]), int(z["B"]), int(z["P"])
    cfg = {
        "data": {"passes_loss": P, "scales_loss": 1, "voxel": None},
        "model": {"name": "RecEVFlowNet", "final_w_scale": 0.01},
        "loss": {"warping": "Iterative", "iterative_mode": "two", "round_ts": False, "flow_scaling": 32,
                 "flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "clip_grad": float(z["clip"])},
        "optimizer": {"name": "Adam", "lr": float(z["lr"])},
        "loader": {"batch_size": B, "resolution": [H, W], "max_num_grad_events": None, "seed": 0},
    }
    # (the switch is a constructor argument here: changing os.environ under a process whose HIP runtime threads are alive
    # is a setenv / getenv race)
    tr = train.Trainer(cfg, dev, streams=two_streams, window_decode=window_decode)
    assert (tr.dec_stream is not None) == (two_streams and not window_decode) and (tr.wgrad_stream is not None) == two_streams
    tr.model.arch.engine.debug_delay = delay
    assert (tr.model.arch.engine.enc_streams is not None) == (two_streams and window_decode)
    tr.model.arch.engine.debug_delay_levels = level_delay
    sd = tr.model.state_dict()
    w = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], int(z["seed"]))
    tr.model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    out = []
    for win in range(int(z["windows"])):
        for t in range(P):
            ev, pm = torch.tensor(z[f"ev{win}_{t}"], device=dev), torch.tensor(z[f"pm{win}_{t}"], device=dev)
            dv, dpm = torch.tensor(z[f"dev{win}_{t}"], device=dev), torch.tensor(z[f"dpm{win}_{t}"], device=dev)
            net_input = encodings.event_list_to_channels(torch.cat([ev, dv], 1), (H, W))
            tr.step({"net_input": net_input, "event_list": ev, "event_list_pol_mask": pm, "d_event_list": dv,
                     "d_event_list_pol_mask": dpm}, new_seq=(win == 0 and t == 0))
            del ev, pm, dv, dpm, net_input
            torch.empty(1 << 20, device=dev).fill_(float("nan"))      # whatever was freed is overwritten at once
        out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
    out.append(float(sum(p.detach().double().abs().sum() for p in tr.model.parameters())))
    return np.array(out)


def test_two_stream_window_has_no_race():
    """The multi-stream window (models/engine.py: encoders of pass t + 1 beside the decoders of pass t, autograd mirroring
    it in BPTT, groups of deferred weight gradients on a third stream) against the same window on one stream, with one of
    the streams held back by a spinning kernel in front of every piece of its work: a missing dependency or a buffer
    released under a lagging stream shows as a different loss / gradient."""
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    ref = _trace_run(False, None)
    spin = 30_000_000                  # ~12 ms per half pass
    for delay in (None, (0, spin, 0), (spin, 0, 0), (0, 0, 4 * spin)):
        got = _trace_run(True, delay)
        err = np.abs(got - ref) / np.abs(ref)
        # (the second window's gradient norm moves by ~5e-5 between two IDENTICAL one-stream runs: float atomics in the
        # weight gradients, then an Adam step; a race is orders of magnitude above that — the unguarded input buffers this
        # test was written against gave a 40 % different loss)
        tol = np.array([1e-5, 1e-5, 1e-5, 5e-4, 1e-6])
        assert np.isfinite(got).all() and (err <= tol).all(), (delay, got, ref)
    # window mode (round 6, the Trainer's default): encoder halves pass by pass, the decoder halves of the whole window as
    # one batch, its weight gradients on the reduction stream beside the encoders' BPTT — on one stream, on its streams, and
    # with the reduction stream / the main stream held back
    for streams, delay in ((False, None), (True, None), (True, (0, 0, 4 * spin)), (True, (spin, spin, 0))):
        got = _trace_run(streams, delay, window_decode=True)
        err = np.abs(got - ref) / np.abs(ref)
        assert np.isfinite(got).all() and (err <= tol).all(), ("window", streams, delay, got, ref)
    # ... whose encoder levels are pipelined over two more streams (levels 0-1 of pass t + 1 beside levels 2-3 of pass t,
    # forward and backward): the lower / the upper range held back in front of every piece of its work
    for level_delay in ((spin, 0), (0, spin)):
        got = _trace_run(True, None, window_decode=True, level_delay=level_delay)
        err = np.abs(got - ref) / np.abs(ref)
        assert np.isfinite(got).all() and (err <= tol).all(), ("window, level ranges", level_delay, got, ref)


@pytest.mark.parametrize("warping,scales,smooth,graph", [("Linear", 2, True, False), ("Iterative", 2, True, False),
                                                         ("Iterative", 1, False, True)])
def test_multi_stream_window_matches_one_stream(warping, scales, smooth, graph):
    """train.Trainer's multi-stream window against TEF_TWO_STREAMS=0 on configurations the golden traces do not cover:
    the Linear loss (its update() samples the newest flow map — on the side stream), two temporal scales, the smoothing
    priors (they read the container's planar flow copies), ragged event counts with detached events, and the window as
    a captured hipGraph.  Same weights, same synthetic passes, three windows chained through the recurrent state."""
    assert torch.cuda.is_available()
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4, scales_loss=scales)
    cfg["loss"].update(warping=warping)
    if smooth:
        cfg["loss"].update(flow_spat_smooth_weight=0.001, flow_temp_smooth_weight=0.1)
    cfg["optimizer"]["capturable"] = graph
    # lr = 0: the windows differ through the recurrent state only.  (With a learning rate, Adam's first steps move every
    # weight by +-lr whatever its gradient's size; weights whose gradient is noise go either way with the summation order of
    # the float atomics, and the next window's gradient norm lands on one of a few values several percent apart — in the
    # one-stream run just as much, from one process to the next.)
    cfg["optimizer"]["lr"] = 0.0

    def run(streams):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=streams)
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
        tr.reset()
        out = []
        if graph:
            win = tr.capture_window([src.next() for _ in range(4)], warmup=1)
            for _ in range(3):
                for b in win.inputs:
                    for k, v in src.next().items():
                        b[k].copy_(v)
                win()
                out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
        else:
            for _ in range(3):
                for _ in range(4):
                    tr.step(src.next(), new_seq=False)
                out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
        # release the trainer (streams, hipGraphs, arenas, autograd records) HERE, with the device idle, not whenever the
        # cycle collector happens to run
        if graph:
            del win          # (CapturedWindow.__del__ -> close(): waits, drops the graphs, waits again — DESIGN section 9d)
        del tr, src
        return np.array(out)

    one, multi = run(False), run(True)
    err = np.abs(multi - one) / np.abs(one)
    print("relative differences (loss, gradient norm per window):", err)
    tol_loss, tol_norm = 1e-6, 1e-5
    assert np.isfinite(multi).all() and (err[0::2] <= tol_loss).all() and (err[1::2] <= tol_norm).all(), (one, multi)


def test_window_cut_short_by_new_seq():
    """train_flow.py:83-87: a new sequence resets the loss container, the recurrent state and the gradients in the middle
    of a window.  With the decoder halves and update() calls of the abandoned passes still in flight on the side stream,
    the reset has to wait for them: multi-stream against one stream (lr = 0, two passes abandoned, then two whole windows)."""
    assert torch.cuda.is_available()
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4)
    cfg["optimizer"]["lr"] = 0.0

    def run(streams):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=streams)
        tr.model.arch.engine.debug_delay = (0, 20_000_000, 0) if streams else None      # the side stream lags
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
        tr.reset()
        out = []
        for t in range(2 + 8):
            stepped = tr.step(src.next(), new_seq=(t == 2))
            if stepped:
                out += [float(tr.last_loss.item()), float(tr.last_grad_norm.item())]
        del tr, src
        return np.array(out)

    one, multi = run(False), run(True)
    print("cut-short window, one stream:", one, "multi-stream:", multi)
    assert len(one) == 4 and np.isfinite(multi).all()
    err = np.abs(multi - one) / np.abs(one)
    assert (err[0::2] <= 1e-6).all() and (err[1::2] <= 1e-5).all(), (one, multi)


def test_fused_adam_optimizer_surface_and_nan_norm():
    """parallel.FusedAdam as `Trainer.optimizer`: param_groups[0]["lr"] is what the step uses, state_dict /
    load_state_dict carry the moments and the step count, zero_grad clears the flat bucket, and a NaN gradient norm poisons
    every parameter like torch's clip_grad_norm_ (clamp propagates NaN) instead of applying the finite elements unclipped."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=1, resolution=[32, 32], max_num_grad_events=200)
    cfg["data"]["passes_loss"] = 2
    cfg["optimizer"]["lr"] = 1e-3
    torch.manual_seed(3)
    tr = train.Trainer(cfg, dev)
    opt = tr.optimizer
    assert opt is tr.fused_opt and opt.param_groups[0]["lr"] == 1e-3 and len(opt.param_groups[0]["params"]) == len(tr.bucket.params)
    p0 = tr.bucket.params[0]
    tr.bucket.flat.fill_(0.5)
    before = p0.detach().clone()
    opt.param_groups[0]["lr"] = 2e-3                       # what a scheduler does
    norm = opt.step(None)
    torch.cuda.synchronize()
    # first Adam step: every weight moves by lr * sign(g) (bias-corrected m / sqrt(v) = 1)
    assert torch.allclose(before - p0.detach(), torch.full_like(before, 2e-3), rtol=1e-4, atol=1e-8)
    assert float(tr.bucket.flat.abs().max()) == 0.0 and float(norm) > 0        # the step cleared the gradient
    sd = opt.state_dict()
    assert float(sd["state"]["step"]) == 1.0 and sd["param_groups"][0]["lr"] == 2e-3
    tr2 = train.Trainer(cfg, dev)
    tr2.optimizer.load_state_dict(sd)
    assert torch.equal(tr2.optimizer.m, opt.m) and torch.equal(tr2.optimizer.v, opt.v) and tr2.optimizer.lr == 2e-3
    tr.bucket.flat.fill_(1.0)
    opt.zero_grad()
    assert float(tr.bucket.flat.abs().max()) == 0.0
    with pytest.raises(ValueError):
        opt.zero_grad(set_to_none=True)
    # NaN norm with clipping on: torch's clamp(max_norm / (norm + 1e-6), max=1) is NaN -> every gradient, then every weight
    tr.bucket.flat.fill_(0.1)
    tr.bucket.flat[7] = float("nan")
    opt.step(5.0)
    torch.cuda.synchronize()
    assert bool(torch.isnan(tr.fused_opt.flat_p).all())
    tr.close()
    tr2.close()


def test_side_stream_update_after_one_node_fallback():
    """with the side stream on, a pass that falls back to ONE autograd node on the caller's stream (here: a
    parameter's .grad set to None by an external zero_grad(set_to_none=True)) leaves its flows on the caller's stream;
    update() on the side stream must wait for them.  The window's loss equals the one-stream trainer's."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"]["passes_loss"] = 4
    cfg["optimizer"]["lr"] = 0.0

    def run(streams):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=streams)
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=100)
        tr.reset()
        next(iter(tr.model.parameters())).grad = None          # the engine now takes the one-node path
        if streams:
            tr.model.arch.engine.debug_delay = (20_000_000, 0, 0)
        for _ in range(4):
            complete = tr._forward_update(src.next())
            if streams:
                assert tr.model.arch.engine.last_pass_split is False
        assert complete
        loss = float(tr.loss_function().item())
        tr.close()
        return loss

    one, multi = run(False), run(True)
    assert abs(one - multi) <= 1e-6 * abs(one), (one, multi)


def test_loss_workspace_lease():
    """A second evaluation of the same loss window while the first one's autograd graph is alive must not clobber the
    workspace the first backward reads (explicit lease, not sys.getrefcount)."""
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import synth
    from taming_event_flow_amd.loss.flow import Iterative

    dev = torch.device("cuda:0")
    B, H, W, P, F = 2, 32, 32, 4, 2
    rng = np.random.default_rng(0)
    win = synth.make_window(rng, B, H, W, P, F, 400, 100, sigma=1.5)
    cfg = {"loader": {"resolution": [H, W], "batch_size": B},
           "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False, "iterative_mode": "two"},
           "data": {"passes_loss": P, "scales_loss": 1}}
    L = Iterative(cfg, dev)
    flows = [[torch.tensor(win["flows"][t][i], device=dev, requires_grad=True) for i in range(F)] for t in range(P)]
    for t in range(P):
        L.update(flows[t], torch.tensor(win["ev"][t], device=dev), torch.tensor(win["pm"][t], device=dev),
                 torch.tensor(win["dev"][t], device=dev), torch.tensor(win["dpm"][t], device=dev))
    flat = [f for row in flows for f in row]
    l1 = L()
    ws1 = L._win.workspace
    l2 = L()                                               # first graph still alive: a fresh workspace
    assert L._win.workspace is not ws1
    g1 = torch.autograd.grad(l1, flat, retain_graph=False)
    g2 = torch.autograd.grad(l2, flat)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    del l1, l2, g1, g2
    ws3 = L._win.workspace
    l3 = L()                                               # both graphs are gone: the buffer is reused
    assert L._win.workspace is ws3
    del l3


def test_graph_teardown_leaves_no_late_writes():
    """Regression test of the round-3 / round-4 corruption: a captured multi-stream window is created,
    replayed and dropped together with its trainer, then a fresh one-stream trainer is built from the same seed — its
    parameters must be bit-identical to the first such trainer's every time.  When dropped graphs were DESTROYED, 6 % of these
    iterations found two weights of the new trainer's first convolution overwritten (one decremented by an ulp, one zeroed)
    by writes the destroyed hipGraph left behind.  The shipped path never destroys a graph: a closed window is parked in its
    trainer, a dying trainer retires its graphs (kept allocated until the process ends) — this test holds that line; the
    destroying path itself is exercised by test_destroying_retired_graphs_diagnostic below."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4)
    cfg["optimizer"].update(lr=0.0, capturable=True)

    def fresh_weights():
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=False)
        w = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).clone()
        tr.close()
        return w

    expected = fresh_weights()
    for it in range(8):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=True)
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
        tr.reset()
        win = tr.capture_window([src.next() for _ in range(4)], warmup=1)
        for _ in range(2):
            win()
        float(tr.last_loss.item())
        del win
        del tr, src
        got = fresh_weights()
        bad = (got != expected).nonzero().reshape(-1)
        assert bad.numel() == 0, (it, bad[:4].tolist(), expected[bad[:4]].tolist(), got[bad[:4]].tolist())


@pytest.mark.skipif(os.environ.get("TEF_RUN_DESTROY_DIAGNOSTIC", "0") != "1",
                    reason="opt-in diagnostic (TEF_RUN_DESTROY_DIAGNOSTIC=1, in a process of its own): destroying captured "
                           "multi-stream hipGraphs on this ROCm stack leaves late device-side writes (DESIGN 9d) — round 5 saw it "
                           "pass alone and take the interpreter down with a segmentation fault behind 21 other trainer tests")
def test_destroying_retired_graphs_diagnostic():
    """train.release_retired_graphs() — the only way to give retired graphs' memory back before the process ends — on the
    workload of the test above: a few rounds of (capture, replay, drop, release) followed by a fresh trainer whose parameters
    must still be what the seed makes them.  Not part of the contract (the default never destroys a graph); it tells when
    the work-around can go."""
    import copy

    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import train

    dev = torch.device("cuda:0")
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[64, 64], max_num_grad_events=1500)
    cfg["data"].update(passes_loss=4)
    cfg["optimizer"].update(lr=0.0, capturable=True)

    def fresh_weights():
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=False)
        w = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).clone()
        tr.close()
        return w

    expected = fresh_weights()
    for it in range(4):
        torch.manual_seed(7)
        tr = train.Trainer(cfg, dev, streams=True)
        src = train.SyntheticSequences(cfg, dev, 2000, seq_len=10 ** 9, seed=3, jitter=300)
        tr.reset()
        win = tr.capture_window([src.next() for _ in range(4)], warmup=1)
        win()
        float(tr.last_loss.item())
        win.close()
        tr.close()
        del win, tr, src
        assert train.retired_graph_count() > 0
        train.release_retired_graphs()
        assert train.retired_graph_count() == 0
        got = fresh_weights()
        bad = (got != expected).nonzero().reshape(-1)
        assert bad.numel() == 0, (it, bad[:4].tolist())
