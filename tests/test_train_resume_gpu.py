"""GPU tests of the training loop's loose ends: captured windows that follow live hyper-parameters, the shared graph pool,
a sequence change in the middle of a captured window (reference train_flow.py:83-87), and checkpoint / resume through
utils.checkpoint + FusedAdam.state_dict (reference utils/utils.py:9-49,60-61)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import __graft_entry__ as g

    g.build()
    return torch.device("cuda:0")


def _cfg(P=3, lr=1e-4):
    from taming_event_flow_amd import train

    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=2, resolution=[32, 32], max_num_grad_events=300)
    cfg["data"]["passes_loss"] = P
    cfg["optimizer"].update(lr=lr, capturable=True)
    return cfg


def _fresh(cfg, dev, nwin, seed=3, streams=None):
    from taming_event_flow_amd import train

    torch.manual_seed(7)
    tr = train.Trainer(cfg, dev, streams=streams)
    src = train.SyntheticSequences(cfg, dev, 380, seq_len=10 ** 9, seed=seed)
    tr.reset()
    P = cfg["data"]["passes_loss"]
    return tr, [[src.next() for _ in range(P)] for _ in range(nwin)]


def _params(tr):
    return torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).clone()


def _clone(win):
    return [{k: v.clone() for k, v in b.items()} for b in win]


def test_captured_window_reads_live_hyperparameters(dev):
    """lr (and betas, eps, the clipping norm) reach the Adam kernel through device memory that replay() refreshes: a
    captured window follows a schedule or a restored optimiser state without being captured again."""
    cfg = _cfg()
    tr, wins = _fresh(cfg, dev, 1)
    cw = tr.capture_window(wins[0], warmup=1)
    cw.replay()
    p0 = _params(tr)
    tr.optimizer.param_groups[0]["lr"] = 0.0      # what a scheduler does
    cw.replay()
    p1 = _params(tr)
    assert torch.equal(p0, p1), "a replay at lr = 0 moved the parameters: the window still runs at its captured lr"
    tr.optimizer.param_groups[0]["lr"] = 1e-3
    cw.replay()
    p2 = _params(tr)
    d_small = (p2 - p1).abs().max().item()
    assert d_small > 0.0
    # Adam's step is ~lr per element whatever the gradient's size: ten times the rate, ten times the step
    tr.optimizer.param_groups[0]["lr"] = 1e-2
    cw.replay()
    d_big = (_params(tr) - p2).abs().max().item()
    assert 5.0 < d_big / d_small < 20.0, (d_small, d_big)
    # a restored optimiser state carries its own lr
    sd = tr.optimizer.state_dict()
    sd["param_groups"][0]["lr"] = 0.0
    tr.optimizer.load_state_dict(sd)
    p3 = _params(tr)
    cw.replay()
    assert torch.equal(p3, _params(tr))
    cw.close()


def test_recapture_of_the_same_shapes_retains_nothing_more(dev):
    """A closed window is parked in its trainer (graphs are never destroyed on this stack: DESIGN 9d) and handed out again
    by the next capture_window() of the same batch shapes: twenty capture / replay / close rounds and twenty lr changes
    retire no graph and leave the reserved memory where the second round left it; the windows keep training."""
    from taming_event_flow_amd import train

    cfg = _cfg()
    tr, wins = _fresh(cfg, dev, 1)
    n0 = train.retired_graph_count()
    reserved, losses = [], []
    for it in range(12):
        cw = tr.capture_window(_clone(wins[0]), warmup=1)
        tr.optimizer.param_groups[0]["lr"] = 1e-4 * (1 + it)
        before = _params(tr)
        cw.replay()
        losses.append(float(tr.last_loss.item()))
        assert not torch.equal(before, _params(tr))
        cw.close()
        del cw
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
    assert train.retired_graph_count() == n0
    assert reserved[-1] <= reserved[1] + (4 << 20), [r >> 20 for r in reserved]
    assert all(np.isfinite(losses)) and len(set(losses)) > 6
    # eager passes between two captures: the parked window picks up the trainer's state as it is then
    for b in _clone(wins[0]):
        tr.step(b)
    cw = tr.capture_window(_clone(wins[0]), warmup=1)
    assert np.isfinite(tr.warmup_losses[0])
    cw.close()
    tr.close()
    assert train.retired_graph_count() > n0      # (parked windows are retired with their trainer)


def test_mid_window_sequence_change_on_the_captured_path(dev):
    """train_flow.py:83-87 resets at ANY pass.  WindowRunner feeds a captured window pass by pass: a reset in the middle
    of a window drops the passes collected so far (the reference ran them and threw them away) and starts the window
    again at that pass — the same windows as the eager Trainer.step sees."""
    from taming_event_flow_amd import train

    cfg = _cfg(P=3)
    P = 3
    # passes: window A (3), then 2 passes of a window that a new sequence cuts short, then the new sequence's window (3),
    # then another full window
    flags = [False] * 3 + [False, False] + [True, False, False] + [False] * 3
    tr_e, wins = _fresh(cfg, dev, 4, streams=False)
    passes = [b for w_ in wins for b in w_][: len(flags)]
    losses_e = []
    for b, f in zip(_clone(passes), flags):
        if tr_e.step(b, new_seq=f):
            losses_e.append(float(tr_e.last_loss.item()))
    assert len(losses_e) == 3
    tr_g, _ = _fresh(cfg, dev, 4, streams=False)
    cw = tr_g.capture_window(_clone(passes[:P]), warmup=1)      # window A runs eagerly (once), then it is captured
    losses_g = [tr_g.warmup_losses[0]]
    runner = train.WindowRunner(cw)
    for b, f in zip(_clone(passes[P:]), flags[P:]):
        if runner.step(b, new_seq=f):
            losses_g.append(float(tr_g.last_loss.item()))
    assert len(losses_g) == 3
    assert abs(losses_g[0] - losses_e[0]) <= 1e-5 * abs(losses_e[0]), (losses_g, losses_e)
    assert abs(losses_g[1] - losses_e[1]) <= 1e-3 * abs(losses_e[1]), (losses_g, losses_e)
    assert abs(losses_g[2] - losses_e[2]) <= 2e-2 * abs(losses_e[2]), (losses_g, losses_e)
    # ... and the dropped passes really were dropped: without the reset the second window would be a different one
    tr_n, _ = _fresh(cfg, dev, 4, streams=False)
    l_noreset = []
    for b in _clone(passes[:9]):
        if tr_n.step(b, new_seq=False):
            l_noreset.append(float(tr_n.last_loss.item()))
    assert abs(losses_g[1] - losses_e[1]) < abs(losses_g[1] - l_noreset[1])
    cw.close()


def test_checkpoint_resume_continues_the_run(dev, tmp_path):
    """One window, save (model through utils.checkpoint.save_model, optimiser through FusedAdam.state_dict), restore
    into a FRESH trainer, second window: parameters and loss as in the uninterrupted run — bit for bit when the run itself
    is reproducible bit for bit (the weight-gradient epilogues add in arrival order: the test measures that first)."""
    from taming_event_flow_amd.utils import checkpoint

    cfg = _cfg(P=3, lr=1e-3)

    def run(interrupt):
        tr, wins = _fresh(cfg, dev, 2, streams=False)
        for b in _clone(wins[0]):
            tr.step(b)
        l1 = float(tr.last_loss.item())
        if interrupt:
            art = str(tmp_path / "run")
            checkpoint.save_model(tr.model, art)
            torch.save(tr.optimizer.state_dict(), str(tmp_path / "opt.pth"))
            states = [s.detach().clone() for s in tr.model.arch.states]
            tr.close()
            del tr
            torch.manual_seed(99)                       # different initial weights: everything must come from the files
            from taming_event_flow_amd import train

            tr = train.Trainer(cfg, dev, streams=False)
            tr.reset()
            tr.model, epoch = checkpoint.load_model(art, tr.model, dev)
            assert epoch == 0
            tr.optimizer.load_state_dict(torch.load(str(tmp_path / "opt.pth"), map_location=dev, weights_only=False))
            tr.model.arch.states = states               # (the reference keeps the recurrent state across windows of a sequence)
        for b in _clone(wins[1]):
            tr.step(b)
        out = (l1, float(tr.last_loss.item()), _params(tr))
        tr.close()
        return out

    # three uninterrupted runs give the run-to-run noise (float atomics in the weight-gradient epilogues add in arrival order;
    # ONE pair underestimates it often enough to fail one run in four: round 6), then the interrupted one
    a, b, b2, c = run(False), run(False), run(False), run(True)
    noise = max((a[2] - b[2]).abs().max().item(), (a[2] - b2[2]).abs().max().item(), (b[2] - b2[2]).abs().max().item())
    lnoise = max(abs(a[1] - b[1]), abs(a[1] - b2[1]), abs(b[1] - b2[1]))
    assert a[0] == c[0] or abs(a[0] - c[0]) <= 1e-6 * abs(a[0])
    if noise == 0.0 and lnoise == 0.0:
        assert c[1] == a[1] and torch.equal(c[2], a[2]), "the resumed run differs from the uninterrupted one"
    else:
        # The noise of this run is heavy-tailed: Adam divides by sqrt(v), so a gradient element near zero whose last bits
        # differ moves its parameter by up to 2 lr in one run and not at all in the next — the LARGEST difference of a few
        # pairs of runs says little about the next pair (this assertion failed one run in four on it, then one in ten).
        # What a broken restore looks like is different in kind: moments, weights or recurrent state not restored move
        # (nearly) EVERY parameter by about lr, and the loss in its leading digits.  So: the share of parameters that moved.
        def moved(x, y):
            return ((x - y).abs() > 1e-7 + 1e-5 * y.abs()).double().mean().item()

        share = max(moved(a[2], b[2]), moved(a[2], b2[2]), moved(b[2], b2[2]))
        assert abs(c[1] - a[1]) <= max(10 * lnoise, 1e-5 * abs(a[1]))
        assert moved(c[2], a[2]) <= max(5 * share, 1e-3), (moved(c[2], a[2]), share, noise)


def test_reference_format_state_dict_loads_through_load_model(dev, tmp_path):
    """A plain state-dict file with the reference's parameter names (the keys of tests/golden/model_32x32.npz, recorded from
    the reference's RecEVFlowNet) restores through utils.checkpoint.load_model and reproduces the reference's flows."""
    from taming_event_flow_amd import synth
    from taming_event_flow_amd.models.model import RecEVFlowNet
    from taming_event_flow_amd.utils import checkpoint

    z = np.load(os.path.join(GOLDEN, "model_32x32.npz"))
    torch.manual_seed(5)
    net = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2).to(dev)
    sd = net.state_dict()
    assert [k for k, _ in net.named_parameters()] == list(z["keys"])
    w = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], int(z["seed"]))
    path = tmp_path / "model" / "data" / "model.pth"
    os.makedirs(path.parent)
    torch.save({k: torch.tensor(v) for k, v in w.items()}, str(path))      # what a reference user's state_dict file holds
    net, _ = checkpoint.load_model(str(tmp_path), net, dev)
    net.train()
    flows = net(torch.tensor(z["x0"], device=dev))["flow"]
    for i, fl in enumerate(flows):
        ref = z[f"flow0_{i}"]
        err = np.abs(fl.detach().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        assert err <= 1e-4, (i, err)


def _dropin_window(model, loss_function, optimizer, cfg, batches, new_seq=False):
    """The reference's loop body, call for call (train_flow.py:83-87, :101-137), on this package's modules."""
    device = next(model.parameters()).device
    loss = gnorm = None
    for t, inputs in enumerate(batches):
        if new_seq and t == 0:
            loss_function.reset()
            model.reset_states()
            optimizer.zero_grad()
        x = model(inputs["net_input"].to(device))
        for i in range(len(x["flow"])):
            x["flow"][i] = x["flow"][i] * cfg["loss"]["flow_scaling"]
        loss_function.update(x["flow"], inputs["event_list"].to(device), inputs["event_list_pol_mask"].to(device),
                             inputs["d_event_list"].to(device), inputs["d_event_list_pol_mask"].to(device))
        if loss_function.num_passes >= cfg["data"]["passes_loss"]:
            loss = loss_function()
            loss_value = loss.item()
            loss.backward()
            if cfg["loss"]["clip_grad"] is not None:
                gnorm = torch.nn.utils.clip_grad.clip_grad_norm_(model.parameters(), cfg["loss"]["clip_grad"])
            optimizer.step()
            optimizer.zero_grad()
            model.detach_states()
            loss_function.reset()
    return loss_value, (None if gnorm is None else float(gnorm))


@pytest.mark.parametrize("trace", ["train_trace_lr1e-5"])
def test_literal_dropin_loop_reproduces_the_reference_trace(dev, trace):
    """No train.Trainer: torch.optim.Adam, clip_grad_norm_ and optimizer.zero_grad() (torch's set_to_none default) around this
    package's RecEVFlowNet and Iterative, exactly as train_flow.py wires them — against the reference's recorded two-window
    trace (loss, pre-clip gradient norm, per-parameter update norms).  The network keeps its own flat gradient buffer
    (arch.own_gradients) so this caller gets in-place parameter gradients and deferred weight gradients as well."""
    from taming_event_flow_amd import synth
    from taming_event_flow_amd.dataloader import encodings
    from taming_event_flow_amd.loss.flow import Iterative
    from taming_event_flow_amd.models.model import RecEVFlowNet

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
    model = RecEVFlowNet(cfg["model"].copy(), 2, key="flow").to(dev)
    w = synth.make_model_weights([(k, v.shape) for k, v in model.state_dict().items()], int(z["seed"]))
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model.train()
    loss_function = Iterative(cfg, dev)
    optimizer = torch.optim.Adam(model.parameters(), lr=cfg["optimizer"]["lr"])
    optimizer.zero_grad()
    for win in range(int(z["windows"])):
        before = [p.detach().clone() for p in model.parameters()]
        batches = []
        for t in range(P):
            ev, pm = torch.tensor(z[f"ev{win}_{t}"], device=dev), torch.tensor(z[f"pm{win}_{t}"], device=dev)
            dv, dpm = torch.tensor(z[f"dev{win}_{t}"], device=dev), torch.tensor(z[f"dpm{win}_{t}"], device=dev)
            batches.append({"net_input": encodings.event_list_to_channels(torch.cat([ev, dv], 1), (H, W)), "event_list": ev,
                            "event_list_pol_mask": pm, "d_event_list": dv, "d_event_list_pol_mask": dpm})
        loss, gn = _dropin_window(model, loss_function, optimizer, cfg, batches, new_seq=(win == 0))
        assert all(p.grad is None for p in model.parameters())       # (torch's zero_grad default: what the next window starts from)
        delta = np.array([float((p.detach() - b0).double().norm()) for p, b0 in zip(model.parameters(), before)])
        tol, gtol = (1e-4, 5e-4) if win == 0 else (1e-3, 1e-3)
        assert abs(loss - float(z[f"loss{win}"])) <= tol * abs(float(z[f"loss{win}"])), (win, loss)
        assert abs(gn - float(z[f"gnorm{win}"])) <= gtol * float(z[f"gnorm{win}"]), (win, gn)
        ref = z[f"delta{win}"]
        assert np.abs(delta - ref).max() <= 5e-2 * ref.max(), win
    assert model.arch.engine.lazy_flows          # (the flows stayed on the side stream: models/lazy.py)
    # the caller's set_to_none really is honoured between windows, and the pass stayed on its fast path
    assert model.arch._bucket is not None and model.arch.direct_grads and model.arch.deferred_wgrad
