"""Batched loader stage on the MI355X (tef_collate_events + tef_encode_event_lists) vs the numpy restatement of the
reference loader (oracle/loader.py): bit-exact lists and masks, exact counts, voxel grids to fp32 rounding."""
import numpy as np
import pytest
import torch

from oracle import loader

pytestmark = pytest.mark.gpu


def _raw(rng, counts, H, W):
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
    n = int(offs[-1])
    xs = rng.integers(0, W, n).astype(np.float32)
    ys = rng.integers(0, H, n).astype(np.float32)
    ts = np.concatenate([np.sort(rng.random(c)) * 0.01 + 1000.0 * (b + 1) for b, c in enumerate(counts)] + [[]]).astype(np.float32)
    ps = rng.integers(0, 2, n).astype(np.float32)
    return xs, ys, ts, ps, offs


def _run(counts, H, W, G, flags, voxel, seed):
    from taming_event_flow_amd.dataloader.base import collate_raw_events

    rng = np.random.default_rng(seed)
    xs, ys, ts, ps, offs = _raw(rng, counts, H, W)
    sampled = None
    if G:
        sampled = np.full((len(counts), G), -1, np.int32)
        for b, c in enumerate(counts):
            if c > G:
                sampled[b] = rng.permutation(c)[:G]
    want = loader.collate_raw_events(xs, ys, ts, ps, offs, (H, W), G, flags, sampled, voxel)
    dev = torch.device("cuda:0")
    got = collate_raw_events(*(torch.tensor(a, device=dev) for a in (xs, ys, ts, ps)), offs, (H, W),
                             max_num_grad_events=G, augmentation=flags,
                             sampled_indices=None if sampled is None else torch.tensor(sampled), voxel=voxel)
    for k in ("event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask", "event_cnt", "event_mask"):
        g = got[k].cpu().numpy()
        assert g.shape == want[k].shape, (k, g.shape, want[k].shape)
        assert np.array_equal(g, want[k]), k
    g, w = got["net_input"].cpu().numpy(), want["net_input"]
    assert g.shape == w.shape
    if voxel is None:
        assert np.array_equal(g, w)
    else:
        np.testing.assert_allclose(g, w, rtol=1e-5, atol=1e-5)
    return got


@pytest.mark.parametrize("voxel", [None, 5])
def test_no_split(voxel):
    _run([300, 50, 1200, 11], 24, 30, None, None, voxel, 0)


@pytest.mark.parametrize("voxel", [None, 3])
def test_split_mixed_batch(voxel):
    # samples above and below the cap, one with <= 10 events (dropped), one empty, one exactly at the cap
    _run([5000, 100, 10, 0, 1000, 1001, 3333], 32, 40, 1000, [0, 1, 2, 4, 7, 3, 5], voxel, 1)


def test_reference_recorded_batch():
    """The HIP loader stage on the raw streams of tests/golden/loader.npz against what the reference's own functions
    produced for them (recorded by tests/golden/make_golden_loader.py): bit-exact lists and masks."""
    import os

    from conftest import GOLDEN
    from taming_event_flow_amd.dataloader.base import collate_raw_events

    z = np.load(os.path.join(GOLDEN, "loader.npz"))
    H, W, B, G = int(z["H"]), int(z["W"]), int(z["B"]), int(z["G"])
    offs = np.concatenate([[0], np.cumsum(z["counts"])]).astype(int)
    dev = torch.device("cuda:0")
    # the reference casts the float64 raw arrays to fp32 on the host first (base.py:164-167)
    cat = lambda n: torch.tensor(np.concatenate([z[f"{n}{b}"] for b in range(B)]).astype(np.float32), device=dev)   # noqa: E731
    sampled = np.full((B, G), -1, np.int32)
    for b in range(B):
        if f"sampled{b}" in z.files:
            sampled[b] = z[f"sampled{b}"]
    got = collate_raw_events(cat("xs"), cat("ys"), cat("ts"), cat("ps"), offs, (H, W), max_num_grad_events=G,
                             augmentation=[int(f) for f in z["flags"]], sampled_indices=torch.tensor(sampled))
    for k in ("event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask"):
        g = got[k].cpu().numpy()
        assert g.shape == z["col_" + k].shape and np.array_equal(g, z["col_" + k]), k


def test_all_empty():
    got = _run([0, 3, 10], 16, 16, 100, None, None, 2)
    assert got["event_list"].shape == (3, 0, 4) and got["d_event_list"].shape == (3, 0, 4)
    assert float(got["event_cnt"].abs().sum()) == 0.0


def test_dsec_sized_batch():
    # BASELINE configs[3]/[4]-like rates: 8 samples x up to 200 k events, 10 k gradient events each, 480 x 640
    got = _run([200000, 150000, 50000, 120000, 10000, 10001, 199999, 64], 480, 640, 10000, [0, 1, 2, 3, 4, 5, 6, 7], None, 3)
    assert got["event_list"].shape == (8, 10000, 4) and got["d_event_list"].shape == (8, 190000, 4)


def test_feeds_the_loss():
    """The collated batch is what Iterative.update consumes (train_flow.py:111-117)."""
    from taming_event_flow_amd.dataloader.base import collate_raw_events
    from taming_event_flow_amd.loss.flow import Iterative

    H = W = 32
    cfg = {"loader": {"resolution": [H, W], "batch_size": 2},
           "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False,
                    "iterative_mode": "two"},
           "data": {"passes_loss": 4, "scales_loss": 1}}
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    loss_fn = Iterative(cfg, dev)
    flows = []
    for _ in range(4):
        xs, ys, ts, ps, offs = _raw(rng, [700, 500], H, W)
        b = collate_raw_events(*(torch.tensor(a, device=dev) for a in (xs, ys, ts, ps)), offs, (H, W), max_num_grad_events=400)
        f = torch.randn(2, 2, H, W, device=dev, requires_grad=True)
        flows.append(f)
        loss_fn.update([f], b["event_list"], b["event_list_pol_mask"], b["d_event_list"], b["d_event_list_pol_mask"])
    loss = loss_fn()
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(f.grad).all() and f.grad.abs().sum() > 0 for f in flows)
