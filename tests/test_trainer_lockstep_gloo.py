"""World-size-2 gloo test (CPU) of the Trainer's DP control flow (reference train_flow.py:83-87, 120-137 applied to a
global batch sharded over ranks): ranks disagree on `new_seq`, every rank still enters the flag exchange on every
pass, both reset together, the gradient all-reduce stays matched, and the weights stay identical across ranks.

The model and the loss are small CPU stand-ins with the interface Trainer uses (forward -> {"flow": [...]},
reset_states / detach_states; update / num_passes / reset / __call__): the subject is the Trainer, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

P = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    """One recurrent 1x1 'conv': state' = tanh(w * x + u * state); flow = state'."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([0.5, -0.3]).view(1, 2, 1, 1))
        self.u = torch.nn.Parameter(torch.tensor([0.2, 0.1]).view(1, 2, 1, 1))
        self.state = None
        self.resets = 0

    def forward(self, x):
        h = self.state if self.state is not None else torch.zeros_like(x)
        self.state = torch.tanh(self.w * x + self.u * h)
        return {"flow": [self.state]}

    def reset_states(self):
        self.state = None
        self.resets += 1

    def detach_states(self):
        self.state = self.state.detach()


class _Loss(torch.nn.Module):
    """Sum over samples (like the CM loss, reference loss/flow.py:129) of the squared flows of the window."""

    def __init__(self):
        super().__init__()
        self.flows = []
        self.resets = 0

    def update(self, flow_list, *event_lists):
        self.flows.append(flow_list[0])

    @property
    def num_passes(self):
        return len(self.flows)

    def reset(self):
        self.flows = []
        self.resets += 1

    def forward(self):
        return sum((f * f).sum() for f in self.flows)


def _cfg():
    return {"data": {"passes_loss": P, "voxel": None}, "model": {"name": "stub"},
            "loss": {"warping": "stub", "flow_scaling": 2.0, "clip_grad": 0.05},
            "optimizer": {"name": "SGD", "lr": 0.1}, "loader": {"batch_size": 2}}


def _batches(lo, hi):
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(4, 2, 3, 3, generator=g) for _ in range(3 * P)]
    empty = torch.zeros(hi - lo, 0, 4)
    return [{"net_input": x[lo:hi].clone(), "event_list": empty, "event_list_pol_mask": empty, "d_event_list": empty,
             "d_event_list_pol_mask": empty} for x in xs]


# pass index -> new_seq flag of (rank 0, rank 1): the ranks disagree, once in the middle of a window
FLAGS = {0: (True, True), 4: (False, True), 7: (True, False)}


def _run(tr, batches, rank):
    trace = []
    for k, b in enumerate(batches):
        flag = FLAGS.get(k, (False, False))[rank if rank is not None else 0]
        if rank is None:                 # single-process reference: any slot of the GLOBAL batch raises the flag
            flag = any(FLAGS.get(k, (False, False)))
        stepped = tr.step(b, new_seq=flag)
        trace.append((stepped, tr.loss_function.num_passes, tr.model.resets, tr.loss_function.resets))
    return trace


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taming_event_flow_amd import parallel, train

    lo, hi = parallel.shard_range(4, rank, world)
    tr = train.Trainer(_cfg(), torch.device("cpu"), model=_Net(), loss_function=_Loss())
    trace = _run(tr, _batches(lo, hi), rank)
    q.put((rank, trace, [p.detach().numpy().copy() for p in tr.model.parameters()],
           float(tr.last_grad_norm), float(tr.last_loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_lockstep_with_disagreeing_new_seq():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])     # a mismatched collective would hang
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the single-process run on the global batch (the semantics DP must reproduce)
    from taming_event_flow_amd import train

    tr = train.Trainer(_cfg(), torch.device("cpu"), model=_Net(), loss_function=_Loss())
    ref_trace = _run(tr, _batches(0, 4), None)
    ref_params = [p.detach().numpy() for p in tr.model.parameters()]
    (_, t0, p0, gn0, l0), (_, t1, p1, gn1, l1) = res
    assert t0 == t1 == ref_trace                       # same optimiser steps, same resets, on both ranks
    # resets happened on passes 0, 4 (mid-window: the partial window is dropped) and 7, although one rank said False
    assert [t[2] for t in t0] == [1, 1, 1, 1, 2, 2, 2, 3, 3]
    assert [t[0] for t in t0] == [False, False, True, False, False, False, True, False, False]
    for a, b, r in zip(p0, p1, ref_params):
        np.testing.assert_array_equal(a, b)            # ranks stay bit-identical (same reduced gradient, same update)
        np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-7)      # and equal the global-batch run
    assert gn0 == gn1 and abs(gn0 - float(tr.last_grad_norm)) <= 1e-5 * gn0     # clip saw the GLOBAL norm
    assert abs((l0 + l1) - float(tr.last_loss)) <= 1e-5 * abs(float(tr.last_loss))


# ---- world 8: the per-window flag exchange for loaders with fixed-length sequences (Trainer.declare_fixed_sequences) ----
SEQ_LEN, NPASS8 = 8, 5 * P          # sequences of 8 passes against windows of 3: restarts fall in the middle of windows


def _batches8(lo, hi):
    g = torch.Generator().manual_seed(12)
    xs = [torch.randn(8, 2, 3, 3, generator=g) for _ in range(NPASS8)]
    empty = torch.zeros(hi - lo, 0, 4)
    return [{"net_input": x[lo:hi].clone(), "event_list": empty, "event_list_pol_mask": empty, "d_event_list": empty,
             "d_event_list_pol_mask": empty} for x in xs]


def _first_pass(rank):
    return 4 * (rank % 2)            # odd ranks are four passes into their sequences: the ranks' boundaries differ


def _worker8(rank, world, port, q, board):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TEF_FLAG_BOARD"] = board
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taming_event_flow_amd import parallel, train

    cfg = _cfg()
    cfg["loader"]["batch_size"] = 1
    lo, hi = parallel.shard_range(8, rank, world)
    tr = train.Trainer(cfg, torch.device("cpu"), model=_Net(), loss_function=_Loss())
    tr.declare_fixed_sequences(SEQ_LEN, _first_pass(rank))
    trace = []
    n0 = parallel._EXCHANGES
    for k, b in enumerate(_batches8(lo, hi)):
        stepped = tr.step(b, new_seq=((_first_pass(rank) + k) % SEQ_LEN == 0))
        trace.append((stepped, tr.loss_function.num_passes, tr.model.resets))
    q.put((rank, trace, [p.detach().numpy().copy() for p in tr.model.parameters()], parallel._EXCHANGES - n0,
           bool(parallel._FLAG_BOARD)))
    dist.barrier()
    parallel.reset_groups()
    dist.destroy_process_group()


@pytest.mark.parametrize("board", ["1", "0"])
def test_per_window_flag_exchange_world8(board):
    """Eight ranks, one sample each, sequences of a DECLARED fixed length whose boundaries differ between ranks and fall in
    the middle of loss windows: one host-side exchange per P passes (over the shared-memory flag board, or over gloo with
    TEF_FLAG_BOARD=0) gives every rank the resets of the single-process run on the global batch with per-pass flags; the
    replicas stay bit-identical."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, board)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from taming_event_flow_amd import train

    cfg = _cfg()
    cfg["loader"]["batch_size"] = 8
    tr = train.Trainer(cfg, torch.device("cpu"), model=_Net(), loss_function=_Loss())
    ref = []
    for k, b in enumerate(_batches8(0, 8)):      # per-pass flags: any slot of the global batch (train_flow.py:83-87)
        stepped = tr.step(b, new_seq=any((_first_pass(r) + k) % SEQ_LEN == 0 for r in range(world)))
        ref.append((stepped, tr.loss_function.num_passes, tr.model.resets))
    ref_params = [p.detach().numpy() for p in tr.model.parameters()]
    assert any(t[0] for t in ref) and ref[-1][2] >= 4          # windows completed, and resets in the middle of windows
    for rank, trace, params, exchanges, used_board in res:
        assert trace == ref, rank
        assert exchanges == NPASS8 // P, (rank, exchanges)     # one exchange per P passes, not one per pass
        assert used_board == (board == "1")
        for a, b_, r_ in zip(params, res[0][2], ref_params):
            np.testing.assert_array_equal(a, b_)
            np.testing.assert_allclose(a, r_, rtol=1e-5, atol=1e-7)
