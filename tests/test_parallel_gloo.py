"""World-size-2 gloo tests (CPU) of the DP plumbing: flat gradient bucket, all-reduce SUM == gradient of the
global batch for a loss that is a sum over samples, clip-after-reduce, lock-step new_seq flag."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _loss(model, x):
    return (model(x) ** 2).sum()        # a SUM over batch samples, like the CM loss (reference loss/flow.py:129)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taming_event_flow_amd import parallel

    torch.manual_seed(123)
    xg = torch.randn(8, 6)
    lo, hi = parallel.shard_range(8, rank, world)
    model = _model()
    bucket = parallel.FlatGradBucket(model.parameters())
    _loss(model, xg[lo:hi]).backward()
    local = bucket.flat.clone()
    # the two-piece reduction train.Trainer overlaps with the last weight-gradient reduction: [second layer | first layer]
    # slices of the flat buffer, the later one first — bit for bit what one all-reduce of the whole buffer gives
    split = bucket.offset_of(list(model.parameters())[2])
    assert 0 < split < bucket.flat.numel()
    bucket.all_reduce_range(split, bucket.flat.numel())
    bucket.all_reduce_range(0, split)
    pieces = bucket.flat.clone()
    bucket.flat.copy_(local)
    bucket.all_reduce_sum()
    reduced = bucket.flat.clone()
    assert torch.equal(pieces, reduced), "bucketed all-reduce differs from the single one"
    norm = bucket.clip_(0.5)
    flag = parallel.any_rank(rank == 1, torch.device("cpu"))
    noflag = parallel.any_rank(False, torch.device("cpu"))
    q.put((rank, local.numpy(), reduced.numpy(), float(norm), bucket.flat.clone().numpy(), flag, noflag))
    dist.destroy_process_group()


def test_dp_sum_equals_global_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    torch.manual_seed(123)
    xg = torch.randn(8, 6)
    model = _model()
    _loss(model, xg).backward()
    full = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    for rank, local, reduced, norm, clipped, flag, noflag in res:
        np.testing.assert_allclose(reduced, full, rtol=1e-5, atol=1e-6)          # SUM of shards == global gradient
        assert abs(norm - np.linalg.norm(full)) <= 1e-4 * np.linalg.norm(full)   # clip sees the GLOBAL norm
        np.testing.assert_allclose(np.linalg.norm(clipped), min(0.5, np.linalg.norm(full)), rtol=1e-4)
        assert flag is True and noflag is False                                   # lock-step new_seq
    np.testing.assert_allclose(res[0][1] + res[1][1], full, rtol=1e-5, atol=1e-6)
    assert not np.allclose(res[0][1], res[1][1])


def test_bucket_views_and_zero():
    from taming_event_flow_amd import parallel

    model = _model()
    bucket = parallel.FlatGradBucket(model.parameters())
    _loss(model, torch.randn(4, 6)).backward()
    assert bucket.flat.abs().sum() > 0
    for p in model.parameters():
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()
        assert p.grad.data_ptr() < bucket.flat.data_ptr() + bucket.flat.numel() * 4
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
    assert parallel.shard_range(64, 3, 8) == (24, 32)


def _worker4(rank, world, port, q):
    """The DP window's reduction order at four ranks: the later slice of the bucket first (its all-reduce runs beside the
    encoders' last weight-gradient reduction), then the earlier slice; clip and the update see the SAME numbers on every
    rank, and the sum of the shards is the gradient of the global batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taming_event_flow_amd import parallel

    torch.manual_seed(123)
    xg = torch.randn(8, 6)
    lo, hi = parallel.shard_range(8, rank, world)
    model = _model()
    bucket = parallel.FlatGradBucket(model.parameters())
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    out = []
    for step in range(3):                      # three windows: the replicas must stay identical through the updates
        bucket.zero()
        _loss(model, xg[lo:hi] + 0.1 * step).backward()
        split = bucket.offset_of(list(model.parameters())[2])
        w1 = dist.all_reduce(bucket.flat[split:], op=dist.ReduceOp.SUM, async_op=True)      # "communication stream"
        # ... the encoder half's last local work would run here ...
        w1.wait()
        bucket.all_reduce_range(0, split)
        bucket.clip_(0.5)
        opt.step()
        out.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone().numpy())
    flag = parallel.any_rank(rank == 3)
    q.put((rank, out, flag))
    dist.destroy_process_group()


def test_two_range_reduction_keeps_four_replicas_identical():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process run of the same three steps on the full batch
    torch.manual_seed(123)
    xg = torch.randn(8, 6)
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    ref = []
    for step in range(3):
        opt.zero_grad()
        _loss(model, xg + 0.1 * step).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        ref.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy())
    for rank, out, flag in res:
        assert flag is True
        for step in range(3):
            assert np.array_equal(out[step], res[0][1][step]), (rank, step)          # replicas: bit for bit
            np.testing.assert_allclose(out[step], ref[step], rtol=2e-5, atol=1e-6)    # = the global-batch run
