"""Data-parallel plumbing for the training loop (new functionality: the reference is single-process).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
The loss is a SUM over batch samples (reference loss/flow.py:129), so the single-process gradient of a global
batch equals the SUM of the per-rank gradients: one all-reduce(SUM) of a flat fp32 gradient bucket per loss
window, BEFORE clipping (the reference clips the global norm, train_flow.py:127-128).  The reference resets loss /
state / gradients for the whole batch when ANY slot starts a new sequence (train_flow.py:83-87); ranks therefore
agree on that flag with an all-reduce(MAX) so that their collectives stay matched.
"""

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class FlatGradBucket:
    """All parameter gradients as views of ONE contiguous buffer: zeroing is a single fill, the DP reduction a
    single collective (125.5 MB for RecEVFlowNet: per-link bound on xGMI, so one large message, not many small)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        p0 = self.params[0]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_sum(self):
        """SUM over ranks (not mean): reproduces the gradient of the global batch exactly."""
        if is_distributed():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        return self.flat

    def all_reduce_range(self, lo, hi):
        """The same for the elements [lo, hi) of the flat buffer (a contiguous view: one collective).  Two ranges that cover
        the buffer give bit for bit what all_reduce_sum gives on two ranks (a sum of two numbers has one rounding); on more
        ranks the reduction order of an element may depend on where the collective cuts its chunks."""
        if is_distributed() and hi > lo:
            dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
        return self.flat

    def offset_of(self, param):
        """First element of `param`'s gradient in the flat buffer (parameters are laid out in iteration order)."""
        o = 0
        for p in self.params:
            if p is param:
                return o
            o += p.numel()
        raise KeyError("parameter is not in the bucket")

    def clip_(self, max_norm):
        """Global-norm clipping on the (already reduced) flat buffer = clip_grad_norm_ (train_flow.py:127-128)."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm


class FusedAdam:
    """torch.optim.Adam (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) + clip_grad_norm_ +
    zero_grad on FLAT buffers, three launches of libtef_hip.so (tef_l2_norm, tef_adam_clip_step) instead of ~12 ATen
    launches over 60 tensors.  The parameters are re-pointed at views of one flat buffer (like the gradients of
    FlatGradBucket); the step counter lives on the device, so the step can be captured into a hipGraph.  The packed GEMM
    operands of the convolutions are keyed on the parameters' version counters, which a kernel behind autograd's back
    does not bump: the caller invalidates them after every step (models.submodules.invalidate_packed)."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8):
        import ctypes

        try:
            from . import _lib
        except ImportError:
            import _lib
        self._lib, self._ct = _lib, ctypes
        self.bucket = bucket
        # one parameter group, like torch.optim.Adam over model.parameters(): schedulers write param_groups[0]["lr"]
        self.param_groups = [{"params": list(bucket.params), "lr": float(lr), "betas": tuple(betas), "eps": float(eps)}]
        flat_g = bucket.flat
        self.flat_p = torch.empty_like(flat_g)
        o = 0
        for p in bucket.params:                      # parameters become views of the flat buffer (values kept)
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            o += p.numel()
        self.m = torch.zeros_like(flat_g)
        self.v = torch.zeros_like(flat_g)
        self.step_count = torch.zeros((1,), dtype=torch.float32, device=flat_g.device)
        self.norm = torch.zeros((1,), dtype=torch.float32, device=flat_g.device)
        self.scratch = torch.empty((_lib.lib().tef_l2_norm_scratch_bytes(),), dtype=torch.uint8, device=flat_g.device)
        # lr, beta1, beta2, eps, max_norm as the KERNEL reads them: device memory, refreshed from param_groups[0] whenever a
        # value changed (refresh_hyperparams).  A window captured in a hipGraph therefore follows a learning-rate schedule
        # or a load_state_dict without being captured again.
        self.hp = torch.zeros((5,), dtype=torch.float64, device=flat_g.device)
        self._hp_uploaded = None
        self._hp_ring = None         # pinned staging buffers of refresh_hyperparams
        self._max_norm = -1.0

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = float(value)

    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    def zero_grad(self, set_to_none=False):
        """Clear the flat gradient (the views stay: the kernels add into them).  set_to_none is refused: the pass engine
        accumulates into the parameters' own .grad buffers."""
        if set_to_none:
            raise ValueError("FusedAdam keeps the flat gradient bucket: zero_grad(set_to_none=True) is not supported")
        self.bucket.zero()

    def state_dict(self):
        """Adam's moments and step count (flat, in parameter order) + the group's hyper-parameters."""
        return {"state": {"exp_avg": self.m.detach().clone(), "exp_avg_sq": self.v.detach().clone(),
                          "step": self.step_count.detach().clone()},
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        st = sd["state"]
        if st["exp_avg"].numel() != self.m.numel() or st["exp_avg_sq"].numel() != self.v.numel():
            raise ValueError("FusedAdam.load_state_dict: moment buffers of a different parameter set")
        self.m.copy_(st["exp_avg"].reshape(-1))
        self.v.copy_(st["exp_avg_sq"].reshape(-1))
        self.step_count.copy_(st["step"].reshape(-1))
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v

    def refresh_hyperparams(self, max_norm=None):
        """Upload (lr, betas, eps, max_norm) if any of them changed since the last upload.  Called by step() outside graph
        capture and by CapturedWindow.replay() before the graph is launched — there with the window's OWN capture-time
        `max_norm`, so that an eager step with another clip between two replays cannot change the captured window's.  The
        values go through a small ring of pinned host buffers with a non-blocking copy: stream-ordered before whatever is
        enqueued next, and the host does not wait for the previous window to drain (a per-step learning-rate schedule would
        otherwise serialise host and device at every replay)."""
        g = self.param_groups[0]
        mn = self._max_norm if max_norm is None else float(max_norm)
        vals = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(mn))
        if vals != self._hp_uploaded:
            if self.hp.is_cuda:
                if self._hp_ring is None:
                    self._hp_ring = [(torch.empty((5,), dtype=torch.float64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
                    self._hp_next = 0
                buf, ev = self._hp_ring[self._hp_next % len(self._hp_ring)]
                if self._hp_next >= len(self._hp_ring):
                    ev.synchronize()          # (the copy that last used this slot, eight uploads ago)
                self._hp_next += 1
                for k_, v_ in enumerate(vals):
                    buf[k_] = v_
                self.hp.copy_(buf, non_blocking=True)
                ev.record()
            else:
                self.hp.copy_(torch.tensor(vals, dtype=torch.float64))
            self._hp_uploaded = vals

    def step(self, max_norm=None):
        """Clip the (already reduced) flat gradient to `max_norm` (None: no clipping), apply Adam, clear the gradient.
        -> the gradient's global norm before clipping (a 1-element device tensor, overwritten by the next step).  The
        hyper-parameters are read by the kernel from device memory (`hp`): param_groups[0] at the time of the call — or,
        for a captured window, at the time of each replay."""
        lib, n = self._lib.lib(), self.bucket.flat.numel()
        st = self._lib.stream_ptr()
        self._max_norm = -1.0 if max_norm is None else float(max_norm)
        if not (self.hp.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.refresh_hyperparams()      # (inside a capture: the values are uploaded before every replay instead)
        elif self._hp_uploaded is None or self._hp_uploaded[4] != self._max_norm:
            raise RuntimeError("FusedAdam.step: captured with a max_norm no eager step has uploaded (run a warm-up window)")
        self._lib.check(lib.tef_l2_norm(self.bucket.flat.data_ptr(), n, self.scratch.data_ptr(), self.norm.data_ptr(),
                                        self.step_count.data_ptr(), st), "tef_l2_norm")
        self._lib.check(lib.tef_adam_clip_step_hp(self.flat_p.data_ptr(), self.bucket.flat.data_ptr(), self.m.data_ptr(),
                                                  self.v.data_ptr(), n, self.norm.data_ptr(), self.hp.data_ptr(),
                                                  self.step_count.data_ptr(), st), "tef_adam_clip_step_hp")
        return self.norm.view(())


_FLAG_GROUP = None


def _flag_group():
    """Host-side (gloo) group for the one-int flag exchange.  With RCCL as the default backend the flag would have to
    travel through the GPU stream, and reading it back would drain the stream on every pass; a CPU group leaves the
    asynchronous launch queue alone.  Created on first use — by every rank together, since every rank exchanges the
    flag on every pass."""
    global _FLAG_GROUP
    if _FLAG_GROUP is None:
        _FLAG_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _FLAG_GROUP


class _FlagBoard:
    """The lock-step exchange for ranks of ONE node (the only DP layout of this package: one process per GPU of a node) without
    a collective: a shared-memory board of one (sequence number, mask) pair per rank and bank.  An exchange writes this
    rank's mask, then its sequence number, into bank `seq % 2`, spins until every rank's number in that bank has reached
    `seq`, and ORs the masks.  Nobody can be two exchanges ahead — exchange seq + 1 completes only when every rank has
    written seq + 1, i.e. has finished reading seq — so the two banks never collide.  Aligned 8-byte stores and loads, store
    order kept by the host's memory model (x86-64 TSO; the store of the number follows the store of the mask).  Measured
    with 8 CPU processes (tools/lockstep_timing.py): ~2 ms per gloo all-reduce of one int in the build container against a few
    microseconds here.  Falls back to the gloo group when the ranks are not on one host or shared memory is unavailable."""

    def __init__(self):
        import socket
        import uuid
        from multiprocessing import shared_memory

        import numpy as np

        grp = _flag_group()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        hosts = [None] * self.world
        dist.all_gather_object(hosts, socket.gethostname(), group=grp)
        if len(set(hosts)) != 1:
            raise RuntimeError("ranks on several hosts")
        name = [f"tef_flags_{uuid.uuid4().hex[:16]}" if self.rank == 0 else None]
        nbytes = 2 * self.world * 2 * 8
        shm, err = None, None
        if self.rank == 0:
            try:
                shm = shared_memory.SharedMemory(name=name[0], create=True, size=nbytes)
                shm.buf[:nbytes] = bytes(nbytes)
            except Exception as e:          # noqa: BLE001
                err, name[0] = repr(e), None
        dist.broadcast_object_list(name, src=0, group=grp)
        if name[0] is None:
            raise RuntimeError(f"shared memory unavailable: {err}")
        if self.rank != 0:
            shm = shared_memory.SharedMemory(name=name[0])
            try:      # (attaching registers the segment with this process's resource tracker, which would unlink it — again — at exit)
                from multiprocessing import resource_tracker

                resource_tracker.unregister(shm._name, "shared_memory")
            except Exception:              # noqa: BLE001
                pass
        self.shm = shm
        self.words = np.ndarray((2, self.world, 2), dtype=np.uint64, buffer=shm.buf)      # [bank][rank][(seq, mask)]
        self.seq = 0
        dist.barrier(group=grp)            # everybody is attached before rank 0 may unlink the name
        if self.rank == 0:
            try:
                shm.unlink()               # the mapping stays; the name goes away with the last process
            except Exception:              # noqa: BLE001
                pass

    def exchange_or(self, mask, timeout=600.0):
        import time

        self.seq += 1
        bank = self.words[self.seq & 1]
        bank[self.rank, 1] = int(mask)
        bank[self.rank, 0] = self.seq
        seqs = bank[:, 0]
        spins, t0 = 0, None
        while int(seqs.min()) < self.seq:
            spins += 1
            if spins > 2000:
                time.sleep(0)              # (oversubscribed hosts: let the rank we wait for run)
                if t0 is None:
                    t0 = time.monotonic()
                elif spins % 4096 == 0 and time.monotonic() - t0 > timeout:
                    raise RuntimeError("lock-step flag exchange timed out: a rank skipped an exchange or died")
        out = 0
        for v in bank[:, 1].tolist():
            out |= v
        return out

    def close(self):
        try:
            self.words = None
            self.shm.close()
        except Exception:                  # noqa: BLE001
            pass


_FLAG_BOARD = None          # False: tried and unavailable (the gloo group is used)


def _exchange_or(mask, nbits):
    """OR over ranks of an integer mask: the shared-memory board when the ranks share a host (TEF_FLAG_BOARD=0: never),
    one all-reduce(MAX) over the host-side gloo group otherwise."""
    global _FLAG_BOARD, _EXCHANGES
    import os

    _EXCHANGES += 1
    if _FLAG_BOARD is None:
        _FLAG_BOARD = False
        if os.environ.get("TEF_FLAG_BOARD", "1") != "0":
            ok = [1]
            try:
                board = _FlagBoard()
            except Exception:              # noqa: BLE001
                board, ok = None, [0]
            # (all ranks or none: a rank on its own would wait on the board for ranks that use the collective)
            t = torch.tensor(ok, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_flag_group())
            if int(t.item()) == 1:
                _FLAG_BOARD = board
            elif board is not None:
                board.close()
    if _FLAG_BOARD:
        return _FLAG_BOARD.exchange_or(mask)
    t = torch.tensor([(mask >> k) & 1 for k in range(nbits)], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_flag_group())
    return sum(int(v) << k for k, v in enumerate(t.tolist()))


def any_rank(flag, device=None):
    """True on every rank if `flag` is True on at least one (lock-step `new_seq`, reference train_flow.py:83-87 applied
    to the global batch).  EVERY rank must call this at the same point of every pass, whatever its own flag is: a rank
    that skipped the exchange would pair the other ranks' flag all-reduce with its next collective (the gradient
    all-reduce) and hang or corrupt it.  No collective (and no device sync) outside DP."""
    if not is_distributed():
        return bool(flag)
    return bool(_exchange_or(1 if flag else 0, 1))


_EXCHANGES = 0       # host-side flag exchanges this process took part in (tests / tools count them)


def any_rank_mask(mask, nbits):
    """The bitwise OR over ranks of an `nbits`-bit mask: ONE host-side exchange for a whole loss window (bit k = "a slot of
    this rank starts a new sequence at pass k of the window").  For loaders that know their sequence boundaries ahead —
    fixed-length sequences, e.g. the reference's dsec_train (200 passes per sequence, configs/train_flow.yml:6) — this
    replaces the per-pass any_rank: `Trainer.declare_fixed_sequences`.  Same rule as any_rank: EVERY rank calls it at the
    same point.  No collective outside DP."""
    mask = int(mask) & ((1 << nbits) - 1)
    if not is_distributed():
        return mask
    return _exchange_or(mask, nbits)


def reset_groups():
    """Forget the cached flag group (after destroy_process_group)."""
    global _FLAG_GROUP, _FLAG_BOARD
    _FLAG_GROUP = None
    if _FLAG_BOARD:
        _FLAG_BOARD.close()
    _FLAG_BOARD = None


def shard_range(global_batch, rank, world):
    """Batch slots [lo, hi) owned by `rank` (rank g owns slots [g*b, (g+1)*b))."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    b = global_batch // world
    return rank * b, (rank + 1) * b
