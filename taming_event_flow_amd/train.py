"""Training-loop counterpart of the reference's ``train_flow.py`` (:80-156) on the MI355X path.

Reproduces the reference's window semantics — reset on ``new_seq`` (any slot, all ranks), forward, ``x * flow_scaling``,
``loss.update``; every ``passes_loss`` passes: loss, backward, [DP all-reduce SUM], clip_grad_norm_(clip_grad),
Adam step, zero_grad, ``detach_states``, ``loss.reset`` — with a synthetic DSEC-shaped batch source instead of the
HDF5 loader (dataloader/h5.py needs h5py/cv2 and a dataset; out of scope).
"""

import numpy as np
import os
import sys

import torch

from . import parallel
from .dataloader import base as dl_base
from .loss.flow import Iterative, Linear  # noqa: F401  (selected by name like the reference's eval(...))
from .models import submodules
from .models.model import RecEVFlowNet  # noqa: F401

DEFAULT_CONFIG = {     # reference configs/train_flow.yml
    "data": {"passes_loss": 10, "scales_loss": 1, "voxel": None},
    "model": {"name": "RecEVFlowNet", "final_w_scale": 0.01},
    "loss": {"warping": "Iterative", "iterative_mode": "two", "round_ts": False, "flow_scaling": 32,
             "flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "clip_grad": 100.0},
    "optimizer": {"name": "Adam", "lr": 0.00001},
    "loader": {"batch_size": 8, "resolution": [128, 128], "max_num_grad_events": 10000, "seed": 0},
}


class SyntheticSequences:
    """Batch source with the reference loader's batch dict (dataloader/h5.py:413-431 + base.py:392-434 collate):
    net_input [B,2|bins,H,W], event_cnt, event_mask, event_list [B,N,4] (ts,y,x,p), event_list_pol_mask [B,N,2],
    d_event_list, d_event_list_pol_mask; `new_seq` raised every `seq_len` passes.  Each pass starts from raw event
    streams (pixel coordinates, raw timestamps, polarity bit) and goes through the batched device loader stage
    (dataloader/base.py::collate_raw_events: formatting, augmentation, split into at most `max_num_grad_events`
    gradient events + detached rest, zero-padded collate, input representation) — nothing is staged on the host."""

    def __init__(self, config, device, events_per_pass, seq_len=200, pool=4, seed=0, jitter=0):
        self.cfg, self.device = config, device
        self.B = config["loader"]["batch_size"]
        self.H, self.W = config["loader"]["resolution"]
        self.seq_len, self.t = seq_len, 0
        self.new_seq = True
        self.max_grad = config["loader"]["max_num_grad_events"]
        aug = config["loader"].get("augment") or []
        rng = np.random.default_rng(seed)
        gen = torch.Generator().manual_seed(seed)
        self.pool = []
        for k in range(pool):       # a small pool of pre-generated raw passes, cycled (host RNG is not the subject here)
            counts = [events_per_pass - int(rng.integers(0, jitter + 1)) for _ in range(self.B)]
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
            n = int(offs[-1])
            raw = {
                "xs": rng.integers(0, self.W, n).astype(np.float32),
                "ys": rng.integers(0, self.H, n).astype(np.float32),
                "ts": np.concatenate([np.sort(rng.random(c)) + float(k) for c in counts] + [[]]).astype(np.float32),
                "ps": rng.integers(0, 2, n).astype(np.float32),
            }
            flags = [sum(dl_base.AUG_BITS[m] for m in aug if rng.random() < 0.5) for _ in range(self.B)]
            sampled = dl_base.draw_sampled_indices(counts, self.max_grad, gen).to(device)
            self.pool.append(({k_: torch.tensor(v, device=device) for k_, v in raw.items()}, offs, flags, sampled))

    def next(self):
        raw, offs, flags, sampled = self.pool[self.t % len(self.pool)]
        batch = dl_base.collate_raw_events(raw["xs"], raw["ys"], raw["ts"], raw["ps"], offs, (self.H, self.W),
                                           max_num_grad_events=self.max_grad, augmentation=flags,
                                           sampled_indices=sampled, voxel=self.cfg["data"]["voxel"])
        self.new_seq = self.t % self.seq_len == 0
        self.t += 1
        return batch


_RETIRED_GRAPHS = []      # hipGraphs of closed CapturedWindows (see CapturedWindow.close)
def retired_graph_count():
    return len(_RETIRED_GRAPHS)


def release_retired_graphs():
    """Destroy the hipGraphs of closed windows (returns their private memory pools): wait for the device, destroy, wait again.
    On this stack a destroyed multi-stream graph can still write into the memory it owned a moment later (DESIGN section
    9d): call this when nothing else is about to allocate device or host memory — or never; the graphs die with the process."""
    live = torch.cuda.is_available() and torch.cuda.is_initialized()
    if live:
        torch.cuda.synchronize()
    _RETIRED_GRAPHS.clear()
    if live:
        torch.cuda.synchronize()


class CapturedWindow:
    """A loss window captured by Trainer.capture_window.  `inputs[t]` is pass t's batch dict of STATIC tensors: write
    the next window's data into them (`.copy_`), then `replay()`.  `new_seq` is the reference's reset flag for the
    window's first pass (train_flow.py:83-87); a captured window is P passes long by construction: for sequences that
    change in the middle of a window feed the passes through a `WindowRunner`."""

    def __init__(self, trainer, graph, graph_tail, inputs, states, graph_mid=None, signature=None, loss_out=None):
        self.trainer, self.graph, self.graph_tail, self.inputs, self.states = trainer, graph, graph_tail, inputs, states
        self.graph_mid = graph_mid        # DP with overlap: the encoder half of the last weight-gradient reduction
        self.signature, self.loss_out = signature, loss_out      # (what Trainer.capture_window parks a closed window under)
        # the clip the window was captured with (None: no clipping): uploaded with the other hyper-parameters at every replay,
        # whatever an eager step in between used
        self.max_norm = trainer.cfg["loss"]["clip_grad"] if trainer is not None else None
        # buffers the graphs read and write that nothing else refers to (the per-pass working copies of the inputs, allocated
        # BEFORE the capture, i.e. outside the graphs' private pool): they must live as long as the graphs can run — until
        # round 5 they died with capture_window()'s frame and the next allocation of the caller landed in them
        self.keep = None
        # a list: replay() appends (start, local work done, stop) events around each DP reduction — start -> stop is the
        # reduction as the main stream sees it, local work done -> stop the part of it nothing was left to hide
        self.allreduce_events = None

    def replay(self, new_seq=False, exchange=True):
        if (parallel.any_rank(new_seq) if exchange else new_seq):       # host-side exchange, outside the graph; all ranks reset together
            for s in self.states:            # loss containers and gradients are already clear at a window boundary
                s.zero_()
        if self.trainer.fused_opt is not None:
            # lr / betas / eps as they are NOW (a schedule, a restored checkpoint); the clip as it was captured
            self.trainer.fused_opt.refresh_hyperparams(max_norm=-1.0 if self.max_norm is None else float(self.max_norm))
        if self.loss_out is not None:
            self.trainer.last_loss = self.loss_out            # (the graph's own output tensor: eager windows in between re-point it)
        self.graph.replay()
        if self.graph_tail is not None:      # DP: the collectives run between the graphs, never captured
            ev = None
            if self.allreduce_events is not None:
                ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
                ev[0].record()
            if self.graph_mid is not None:
                # the decoder half's gradients are complete: their all-reduce runs on the communication stream beside the
                # encoder half's last weight-gradient reduction
                self.trainer._allreduce_tail_begin()
                self.graph_mid.replay()
                if ev:
                    ev[1].record()
                self.trainer._allreduce_head_and_join()
            else:
                if ev:
                    ev[1].record()
                self.trainer.bucket.all_reduce_sum()
            if ev:
                ev[2].record()
                self.allreduce_events.append(ev)
            self.graph_tail.replay()
        return self.trainer.last_loss

    __call__ = replay

    def close(self):
        """Wait for the last replay, release the hipGraphs (and with them their private memory pool), then WAIT AGAIN.
        Idempotent.  Dropping the object without calling this does the same from __del__.

        The second wait is the fix for round 3's "hidden heap corruption" (DESIGN section 9d): destroying a captured
        multi-stream hipGraph on this stack (ROCm 7.0 runtime of torch 2.10) leaves device-side work behind — after the
        destructor has returned, and although the device was idle before it, one 32-bit word inside memory the graph owned
        is decremented by one and another, 736 bytes on, is zeroed (the signature of a completion signal).  If that memory
        has been handed to a new allocation by then, the writes land there: observed in the first 4.6 KB of the NEXT
        trainer's parameter buffer (two weights of its first convolution changed, one by an ulp, one to 0.0: 6 % of the
        iterations of tools/interference_trace.py; 0 of 100 with this wait there but still 1 of 24 in the longer test
        sequence of tools/pytest_sequence_probe.py; 0 of 90 when graphs are never destroyed) and, in round 3, as glibc
        heap-corruption aborts when the recycled memory was the host's.  Hence the default below: retire, do not destroy."""
        if self.graph is None and self.graph_tail is None:
            return
        live = torch.cuda.is_available() and torch.cuda.is_initialized()
        if live:
            torch.cuda.synchronize()
        # A window closed while its trainer lives is PARKED there, graphs, pool and all: capture_window() for the same batch
        # shapes hands it out again instead of capturing (a retired graph pins its private pool — a whole window's memory —
        # for the life of the process; with this, what is retained is one window per distinct shape and trainer, and it is
        # retired once, with the trainer).  Hyper-parameters are read from device memory at every replay, the inputs are
        # static buffers the caller fills: a parked window is as good as a new capture — ONLY with the fused optimiser: any
        # torch.optim optimiser (TEF_TORCH_ADAM=1, or another name in the config) bakes lr / betas / eps into the captured
        # kernels, and such a window is retired as before (a new capture_window() records the values of its own time).
        tr = self.trainer
        if (tr is not None and tr.fused_opt is not None and self.signature is not None and getattr(tr, "_parked", None) is not None
                and self.signature not in tr._parked and os.environ.get("TEF_DESTROY_GRAPHS", "0") != "1"):
            tr._parked[self.signature] = {"graph": self.graph, "graph_tail": self.graph_tail, "graph_mid": self.graph_mid,
                                          "inputs": self.inputs, "states": self.states, "loss_out": self.loss_out,
                                          "keep": self.keep}
            self.graph = self.graph_tail = self.graph_mid = None
            self.inputs = self.states = self.keep = None
            return
        if os.environ.get("TEF_DESTROY_GRAPHS", "0") != "1":
            # default: the graphs are RETIRED, not destroyed — they (and their private memory pool) stay allocated until the
            # process ends or release_retired_graphs() is called.  The wait after destruction cuts the late writes from 6 %
            # of the probe's iterations to ~1 % (1 in 124), never destroying a graph to 0 in 90: correctness over memory.
            _RETIRED_GRAPHS.extend(g_ for g_ in (self.graph_tail, self.graph_mid, self.graph) if g_ is not None)
        self.graph_tail = None
        self.graph_mid = None
        self.graph = None
        self.inputs = self.states = self.keep = None
        if live:
            torch.cuda.synchronize()

    def __del__(self):
        try:                       # (a destructor must not raise — not even at interpreter shutdown, when `sys` may be gone)
            if not sys.is_finalizing():
                self.close()
        except Exception:          # noqa: BLE001
            pass


class WindowRunner:
    """Pass-by-pass front end of a captured window with the reference's reset semantics (train_flow.py:83-87: `new_seq` in
    ANY slot, at ANY pass, resets loss containers, recurrent states and gradients).  Passes are collected in the window's
    static input buffers; P of them are one replay.  A reset at pass k of a window discards the k passes collected so far
    — in the reference they were run and then thrown away by the reset: nothing of them survives but a log line — and the
    window starts again at this pass, with the recurrent state cleared at the replay.  Under DP every rank exchanges its
    flag at every pass (parallel.any_rank), exactly like Trainer.step."""

    def __init__(self, window):
        self.window, self.count, self.pending_reset = window, 0, False

    def step(self, inputs, new_seq=False):
        """-> True when this pass completed a window (an optimiser step happened)."""
        if parallel.any_rank(new_seq):
            self.count, self.pending_reset = 0, True
        dst = self.window.inputs[self.count]
        for k in dst:
            dst[k].copy_(inputs[k], non_blocking=True)
        self.count += 1
        if self.count < len(self.window.inputs):
            return False
        reset, self.count, self.pending_reset = self.pending_reset, 0, False
        if reset:      # (the flag has been exchanged already: clear the states here, replay without another exchange)
            for s_ in self.window.states:
                s_.zero_()
        self.window.replay(new_seq=False, exchange=False)
        return True


class Trainer:
    """model + loss + optimiser wired like reference train_flow.py:60-70, plus the DP gradient bucket."""

    def __init__(self, config, device, model=None, loss_function=None, streams=None, window_decode=None):
        self.cfg, self.device = config, device
        num_bins = 2 if config["data"]["voxel"] is None else config["data"]["voxel"]
        if model is None:
            model = eval(config["model"]["name"])(config["model"].copy(), num_bins, key="flow")
        self.model = model.to(device)
        self.model.train()
        self.loss_function = loss_function if loss_function is not None else eval(config["loss"]["warping"])(config, device)
        arch_own = getattr(self.model, "arch", None)
        if arch_own is not None and hasattr(arch_own, "auto_grads"):
            arch_own.auto_grads = False                 # (the trainer's bucket, its flushes, its streams)
        self.bucket = parallel.FlatGradBucket(self.model.parameters())
        submodules.enable_direct_grads(self.model)      # conv kernels add into the bucket's .grad views
        # weight gradients of a window's passes: one long reduction per layer after backward (opt out: TEF_NO_DEFERRED_WGRAD=1)
        self.deferred_wgrad = os.environ.get("TEF_NO_DEFERRED_WGRAD", "0") != "1"
        submodules.enable_deferred_wgrad(self.model, self.deferred_wgrad)
        opt_kwargs = {"lr": config["optimizer"]["lr"]}
        if config["optimizer"].get("capturable"):
            opt_kwargs["capturable"] = True      # optimiser state stays on the device: the window can live in a hipGraph
        # Adam with its defaults (the reference's configs/train_flow.yml) on a HIP device: clip + step + zero_grad as three
        # launches over the flat buffers (parallel.FusedAdam); anything else goes to torch.optim
        self.fused_opt = None
        extra = set(config["optimizer"]) - {"name", "lr", "capturable"}
        if (config["optimizer"]["name"] == "Adam" and not extra and torch.device(device).type == "cuda"
                and os.environ.get("TEF_TORCH_ADAM", "0") != "1"):
            self.fused_opt = parallel.FusedAdam(self.bucket, config["optimizer"]["lr"])
            # the optimiser-like surface callers reach for (param_groups[0]["lr"] for schedulers, state_dict /
            # load_state_dict for checkpoints, zero_grad); TEF_TORCH_ADAM=1 puts torch.optim.Adam here instead
            self.optimizer = self.fused_opt
        else:
            self.optimizer = getattr(torch.optim, config["optimizer"]["name"])(self.model.parameters(), **opt_kwargs)
        self.last_loss = None
        self.last_grad_norm = None
        self._fixed_seq, self._seq_pos, self._win_mask, self._win_pos = 0, 0, None, 0      # declare_fixed_sequences
        self._parked = {}       # closed captured windows by batch-shape signature (CapturedWindow.close)
        # DP: the gradient bucket is reduced in two pieces — [encoders | residual blocks, decoders, heads], the order of
        # model.parameters() — so that the second piece's all-reduce overlaps the encoders' last weight-gradient reduction
        # (TEF_DP_OVERLAP=0: one all-reduce of the whole bucket after everything)
        self.comm_stream = None
        self._dp_split = None
        self._dp_tail_started = False
        arch_ = getattr(self.model, "arch", None)
        if arch_ is not None and hasattr(arch_, "resblocks") and os.environ.get("TEF_DP_OVERLAP", "1") != "0":
            try:
                self._dp_split = self.bucket.offset_of(next(iter(arch_.resblocks.parameters())))
            except (KeyError, StopIteration):
                self._dp_split = None
        # Two streams per window (models/engine.py): the decoder half of pass t, and the loss container's update() behind
        # it, run on a side stream beside the encoders of pass t + 1; BPTT mirrors it through autograd's own stream
        # handling.  TEF_TWO_STREAMS=0 keeps every launch on one stream.
        self.dec_stream = self.wgrad_stream = None
        eng = getattr(getattr(self.model, "arch", None), "engine", None)
        if streams is None:      # (an explicit argument wins over the environment switch)
            streams = os.environ.get("TEF_TWO_STREAMS", "1") != "0"
        # Window mode (round 6; TEF_WINDOW_DECODE=0 opts out): the encoder halves run pass by pass, the decoder halves of the
        # whole window — independent of each other: only the recurrent states cross passes — as ONE batch of P x B samples
        # when the window is complete (models/engine.py decode_window): ten times fewer launches on GEMMs ten times
        # larger.  The loss container's update() calls follow it, the batches of the window stay referenced until then.
        self.window_decode = (eng is not None and torch.device(device).type == "cuda" and hasattr(self.model.arch, "encode")
                              and os.environ.get("TEF_WINDOW_DECODE", "1") != "0") if window_decode is None else bool(window_decode)
        self._win_inputs = []
        if self.window_decode and hasattr(self.loss_function, "defer_update"):
            # the P update() calls of a window follow its batched decoders back to back: packed in ONE launch when the loss is
            # evaluated (tef_update_window) instead of ten (the module refuses a list buffer it has already recorded: below)
            self.loss_function.defer_update = True
        if eng is not None and torch.device(device).type == "cuda" and streams:
            if not self.window_decode:       # (pass-by-pass decoders: on a side stream beside the next pass's encoders)
                self.dec_stream = torch.cuda.Stream(device=device)
                eng.side_stream = self.dec_stream
            # ... and a third one for the deferred weight gradients: every TEF_WGRAD_GROUP (default 3) finished backward
            # passes are reduced beside the rest of BPTT instead of all of them after it (0: after it)
            group = int(os.environ.get("TEF_WGRAD_GROUP", "3"))
            if self.deferred_wgrad and group > 0:
                self.wgrad_stream = torch.cuda.Stream(device=device)
                eng.wgrad_stream, eng.wgrad_group = self.wgrad_stream, group
            # Window mode: the encoder levels of consecutive passes pipelined over two streams (models/engine.py encode_pass:
            # the lower two levels of pass t + 1 beside the upper two of pass t, forward and backward; TEF_ENC_PIPELINE=0:
            # the encoder half of a pass as one call on the caller's stream)
            if self.window_decode and os.environ.get("TEF_ENC_PIPELINE", "1") != "0":
                eng.enc_streams = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))

    def close(self):
        """Wait for every stream the trainer launched on and release the engine's device buffers in a fixed order.
        Idempotent; also runs from __del__, so that a trainer dropped with work in flight (or collected late) never has
        its streams' buffers released under a running kernel."""
        eng = getattr(getattr(self.model, "arch", None), "_engine", None) if getattr(self, "model", None) is not None else None
        parked, self._parked = getattr(self, "_parked", None) or {}, None
        for rec in parked.values():       # parked windows die with their trainer: retired, like any closed window's graphs
            _RETIRED_GRAPHS.extend(g_ for g_ in (rec["graph_tail"], rec["graph_mid"], rec["graph"]) if g_ is not None)
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            for st in (self.dec_stream, self.wgrad_stream, self.comm_stream):
                if st is not None:
                    st.synchronize()
            torch.cuda.current_stream().synchronize()
        if eng is not None:
            eng.close()
            eng.side_stream = eng.wgrad_stream = None
        self.dec_stream = self.wgrad_stream = None

    def __del__(self):
        try:                       # (a destructor must not raise — not even at interpreter shutdown, when `sys` may be gone)
            if not sys.is_finalizing():
                self.close()
        except Exception:          # noqa: BLE001
            pass

    def reset(self):
        """train_flow.py:83-87"""
        if self.dec_stream is not None:      # (a window cut short: its decoder halves / updates may still be running)
            torch.cuda.current_stream().wait_stream(self.dec_stream)
        if self._win_inputs:                 # window mode: the passes collected so far are dropped with the window
            self._win_inputs = []
            self.model.arch.drop_window()
        self.loss_function.reset()
        self.model.reset_states()
        self.bucket.zero()

    def capture_window(self, batches, warmup=2):
        """Capture one whole loss window (P passes, loss, BPTT backward, [DP all-reduce], clip, optimiser step) into a
        hipGraph: ~10^3 kernel launches become one graph launch, so the window is device-bound instead of host-bound.
        Returns a `CapturedWindow`: the caller writes each new window's batch tensors into `.inputs` (static buffers,
        same shapes as `batches`) and calls `.replay(new_seq)`.  Under DP the window is TWO graphs around the gradient
        all-reduce — [P passes, loss, BPTT backward] -> RCCL all-reduce(SUM), enqueued eagerly on the same stream ->
        [clip, optimiser step, state hand-over] — so no collective is ever stream-captured; the lock-step `new_seq`
        flag is exchanged on the host before the first graph is launched."""
        P = self.cfg["data"]["passes_loss"]
        assert len(batches) == P
        if warmup < 1:
            raise ValueError("capture needs at least one eager window (allocations, recurrent state buffers)")
        if self.loss_function.num_passes != 0 or self._win_inputs:
            raise RuntimeError("capture_window must start at a window boundary")
        # everything else that is baked into the graphs at capture time: the optimiser object, the clip, how the DP
        # reduction is split, the loss configuration (a parked window is handed out again only when all of it still holds)
        lcfg = self.cfg["loss"]
        signature = (warmup >= 1, parallel.is_distributed(), id(self.optimizer), lcfg.get("clip_grad"),
                     self._dp_overlap() is not None, os.environ.get("TEF_DP_OVERLAP", "1"),
                     tuple(sorted((k, repr(v)) for k, v in lcfg.items())), repr(self.cfg["data"].get("passes_loss")),
                     repr(self.cfg["data"].get("scales_loss")),
                     tuple(tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(b.items())) for b in batches))
        rec = self._parked.pop(signature, None) if self._parked is not None else None
        if rec is not None:
            # a window of these shapes was captured before and closed: take it out of the park.  The `warmup` windows run
            # as replays (they are training steps either way); the recurrent state carries on from where the trainer is.
            cw = CapturedWindow(self, rec["graph"], rec["graph_tail"], rec["inputs"], rec["states"], rec["graph_mid"],
                                signature, rec["loss_out"])
            cw.keep = rec["keep"]
            for dst, src in zip(cw.inputs, batches):
                for k in dst:
                    dst[k].copy_(src[k])
            cur = self.model.arch.states
            for dst, src in zip(cw.states, cur):
                if src is None:
                    dst.zero_()
                elif dst is not src:
                    dst.copy_(src.detach())
            self.model.arch.states = cw.states
            self.warmup_losses = []
            for _ in range(warmup):
                cw.replay()
                self.warmup_losses.append(float(self.last_loss.item()))
            return cw
        inputs = [{k: v.clone() for k, v in b.items()} for b in batches]          # public static input buffers
        # update() shifts the time stamps of the two event lists in place (loss/flow.py:457-458): those two get a working copy
        # per replay, everything else is read from the static buffers directly (round 6: 70 -> 20 copy launches per window)
        inplace = ("event_list", "d_event_list")
        work = [{k: (torch.empty_like(v) if k in inplace else inputs[t][k]) for k, v in b.items()} for t, b in enumerate(batches)]
        # In window mode a CAPTURED window keeps its weight-gradient reductions on the capture stream (round 6, measured: the
        # window is bound by the sum of its kernels, and a replayed graph gains nothing from the third stream — 31.9 ms
        # against 32.4 with groups of three passes on it, 30.7 against 31.4 with the pipelined encoder levels; the eager
        # window, whose launches the host feeds, keeps the stream)
        eng_ = getattr(getattr(self.model, "arch", None), "_engine", None)
        saved_wgrad = None
        if self.window_decode and eng_ is not None and eng_.wgrad_stream is not None:
            saved_wgrad = (eng_.wgrad_stream, eng_.wgrad_group, self.wgrad_stream)
            torch.cuda.current_stream().wait_stream(eng_.wgrad_stream)
            eng_.wgrad_stream, eng_.wgrad_group, self.wgrad_stream = None, 0, None
        try:
            return self._capture(batches, warmup, inputs, work, signature)
        finally:
            if saved_wgrad is not None:
                eng_.wgrad_stream, eng_.wgrad_group, self.wgrad_stream = saved_wgrad

    def _capture(self, batches, warmup, inputs, work, signature):
        P = self.cfg["data"]["passes_loss"]

        split = parallel.is_distributed()

        overlap = split and self._dp_overlap() is not None

        # the working copies of the whole window in one multi-tensor copy (the static buffers hold every pass before the replay)
        pairs = [(dst[k], src[k]) for src, dst in zip(inputs, work) for k in src if dst[k] is not src[k]]

        def run_head(stage=0):   # everything up to the local gradient (stage 1: without the encoder half's last reduction)
            if pairs:
                torch._foreach_copy_([d_ for d_, _ in pairs], [s_ for _, s_ in pairs])
            for dst in work:
                complete = self._forward_update(dst)
            assert complete
            loss = self.loss_function()
            loss.backward()
            self._flush_wgrads(stage=stage)
            self.last_loss = loss.detach()

        def run():
            run_head()
            self.all_reduce_gradients()
            self._apply_update()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.warmup_losses = []
        with torch.cuda.stream(side):
            for _ in range(warmup):
                run()
                self.warmup_losses.append(self.last_loss)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_losses = [float(x.item()) for x in self.warmup_losses]
        # the recurrent state lives in static buffers: the graph reads them at its start and writes the detached
        # end-of-window state back, so consecutive replays carry the state exactly like the eager loop
        static_states = [s.detach().clone() for s in self.model.arch.states]
        self.model.arch.states = static_states
        def tail():
            self._apply_update()
            torch._foreach_copy_(list(static_states), [src.detach() for src in self.model.arch.states])

        graph, graph_tail, graph_mid = torch.cuda.CUDAGraph(), None, None
        with torch.cuda.graph(graph):
            run_head(1 if overlap else 0)
            if not split:
                tail()
        if overlap:
            graph_mid = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_mid, pool=graph.pool()):
                self._flush_wgrads(stage=2)
        if split:
            graph_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_tail, pool=graph.pool()):
                tail()
        self.model.arch.states = static_states
        cw = CapturedWindow(self, graph, graph_tail, inputs, static_states, graph_mid, signature, self.last_loss)
        cw.keep = work
        return cw

    def declare_fixed_sequences(self, seq_len, first_pass=0):
        """The loader's sequences are all `seq_len` passes long (the reference's dsec_train: 2 s = 200 passes,
        configs/train_flow.yml:6) and the next pass handed to step() is pass `first_pass` of its sequence.  Under DP the
        ranks then agree on the reset flags ONCE PER LOSS WINDOW instead of once per pass: at a window's first pass every
        rank contributes the passes of the coming window at which ITS sequences restart (from its own pass counter) and
        takes the OR over ranks (parallel.any_rank_mask); the host-side exchange — 10 blocking gloo all-reduces per window
        otherwise, DESIGN section 7 — happens once every P passes.  A loader flag that the declared length does not predict
        raises.  `seq_len` None / 0: back to the per-pass exchange."""
        self._fixed_seq = int(seq_len) if seq_len else 0
        self._seq_pos = int(first_pass)
        self._win_mask, self._win_pos = None, 0

    def _lockstep_flag(self, new_seq):
        """The reset flag of this pass as every rank sees it (train_flow.py:83-87 applied to the global batch)."""
        if not getattr(self, "_fixed_seq", 0):
            return parallel.any_rank(new_seq)      # every rank exchanges its flag on every pass (no-op outside DP)
        P = self.cfg["data"]["passes_loss"]
        if self._win_mask is None:             # once every P passes, whatever the window boundaries are
            mine = 0
            for k in range(P):
                if (self._seq_pos + k) % self._fixed_seq == 0:
                    mine |= 1 << k
            self._win_mask, self._win_pos = parallel.any_rank_mask(mine, P), 0
        predicted = (self._seq_pos % self._fixed_seq) == 0
        if bool(new_seq) != predicted:
            raise RuntimeError(f"the loader's new_seq flag ({bool(new_seq)}) disagrees with the declared sequence length "
                               f"{self._fixed_seq} at pass {self._seq_pos} of the sequence")
        flag = bool((self._win_mask >> self._win_pos) & 1)
        self._win_pos += 1
        self._seq_pos += 1
        if self._win_pos >= P:
            self._win_mask = None              # the P passes the exchange covered are used up: the next pass exchanges again
        return flag

    def step(self, inputs, new_seq=False):
        """One pass (train_flow.py:83-137).  Returns True when an optimiser step happened."""
        if self._lockstep_flag(new_seq):
            self.reset()
        return self._pass(inputs)

    def _pass(self, inputs):
        if not self._forward_update(inputs):
            return False
        self._backward_window()
        self.all_reduce_gradients()                     # DP: gradient of the global batch (sum of shards)
        self._apply_update()
        return True

    def _forward_update(self, inputs):
        """train_flow.py:101-118: forward, scale, hand the pass to the loss.  True when the window is complete."""
        cfg = self.cfg
        # flow_scaling (train_flow.py:107-108) rides on the pass's final up-sampling kernel when the model offers it
        arch = getattr(self.model, "arch", None)
        if self.window_decode and torch.is_grad_enabled() and arch.direct_grads:
            P = cfg["data"]["passes_loss"]
            arch.encode(inputs["net_input"])
            # the event lists are read when the window is complete: a loader that writes every pass into ONE buffer would have
            # them all read that buffer's last contents
            for k in ("event_list", "d_event_list"):
                t_ = inputs[k]
                if isinstance(t_, torch.Tensor) and t_.numel() and any(b_[k].data_ptr() == t_.data_ptr() for b_ in self._win_inputs
                                                                        if isinstance(b_[k], torch.Tensor) and b_[k].numel()):
                    raise RuntimeError(f"train.Trainer (window mode): the {k} of this pass lives in the storage of an earlier pass of "
                                       "the window; the batches of a window must stay alive and unchanged until its last pass — hand "
                                       "in a tensor per pass, or Trainer(..., window_decode=False)")
            self._win_inputs.append(inputs)
            if len(self._win_inputs) < P:
                return False
            # the window is complete: the decoder halves of all its passes as ONE batch, then their update() calls, on the
            # caller's stream.  (Measured, round 6: in chunks of 5 passes on a side stream beside the later passes' encoders the
            # window is 0.2-0.6 ms SLOWER than as one batch here — the window is bound by the sum of its kernels now, and
            # two half-size batches are less efficient than one.)
            arch.flow_scale = float(cfg["loss"]["flow_scaling"])
            try:
                flows_all = arch.decode_window()
            finally:
                arch.flow_scale = 1.0
            batches, self._win_inputs = self._win_inputs, []
            for flows, b in zip(flows_all, batches):
                self.loss_function.update(flows, b["event_list"], b["event_list_pol_mask"], b["d_event_list"],
                                          b["d_event_list_pol_mask"])
            return self.loss_function.num_passes >= P
        if hasattr(arch, "flow_scale"):
            arch.flow_scale = float(cfg["loss"]["flow_scaling"])
            if self.dec_stream is not None:
                arch.engine.defer_join = True      # (only here: anyone else calling the model gets joined flows)
            try:
                flows = self.model(inputs["net_input"])["flow"]
            finally:
                arch.flow_scale = 1.0
                if self.dec_stream is not None:
                    arch.engine.defer_join = False
        else:
            flows = [f * cfg["loss"]["flow_scaling"] for f in self.model(inputs["net_input"])["flow"]]
        if self.dec_stream is None:
            self.loss_function.update(flows, inputs["event_list"], inputs["event_list_pol_mask"], inputs["d_event_list"],
                                      inputs["d_event_list_pol_mask"])
            return self.loss_function.num_passes >= cfg["data"]["passes_loss"]
        # the flows are on the side stream (behind this pass's decoders): so is the container's update; the window's loss
        # (and everything after it) waits for the side stream once, when the window is complete
        for k in ("event_list", "event_list_pol_mask", "d_event_list", "d_event_list_pol_mask"):
            if isinstance(inputs[k], torch.Tensor) and inputs[k].is_cuda:
                inputs[k].record_stream(self.dec_stream)       # (the caller may drop them before the side stream read them)
        if not arch.engine.last_pass_split:
            # the pass fell back to one node on THIS stream (a parameter without a contiguous .grad, e.g. after an external
            # zero_grad(set_to_none=True)): the flows and the loader's tensors are produced here, the side stream must see them
            self.dec_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.dec_stream):
            self.loss_function.update(flows, inputs["event_list"], inputs["event_list_pol_mask"], inputs["d_event_list"],
                                      inputs["d_event_list_pol_mask"])
        complete = self.loss_function.num_passes >= cfg["data"]["passes_loss"]
        if complete:
            torch.cuda.current_stream().wait_stream(self.dec_stream)
        return complete

    def _backward_window(self):
        """train_flow.py:120-125: loss over the window, BPTT backward (local shard of the batch)."""
        loss = self.loss_function()
        loss.backward()
        self._flush_wgrads(stage=0)
        self.last_loss = loss.detach()

    def _dp_overlap(self):
        eng = getattr(getattr(self.model, "arch", None), "_engine", None)
        return (eng if (self._dp_split and self.deferred_wgrad and parallel.is_distributed()
                        and torch.device(self.device).type == "cuda") else None)

    def _flush_wgrads(self, stage=0):
        """The window's deferred weight gradients.  Outside DP (or with TEF_DP_OVERLAP=0): all layers, then wait for the
        reductions that ran beside BPTT.  Under DP the last reduction is split: the decoder half first, then `between`
        (eager: start that half's all-reduce on the communication stream), then the encoder half.  A captured window
        records the two halves in two graphs (stage 1, stage 2) with the collective between them."""
        eng_ = getattr(getattr(self.model, "arch", None), "_engine", None)
        if eng_ is not None:
            eng_.join_encoders()      # (window mode: the encoder halves' backward ran on the level streams)
        if not self.deferred_wgrad:
            return
        def tail_done():       # every reduction has been issued: wait for the ones that ran beside BPTT on the side stream
            if self.wgrad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)

        eng = self._dp_overlap()
        if eng is None:
            if stage in (0, 1):
                submodules.flush_deferred_wgrads()
                tail_done()
            return

        def layer_path_first():
            # Weight gradients queued by layer-by-layer modules (not the fused pass: normally none) may belong to parameters of
            # the decoder slice: they go in BEFORE that slice's all-reduce starts — added behind it they would be lost to the
            # other replicas without an error.
            for pk in list(submodules._DEFERRED):
                submodules.flush_deferred_wgrads(pk)

        if stage == 0:         # eager
            def between():
                tail_done()
                self._allreduce_tail_begin()
            layer_path_first()
            eng.flush_window(parts=(2, 1), between=between)
            submodules.flush_deferred_wgrads()      # (other engines: normally nothing)
        elif stage == 1:       # captured: decoder half, the backward calls stay queued
            layer_path_first()
            eng.flush_window(parts=(2,), keep=True)
            tail_done()
        else:                  # captured: encoder half
            eng.flush_window(parts=(1,))
            submodules.flush_deferred_wgrads()

    def _allreduce_tail_begin(self):
        """all-reduce(SUM) of the decoder half's slice of the bucket on the communication stream, behind the current one."""
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            self.bucket.all_reduce_range(self._dp_split, self.bucket.flat.numel())
        self._dp_tail_started = True

    def _allreduce_head_and_join(self):
        """The encoder half's slice on the current stream, then wait for the other half."""
        self.bucket.all_reduce_range(0, self._dp_split)
        torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._dp_tail_started = False

    def all_reduce_gradients(self):
        """DP: SUM of the shards' gradients (= the gradient of the global batch), before clipping."""
        if self._dp_tail_started:
            self._allreduce_head_and_join()
        else:
            self.bucket.all_reduce_sum()

    def _apply_update(self):
        """train_flow.py:127-137 on the (already reduced) flat gradient: clip, step, clear, cut the graph."""
        cfg = self.cfg
        if self.fused_opt is not None:
            self.last_grad_norm = self.fused_opt.step(cfg["loss"]["clip_grad"])      # clip + Adam + zero_grad
            submodules.invalidate_packed(self.model)
        else:
            if cfg["loss"]["clip_grad"] is not None:
                self.last_grad_norm = self.bucket.clip_(cfg["loss"]["clip_grad"])
            self.optimizer.step()
            self.bucket.zero()                          # optimizer.zero_grad() keeping the flat views
        self.model.detach_states()
        self.loss_function.reset()
