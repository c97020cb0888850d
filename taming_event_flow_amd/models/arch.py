"""Layer table and parameter container of the recurrent multi-resolution U-Net behind RecEVFlowNet.

Counterpart of the reference's ``models/arch.py`` (``MultiResUNetRecurrent``, :197-242, on ``BaseUNet``, :6-194), built
the other way round: the reference assembles the network from generic builder methods and walks module lists at run
time; here the architecture is a small static TABLE (`NetPlan`: one row per convolution with its channel counts,
stride, activation and the level it lives on) from which both the parameter container and the fused pass
(`engine.PassEngine`) are derived.  Only the configuration RecEVFlowNet uses is representable — recurrent ConvGRU
encoders, sum skip connections, bilinear up-sampling decoders, one prediction per decoder — anything else is refused
when the plan is made, not silently approximated.

The module tree keeps the reference's attribute names (``encoders[i].conv.conv2d``, ``encoders[i].recurrent_block.
{reset,update,out}_gate``, ``resblocks[i].conv{1,2}``, ``decoders[i].conv2d``, ``preds[i].conv2d``) because they are
the ``state_dict`` keys checkpoints are exchanged by (reference utils/utils.py:20-31).
"""

import os
from collections import namedtuple

import torch.nn as nn

from .engine import PassEngine, decode_passes, encode_pass, run_pass
from .submodules import ConvLayer, RecurrentConvLayer, ResidualBlock, UpsampleConvLayer, upsample_bilinear

Row = namedtuple("Row", "kind level cin cout ksize stride act")


class NetPlan:
    """Static description of the network: `rows` lists every convolution (in state_dict order)."""

    def __init__(self, num_bins, base_channels=64, num_encoders=4, num_residual_blocks=2, num_output_channels=2,
                 skip_type="sum", norm=None, use_upsample_conv=True, kernel_size=3, encoder_stride=2, channel_multiplier=2,
                 activations=("relu", None), final_activation=None, final_bias=True, final_w_scale=None,
                 recurrent_block_type="convgru", min_size=None):
        unsupported = []
        if skip_type != "sum":
            unsupported.append(f"skip_type={skip_type!r}")
        if norm is not None:
            unsupported.append(f"norm={norm!r}")
        if not use_upsample_conv:
            unsupported.append("transposed-convolution decoders")
        if recurrent_block_type != "convgru":
            unsupported.append(f"recurrent_block_type={recurrent_block_type!r}")
        if encoder_stride != 2 or kernel_size != 3:
            unsupported.append("encoder_stride / kernel_size other than 2 / 3")
        if tuple(activations) != ("relu", None):
            unsupported.append(f"activations={activations!r}")
        if unsupported:
            raise NotImplementedError("RecEVFlowNet on the MI355X path is the reference's default architecture; not built: "
                                      + ", ".join(unsupported))
        self.num_bins, self.levels, self.nres, self.nout = num_bins, num_encoders, num_residual_blocks, num_output_channels
        self.stride, self.ksize = encoder_stride, kernel_size
        self.final_activation, self.final_bias, self.final_w_scale = final_activation, final_bias, final_w_scale
        self.width = [int(base_channels * channel_multiplier ** i) for i in range(num_encoders)]     # 64, 128, 256, 512
        self.multiple = encoder_stride ** num_encoders if min_size is None else min_size
        rows, cin = [], num_bins
        for i, c in enumerate(self.width):
            rows.append(Row("head", i, cin, c, kernel_size, encoder_stride, "relu"))
            rows.append(Row("gru", i, 2 * c, c, 3, 1, None))
            cin = c
        top = self.width[-1]
        for _ in range(num_residual_blocks):
            rows.append(Row("res", num_encoders - 1, top, top, 3, 1, "relu"))
        src = top
        for k in range(num_encoders):                  # decoder k works on level levels-1-k and emits half its width
            lvl = num_encoders - 1 - k
            out = self.width[lvl] // channel_multiplier if lvl > 0 else base_channels // channel_multiplier
            rows.append(Row("dec", lvl, src + (num_output_channels if k else 0), out, kernel_size, 1, "relu"))
            rows.append(Row("pred", lvl, out, num_output_channels, 1, 1, final_activation))
            src = out
        self.rows = rows

    def padding(self, H, W):
        """Rows / columns to add at the top / left so that every level halves exactly (model_util.py:52-60)."""
        m = self.multiple
        return (m - H % m) % m, (m - W % m) % m

    def of(self, kind):
        return [r for r in self.rows if r.kind == kind]


class MultiResUNetRecurrent(nn.Module):
    """Parameters + recurrent state of the network; the arithmetic lives in `engine.PassEngine`."""

    def __init__(self, kwargs):
        super().__init__()
        self.plan = plan = NetPlan(**kwargs)
        self.num_encoders = self.num_states = plan.levels
        self.encoders = nn.ModuleList(
            RecurrentConvLayer(r.cin, r.cout, kernel_size=r.ksize, stride=r.stride, recurrent_block_type="convgru",
                               activation_ff=r.act, activation_rec=None) for r in plan.of("head"))
        self.resblocks = nn.ModuleList(ResidualBlock(r.cin, r.cout, activation=r.act) for r in plan.of("res"))
        self.decoders = nn.ModuleList(UpsampleConvLayer(r.cin, r.cout, kernel_size=r.ksize, activation=r.act)
                                      for r in plan.of("dec"))
        self.preds = nn.ModuleList(ConvLayer(r.cin, r.cout, 1, activation=r.act, w_scale=plan.final_w_scale,
                                             bias=plan.final_bias) for r in plan.of("pred"))
        self.states = [None] * self.num_states
        self._engine = None
        # The literal drop-in loop (reference train_flow.py:60-70, :120-131: torch.optim.Adam, clip_grad_norm_,
        # optimizer.zero_grad() with torch's set_to_none default) never sees train.Trainer.  The network then owns its
        # gradient accumulators itself: ONE flat buffer with a view per parameter as .grad (re-attached and cleared when the
        # optimiser has set them to None), which is what the fused pass needs to add parameter gradients in place, to keep
        # the two halves of a pass apart and to defer the weight gradients of a window to one reduction per layer (flushed
        # by a callback at the end of backward()).  train.Trainer switches this off and brings its own bucket.
        self.auto_grads = True
        self._bucket = None
        # extra factor on the full-resolution flows of the fused pass: a training loop that multiplies the network's
        # output by loss.flow_scaling (reference train_flow.py:107-108) can have the pass do it in its last kernel
        self.flow_scale = 1.0
        self._window = []        # (record, new states) of the encoder halves queued by encode()

    # -- switches train.Trainer flips on every PackedWeights of the tree (submodules.enable_*): read them where they live
    @property
    def direct_grads(self):
        return self.encoders[0].conv._packed.direct_grads

    @property
    def deferred_wgrad(self):
        return self.encoders[0].conv._packed.defer_wgrad

    @property
    def engine(self):
        if self._engine is None:
            self._engine = PassEngine(self)
        return self._engine

    def own_gradients(self):
        """-> True when every trainable parameter's .grad is a view of this module's flat buffer (see __init__)."""
        if not self.auto_grads:
            return False
        params = [p for p in self.parameters() if p.requires_grad]
        if not params or not params[0].is_cuda:
            return False
        b = self._bucket
        if b is None and all(p.grad is not None and p.grad.is_contiguous() for p in params):
            return False                        # the caller keeps gradient buffers of its own (and flushes what it defers)
        if b is None or len(b.params) != len(params) or any(x is not y for x, y in zip(b.params, params)) or \
                b.flat.device != params[0].device:
            try:
                from ..parallel import FlatGradBucket
                from . import submodules as sm
            except ImportError:                 # drop-in mode: this directory is the top-level package `models`
                from parallel import FlatGradBucket
                import models.submodules as sm
            old = [p.grad for p in params]
            self._bucket = b = FlatGradBucket(params)
            for p, g in zip(params, old):       # (gradients accumulated before the first pass through here)
                if g is not None:
                    p.grad.copy_(g)
            sm.enable_direct_grads(self)
            sm.enable_deferred_wgrad(self, os.environ.get("TEF_NO_DEFERRED_WGRAD", "0") != "1")
            for m_ in self.modules():           # (layer-by-layer passes through this network flush at the end of backward too)
                for v_ in vars(m_).values():
                    if isinstance(v_, sm.PackedWeights):
                        v_.auto_flush = True
            # the deferred weight gradients of every few finished backward passes are reduced on a stream of their own
            # beside the rest of BPTT (as under train.Trainer); the callback at the end of backward() joins it
            group = int(os.environ.get("TEF_WGRAD_GROUP", "3"))
            eng = self.engine
            if self.deferred_wgrad and group > 0 and os.environ.get("TEF_TWO_STREAMS", "1") != "0" and eng.wgrad_stream is None:
                import torch

                eng.wgrad_stream, eng.wgrad_group = torch.cuda.Stream(device=params[0].device), group
            # ... and the decoder half of a pass runs on a side stream beside the next pass's encoders, its flows handed out
            # as LazyFlow tensors (models/lazy.py: scaling and the loss container's update() stay on that stream)
            if os.environ.get("TEF_TWO_STREAMS", "1") != "0" and os.environ.get("TEF_LAZY_FLOWS", "1") != "0" and eng.side_stream is None:
                import torch

                eng.side_stream = torch.cuda.Stream(device=params[0].device)
                eng.lazy_flows = True
            return True
        # optimizer.zero_grad(set_to_none=True) — every gradient None at once — clears the whole buffer in one fill; when only
        # SOME parameters lost their view (a param group zeroed, p.grad = None for a subset, gradient accumulation over
        # several backward() calls with a partial zero_grad) only THEIR slices are cleared: the gradients already accumulated
        # in the views that are still attached stay
        if all(p.grad is None for p in params):
            b.flat.zero_()
            o = 0
            for p in params:
                p.grad = b.flat[o:o + p.numel()].view_as(p)
                o += p.numel()
            return True
        o = 0
        for p in params:
            g = p.grad
            if g is None or g.data_ptr() != b.flat.data_ptr() + 4 * o or not g.is_contiguous():
                view = b.flat[o:o + p.numel()].view_as(p)
                if g is None:
                    view.zero_()
                else:
                    view.copy_(g)
                p.grad = view
            o += p.numel()
        return True

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None                 # workspaces are not part of a checkpoint
        state["_bucket"] = None
        state["states"] = [None] * self.num_states
        state["_window"] = []
        return state

    # ---- window mode (train.Trainer): encoder halves pass by pass, the decoder halves of the whole window as one batch ----
    def encode(self, x):
        """The encoder half of one pass: updates `states` and queues the pass for decode_window().  Needs the training
        loop's in-place gradient buffers (direct_grads) like every split pass."""
        # (first: no pass of this window has gone onto the level streams yet — a fresh sequence's first pass does not)
        new_states, rec = encode_pass(self.engine, x, self.states, first=not any(r.hub is not None for r, _ in self._window))
        self.states = new_states
        self._window.append((rec, new_states))

    def decode_window(self):
        """-> flows[t][k] of every pass queued by encode() since the last call: residual blocks, decoders and heads of all of
        them as ONE batch (only the recurrent states cross passes: reference models/arch.py:225-227)."""
        recs, states = [r for r, _ in self._window], [s for _, s in self._window]
        self._window = []
        return decode_passes(self.engine, recs, states)

    def drop_window(self):
        self._window = []

    def step(self, x):
        """One fused pass on an input of any size: -> 4 flows at the input resolution, coarse to fine; updates `states`."""
        flows, self.states = run_pass(self.engine, x, self.states)
        return flows

    def forward(self, x):
        """Multi-resolution predictions [B, 2, h_k, w_k] (coarse to fine) for an input whose sides are multiples of
        2**num_encoders, as the reference's forward (:217-242) returns them: layer by layer through the modules' own
        autograd nodes.  RecEVFlowNet does not come through here (it calls `step`); this path exists for users of the
        bare architecture and as the independent check of the fused pass (tests/test_model_gpu.py)."""
        feats, cur = [], x
        for i, enc in enumerate(self.encoders):
            cur, self.states[i] = enc(cur, self.states[i])
            feats.append(cur)
        for rb in self.resblocks:
            cur, _ = rb(cur)
        preds = []
        for k, (dec, head) in enumerate(zip(self.decoders, self.preds)):
            fused = dec.conv_after_upsample(upsample_bilinear(cur + feats[-1 - k], 2, 2),
                                            upsample_bilinear(preds[-1], 2, 2) if preds else None)
            cur = fused
            preds.append(head(cur))
        return preds
