"""One recurrent pass of RecEVFlowNet as ONE autograd node and ONE call into libtef_hip.so per direction.

The reference runs the pass as ~60 autograd nodes (models/arch.py:217-242, models/model.py:65-85) and lets autograd sum
the gradients of every tensor that has several consumers.  Round 2 turned the pass into one autograd node whose Python
body issued ~100 ctypes calls and ~60 tensor allocations; an eager caller — a drop-in ``train_flow.py`` loop — was then
bound by the host (49 ms of Python per window for 38 ms of GPU work).  Now the layer table of `arch.NetPlan` is walked in C
(csrc/tef_net.hip, include/tef.h ``tef_net_*``):

  forward   `tef_net_pass_forward`: [pad] -> 4 x (strided head conv, fused ConvGRU cell) -> residual blocks -> 4 x
            (bilinear x2 of (features + encoder skip) [+ of the previous prediction], conv over the two sources without
            materialising their concatenation, 1x1 tanh head, bilinear to the input size x 2^(3-k) x flow_scale, crop);
            all activations the backward needs and the pass's outputs live in ONE arena (the "tape"), the flows and the new
            states are views of it
  backward  `tef_net_pass_backward`: the same table in reverse, a gradient with several producers summed where it is
            consumed; parameter gradients go straight into the caller's accumulators (.grad of the flat bucket, or one
            fresh zero buffer handed back to autograd) or, inside a BPTT window, stay as pre-activation gradients in the
            call's gradient arena until `flush_window` reduces them over all passes with ONE `tef_net_window_wgrads`.

This file only fills the plan (pointers of packed weights, biases, gradient targets), owns the arenas and is the autograd
boundary.  No ATen arithmetic runs inside a pass (padding a non-multiple-of-16 input is one strided copy).

Two streams (opt-in, `PassEngine.side_stream`; train.Trainer switches it on): only the recurrent states cross passes
(reference models/arch.py:225-227), so the residual blocks / decoders / heads of pass t are independent of the encoders
of pass t + 1.  The pass is then TWO autograd nodes — `_EncFn` on the caller's stream, `_DecFn` on the side stream
(`tef_net_pass_forward_part` / `_backward_part`) — and autograd itself runs each node's backward on the stream of its
forward, so the halves of consecutive passes overlap in both directions and fill each other's launch ramps and tails.
The flows then live on the side stream: whoever consumes them either works on that stream or waits for it
(`PassEngine.join`); without `defer_join` the pass waits itself before returning.
"""

import ctypes
import os
import weakref

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib

from . import submodules as sm
from .lazy import LazyFlow

ACT = _lib.ACT


def _p(t):
    return t.data_ptr() if t is not None else None


class _Tape:
    """What one forward pass leaves for its backward: the plan it ran with, its inputs and its activation arena."""

    __slots__ = ("plan", "xp", "states_in", "states_arr", "tape", "geom", "x_shape", "gtape", "ran", "targets_set", "queued",
                 "stream", "hn", "new_states", "dec_dstates", "dec_stream", "high_stream", "above_valid", "hub", "dec_token")


class PassEngine:
    def __init__(self, arch):
        # weak: the architecture owns its engine; a strong back-reference would make every model a reference cycle that only
        # the cycle collector — at whatever allocation it wakes up on, with streams and hipGraphs possibly in flight — frees
        self._arch = weakref.ref(arch)
        self.plan = arch.plan
        self._ws = {}                 # one workspace per stream the engine launches on
        self.side_stream = None       # set: passes that record a graph run as two nodes on two streams (module docstring)
        self.defer_join = False       # with side_stream: leave the flows on the side stream (the caller joins)
        self.lazy_flows = False       # with side_stream, without defer_join: hand the flows out as LazyFlow tensors (models/lazy.py)
        # window mode: (low stream, high stream) — the encoder levels [0, n/2) of pass t + 1 run beside the levels [n/2, n) of
        # pass t (encode_pass); None: the encoder half of a pass is one call on the caller's stream
        self.enc_streams = None
        self._dec_seen = {}              # stream -> the window (token) whose batched decoders' backward it has waited for
        self.debug_delay_levels = None   # tests: cycles of spinning in front of the (lower, upper) level ranges, forward and backward
        self.debug_delay = None       # tests: cycles of spinning put in front of the (encoder halves, decoder halves[, weight-gradient groups])
        # two-stream windows: the deferred weight gradients of every `wgrad_group` finished backward passes are reduced on
        # `wgrad_stream` while BPTT goes on (the encoder chain leaves the side stream idle about half of the time and
        # a long reduction fills it better than its turn at the end, alone); None / 0: one reduction per layer over the
        # whole window at the flush
        self.wgrad_stream, self.wgrad_group = None, 0
        self._zeros = {}
        self._layout = {}
        self._pending = []            # backward calls of the current window whose weight gradients are still deferred
        self._callback_queued = False # an end-of-backward flush is queued with autograd (cleared by the callback itself)
        self._own_packers = None      # ids of the PackedWeights of this engine's network (end-of-backward flush)
        self.last_pass_split = False  # whether the last run_pass left its flows on the side stream

    @property
    def arch(self):
        a = self._arch()
        if a is None:
            raise RuntimeError("RecEVFlowNet pass engine used after its model was released")
        return a

    def close(self):
        """Wait for the engine's streams and drop its device buffers (workspaces, cached zero states, unreduced weight
        gradients).  Idempotent; the engine can be used again afterwards."""
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            for st in (self.side_stream, self.wgrad_stream) + tuple(self.enc_streams or ()):
                if st is not None:
                    st.synchronize()
            torch.cuda.current_stream().synchronize()
        for r in self._pending:
            r.queued = False
        self._pending = []
        sm._DEFERRED_ENGINES.discard(self)
        self._ws.clear()
        self._zeros.clear()

    # ---- buffers -----------------------------------------------------------------------------------------------
    def workspace(self, nbytes, device):
        key = torch.cuda.current_stream().cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = self._ws[key] = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)
        return ws

    def join(self):
        """Make the current stream wait for the side stream (the decoder halves issued so far)."""
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
        self.join_encoders()

    def dec_wait_needed(self, stream, rec):
        """Whether `stream` still has to wait for the batched decoders' backward of rec's window: the first encoder node of
        the window on that stream waits, the later ones are ordered behind it."""
        key = stream.cuda_stream
        if self._dec_seen.get(key) is rec.dec_token:
            return False
        self._dec_seen[key] = rec.dec_token
        return True

    def join_encoders(self):
        """Make the current stream wait for the pipelined encoder halves issued so far (enc_streams)."""
        if self.enc_streams is not None:
            cur = torch.cuda.current_stream()
            if cur in self.enc_streams:
                # called from a level stream (a group of weight gradients flushed from the lower range's backward): what the
                # other one did for these passes is already ordered before this point (_EncLowFn.backward), and a direct
                # wait between the two in this direction is what hipStreamEndCapture does not survive
                return
            for s_ in self.enc_streams:
                cur.wait_stream(s_)

    def zero_state(self, shape, device):
        key = (tuple(shape), device)
        z = self._zeros.get(key)
        if z is None:
            z = self._zeros[key] = torch.zeros(shape, dtype=torch.float32, device=device)
        return z

    # ---- the plan ----------------------------------------------------------------------------------------------
    def _conv_fields(self, nc, packer, weights, biases, desc, batch=None):
        wp, wt = packer.get(weights, desc, batch)      # (stale operands: queued in `batch`, packed by make_plan in one launch)
        bias = packer.bias(biases)
        nc.wp, nc.w2, nc.bias = wp.data_ptr(), wt.data_ptr(), _p(bias)

    def make_plan(self, B, Hp, Wp, ph, pw, part=3, pl=None, dec_only=False):
        """struct tef_net_plan for a (padded) input of B x bins x Hp x Wp: geometry + the packed weights / biases of
        every convolution (re-packed by PackedWeights only when a parameter changed).  part 1: geometry and the encoders'
        layers only; part 2 completes the plan `pl` with the residual blocks, decoders and heads — on the stream of the
        caller, so that after an optimiser step the decoder half's weights are re-packed beside the first encoders
        instead of in front of them."""
        a, np_ = self.arch, self.plan
        dec_rows = np_.of("dec")
        if (part & 1) or pl is None:
            pl = _lib.NetPlan()
            pl.dec_only = 1 if dec_only else 0
            pl.copy_batch = 1        # (flush_window hands tef_net_window_wgrads its workspace: _launch_wgrads)
            pl.B, pl.H, pl.W = B, Hp, Wp
            pl.bins, pl.levels, pl.nres, pl.nout = np_.num_bins, np_.levels, np_.nres, np_.nout
            pl.final_act = ACT[np_.final_activation]
            pl.crop_top, pl.crop_left, pl.flow_scale = ph, pw, float(a.flow_scale)
            for i, c in enumerate(np_.width):
                pl.width[i] = c
                pl.dec_out[i] = dec_rows[i].cout
        cin, h, w = np_.num_bins, Hp, Wp
        batch = []      # weight parts to (re-)pack: all of a half of the network in ONE launch (after an optimiser step)
        for i, enc in enumerate(a.encoders):
            c = np_.width[i]
            if part & 1:
                self._conv_fields(pl.head[i], enc.conv._packed, (enc.conv.conv2d.weight,), (enc.conv.conv2d.bias,),
                                  _lib.ConvDesc(B, cin, 0, h, w, c, 3, np_.stride, ACT["relu"]), batch)
            h, w = h // 2, w // 2
            g = enc.recurrent_block
            if part & 1:
                self._conv_fields(pl.gate_ur[i], g._packed_ur, (g.update_gate.weight, g.reset_gate.weight),
                                  (g.update_gate.bias, g.reset_gate.bias), _lib.ConvDesc(B, c, c, h, w, 2 * c, 3, 1, ACT["sigmoid"]), batch)
                self._conv_fields(pl.gate_o[i], g._packed_o, (g.out_gate.weight,), (g.out_gate.bias,),
                                  _lib.ConvDesc(B, c, c, h, w, c, 3, 1, ACT["tanh"]), batch)
            cin = c
        sm.run_pack_jobs(batch)
        if not (part & 2):
            return pl
        top = np_.width[-1]
        for j, rb in enumerate(a.resblocks):
            d = _lib.ConvDesc(B, top, 0, h, w, top, 3, 1, ACT["relu"])
            self._conv_fields(pl.res1[j], rb._packed1, (rb.conv1.weight,), (rb.conv1.bias,), d, batch)
            self._conv_fields(pl.res2[j], rb._packed2, (rb.conv2.weight,), (rb.conv2.bias,), d, batch)
        src = top
        for k, (dec, head) in enumerate(zip(a.decoders, a.preds)):
            h, w = h * 2, w * 2
            c0, c1 = (np_.nout, src) if k else (src, 0)
            out = dec_rows[k].cout
            self._conv_fields(pl.dec[k], dec._packed, (dec.conv2d.weight,), (dec.conv2d.bias,),
                              _lib.ConvDesc(B, c0, c1, h, w, out, 3, 1, ACT["relu"]), batch)
            self._conv_fields(pl.pred[k], head._packed, (head.conv2d.weight,), (head.conv2d.bias,),
                              _lib.ConvDesc(B, out, 0, h, w, np_.nout, 1, 1, ACT[np_.final_activation]), batch)
            src = out
        sm.run_pack_jobs(batch)
        return pl

    def layout(self, pl):
        """(tape floats, gradient-arena floats, workspace bytes, flow / state / dstate offsets, dx offset) of a geometry."""
        key = (pl.B, pl.H, pl.W, pl.crop_top, pl.crop_left, pl.dec_only)
        lay = self._layout.get(key)
        if lay is None:
            lib = _lib.lib()
            n = self.plan.levels
            fo, so, do = ((ctypes.c_size_t * n)() for _ in range(3))
            dx = ctypes.c_size_t()
            _lib.check(lib.tef_net_layout(ctypes.byref(pl), fo, so, do, ctypes.byref(dx)), "tef_net_layout")
            tape, gtape = lib.tef_net_tape_floats(ctypes.byref(pl)), lib.tef_net_gtape_floats(ctypes.byref(pl))
            if not tape or not gtape:
                _lib.check(-1, "tef_net_tape_floats")
            lay = self._layout[key] = (tape, gtape, lib.tef_net_workspace_bytes(ctypes.byref(pl)), list(fo), list(so),
                                       list(do), dx.value)
        return lay

    def _grad_targets(self, pl, params_grad, defer):
        """Fill the gradient accumulators of the plan: params_grad(parameter) -> tensor to add into (or None)."""
        a = self.arch

        def tgt(nc, weights, biases):
            gw = [params_grad(w) for w in weights]
            gb = [params_grad(b) if b is not None else None for b in biases]
            nc.dw, nc.dw2 = _p(gw[0]), (_p(gw[1]) if len(gw) > 1 else None)
            nc.db, nc.db2 = _p(gb[0]), (_p(gb[1]) if len(gb) > 1 else None)
            nc.defer = 1 if (defer and all(g_ is not None for g_ in gw)) else 0

        for i, enc in enumerate(a.encoders):
            g = enc.recurrent_block
            tgt(pl.head[i], (enc.conv.conv2d.weight,), (enc.conv.conv2d.bias,))
            tgt(pl.gate_ur[i], (g.update_gate.weight, g.reset_gate.weight), (g.update_gate.bias, g.reset_gate.bias))
            tgt(pl.gate_o[i], (g.out_gate.weight,), (g.out_gate.bias,))
        for j, rb in enumerate(a.resblocks):
            tgt(pl.res1[j], (rb.conv1.weight,), (rb.conv1.bias,))
            tgt(pl.res2[j], (rb.conv2.weight,), (rb.conv2.bias,))
        for k, (dec, head) in enumerate(zip(a.decoders, a.preds)):
            tgt(pl.dec[k], (dec.conv2d.weight,), (dec.conv2d.bias,))
            tgt(pl.pred[k], (head.conv2d.weight,), (head.conv2d.bias,))

    # ---- the pass ----------------------------------------------------------------------------------------------
    def forward(self, x, states, keep, part=3, rec=None, levels=None):
        """x [B, bins, H, W] -> (flows: 4 x [B, 2, H, W], new states: 4 x [B, C_i, h_i, w_i], tape | None).
        part 1 (encoders only): flows is None, the record is always returned; part 2 (the rest of the pass `rec` began,
        on the current stream): new states is None.
        levels (lo, hi), part 1 only: the encoder levels [lo, hi) alone, on the current stream (tef_net_pass_forward_levels:
        the levels of consecutive passes pipelined over streams) — lo = 0 begins the pass's record (`states`: ALL incoming
        states), lo > 0 continues `rec`; new states: those of the range (None elsewhere)."""
        plan = self.plan
        n = plan.levels
        if levels is not None and levels[0] > 0:
            cur = torch.cuda.current_stream()
            if cur != rec.stream and not torch.cuda.is_current_stream_capturing():
                rec.tape.record_stream(cur)
            pl = rec.plan
            _, _, wsb, _, so, _, _ = self.layout(pl)
            ws = self.workspace(wsb, rec.tape.device)
            if self.debug_delay_levels and self.debug_delay_levels[1]:
                torch.cuda._sleep(int(self.debug_delay_levels[1]))
            rc = _lib.lib().tef_net_pass_forward_levels(ctypes.byref(pl), levels[0], levels[1], _p(rec.xp), rec.states_arr,
                                                        rec.tape.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, "tef_net_pass_forward_levels")
            return None, [rec.tape[so[i]:so[i] + rec.states_in[i].numel()].view(rec.states_in[i].shape)
                          if levels[0] <= i < levels[1] else None for i in range(n)], rec
        if part & 1:
            _lib.require_device_tensor(x, "network input")
            x = x.contiguous()
            if x.dtype != torch.float32:
                x = x.to(torch.float32)
            B, _, H, W = x.shape
            ph, pw = plan.padding(H, W)
            if ph or pw:        # E-RAFT style padding at the top / left (reference models/model_util.py:52-65)
                xp = x.new_zeros((B, x.shape[1], H + ph, W + pw))
                xp[:, :, ph:, pw:] = x
            else:
                xp = x
            pl = self.make_plan(B, H + ph, W + pw, ph, pw, part)
            ntape = self.layout(pl)[0]
            st, h, w = [], H + ph, W + pw
            for i in range(n):
                h, w = h // 2, w // 2
                s = states[i]
                if s is None:
                    s = self.zero_state((B, plan.width[i], h, w), x.device)
                else:
                    _lib.require_device_tensor(s, "recurrent state")
                    if s.dtype != torch.float32 or not s.is_contiguous():
                        s = s.to(torch.float32).contiguous()
                    if tuple(s.shape) != (B, plan.width[i], h, w):
                        raise RuntimeError(f"recurrent state {i} has shape {tuple(s.shape)}, expected {(B, plan.width[i], h, w)}")
                st.append(s)
            rec = _Tape()
            rec.plan, rec.xp, rec.states_in, rec.tape = pl, xp, st, torch.empty((ntape,), dtype=torch.float32, device=x.device)
            rec.states_arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in st])
            rec.geom, rec.x_shape = (ph, pw), tuple(x.shape)
            rec.gtape, rec.ran, rec.targets_set, rec.queued, rec.stream = None, 0, False, False, torch.cuda.current_stream()
            rec.hn = rec.new_states = rec.dec_dstates = rec.dec_stream = rec.high_stream = rec.hub = rec.dec_token = None
            rec.above_valid = 0
        pl = rec.plan
        if part == 2:
            self.make_plan(pl.B, pl.H, pl.W, pl.crop_top, pl.crop_left, 2, pl)
        _, _, wsb, fo, so, _, _ = self.layout(pl)
        B, H, W = rec.x_shape[0], rec.x_shape[2], rec.x_shape[3]
        ws = self.workspace(wsb, rec.tape.device)
        if levels is not None:
            if self.debug_delay_levels and self.debug_delay_levels[0]:
                torch.cuda._sleep(int(self.debug_delay_levels[0]))
            rc = _lib.lib().tef_net_pass_forward_levels(ctypes.byref(pl), 0, levels[1], rec.xp.data_ptr(), rec.states_arr,
                                                        rec.tape.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        elif part == 3:
            rc = _lib.lib().tef_net_pass_forward(ctypes.byref(pl), rec.xp.data_ptr(), rec.states_arr, rec.tape.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        else:
            rc = _lib.lib().tef_net_pass_forward_part(ctypes.byref(pl), part, rec.xp.data_ptr(), rec.states_arr, rec.tape.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_net_pass_forward")
        flows = new_states = None
        if part & 2:
            fshape = (B, plan.nout, H, W)
            flows = [rec.tape[fo[k]:fo[k] + B * plan.nout * H * W].view(fshape) for k in range(n)]
        if part & 1:
            new_states = [rec.tape[so[i]:so[i] + rec.states_in[i].numel()].view(rec.states_in[i].shape)
                          if (levels is None or i < levels[1]) else None for i in range(n)]
        return flows, new_states, (rec if (keep or part != 3) else None)

    # ---- the decoder half of a whole loss window as ONE batch (round 6) --------------------------------------------------
    def decode_window(self, recs, states):
        """recs: the records of P encoder halves (forward(part=1)) of one geometry; states[t][i]: the new state of level i
        of pass t.  -> (flows[t][k] as views of one arena, the window's record).  Only the recurrent states cross passes
        (reference models/arch.py:225-227): the residual blocks, decoders and heads of the P passes are independent, and as
        one batch of P x B samples they are ten times fewer launches on GEMMs ten times larger (the 8 x 8 level: 512 pixels,
        split 16..32 ways over k at 50 TFLOP/s, become 5120 at 110)."""
        plan, n, P = self.plan, self.plan.levels, len(recs)
        p0 = recs[0].plan
        B = p0.B
        dev = recs[0].tape.device
        if self.enc_streams is not None:       # the states were written on the level streams
            self.join_encoders()
            cur = torch.cuda.current_stream()
            for r in recs:
                if r.stream != cur and not torch.cuda.is_current_stream_capturing():
                    r.tape.record_stream(cur)
        # the passes' states stacked along the batch: [P * B, C_i, h_i, w_i] (4 copies per window, 16 MB per pass)
        hn = [torch.cat([states[t][i].detach() for t in range(P)], 0) for i in range(n)]
        pl = self.make_plan(P * B, p0.H, p0.W, p0.crop_top, p0.crop_left, 2, None, dec_only=True)
        for i in range(n):
            pl.hn_ext[i] = hn[i].data_ptr()
        ntape, _, wsb, fo, _, _, _ = self.layout(pl)
        rec = _Tape()
        rec.plan, rec.xp, rec.states_in, rec.states_arr = pl, None, None, None
        rec.tape = torch.empty((ntape,), dtype=torch.float32, device=dev)
        rec.geom, rec.x_shape = recs[0].geom, (P * B,) + tuple(recs[0].x_shape[1:])
        rec.gtape, rec.ran, rec.targets_set, rec.queued, rec.stream = None, 0, False, False, torch.cuda.current_stream()
        rec.hn, rec.new_states, rec.dec_dstates, rec.dec_stream, rec.high_stream, rec.above_valid, rec.hub, rec.dec_token = hn, None, None, None, None, 0, None, None
        ws = self.workspace(wsb, dev)
        rc = _lib.lib().tef_net_pass_forward_part(ctypes.byref(pl), 2, None, None, rec.tape.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _lib.stream_ptr())
        _lib.check(rc, "tef_net_pass_forward (window decoders)")
        H, W = recs[0].x_shape[2], recs[0].x_shape[3]
        per = B * plan.nout * H * W
        flows = [[rec.tape[fo[k] + t * per:fo[k] + (t + 1) * per].view(B, plan.nout, H, W) for k in range(n)] for t in range(P)]
        return flows, rec

    def backward(self, rec, dflows, dstates, params, want_dx, part=3, dstates2=None, levels=None):
        """-> (gradients w.r.t. the incoming states, gradient w.r.t. the network input or None, parameter gradients for
        autograd).  Parameter gradients are added into the parameters' own .grad buffers when the training loop owns them
        (`direct_grads`: nothing is handed to autograd), otherwise into one fresh zero buffer whose views are returned.
        part 2 (decoder half, two-stream passes; direct gradients only): the first result is what the half sends to each
        NEW state (one tensor or None per level); part 1 (encoder half): `dstates` are the total gradients of the new
        states."""
        if rec is None:
            raise RuntimeError("RecEVFlowNet pass: backward a second time through the same pass — its activation arena is "
                               "released after the first backward (retain_graph=True is not supported by the fused pass; "
                               "model.arch(x) runs the network through ordinary autograd nodes)")
        a, plan = self.arch, self.plan
        n = plan.levels
        pl = rec.plan
        dev = rec.tape.device
        direct = a.direct_grads and all(p.grad is not None and p.grad.is_contiguous() for p in params if p.requires_grad)
        if part != 3 and not direct:
            raise RuntimeError("two-stream passes accumulate parameter gradients in place: they need direct_grads")
        fresh = None
        if direct:
            if not rec.targets_set:
                self._grad_targets(pl, lambda p: p.grad if p.requires_grad else None, a.deferred_wgrad)
                rec.targets_set = True
        else:
            tot = sum(p.numel() for p in params if p.requires_grad)
            flat = torch.zeros((tot,), dtype=torch.float32, device=dev)
            fresh, o = {}, 0
            for p in params:
                if p.requires_grad:
                    fresh[id(p)] = flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
            self._grad_targets(pl, lambda p: fresh.get(id(p)), False)
        _, ngt, wsb, _, _, do, dxo = self.layout(pl)
        if self.debug_delay and part != 3 and self.debug_delay[part - 1]:
            torch.cuda._sleep(int(self.debug_delay[part - 1]))
        if rec.gtape is None:
            rec.gtape = torch.empty((ngt,), dtype=torch.float32, device=dev)
        gtape = rec.gtape
        if part != 3 and rec.stream is not None:      # the halves of a pass share the arenas across their two streams
            cur = torch.cuda.current_stream()
            if cur != rec.stream:
                gtape.record_stream(rec.stream)
                gtape.record_stream(cur)
        dfl = [None if g is None else g.to(torch.float32).contiguous() for g in (dflows or [None] * n)]
        dst = [None if g is None else g.to(torch.float32).contiguous() for g in (dstates or [None] * n)]
        a_dfl = (ctypes.c_void_p * n)(*[_p(g) for g in dfl])
        a_dst = (ctypes.c_void_p * n)(*[_p(g) for g in dst])
        ran = ctypes.c_ulonglong(0)
        off = (ctypes.c_longlong * n)()
        dxv = ctypes.c_int(0)
        ws = self.workspace(wsb, dev)
        if levels is not None:
            # the encoder levels [lo, hi) alone (part 1), on the current stream; the range above left rec.above_valid
            lo, hi = levels
            k_ = 0 if lo == 0 else 1
            if self.debug_delay_levels and self.debug_delay_levels[k_]:
                torch.cuda._sleep(int(self.debug_delay_levels[k_]))
            dst2 = [None if g is None else g.to(torch.float32).contiguous() for g in (dstates2 or [None] * n)]
            a_dst2 = (ctypes.c_void_p * n)(*[_p(g) for g in dst2])
            rc = _lib.lib().tef_net_pass_backward_levels(ctypes.byref(pl), lo, hi, int(rec.above_valid) if hi < n else 0, _p(rec.xp),
                                                         rec.states_arr, rec.tape.data_ptr(), a_dst, a_dst2, 1 if want_dx else 0,
                                                         gtape.data_ptr(), ctypes.byref(ran), off, ctypes.byref(dxv), ws.data_ptr(),
                                                         ws.numel(), _lib.stream_ptr())
            if lo > 0:
                rec.above_valid = int(dxv.value)
        elif dstates2 is not None and any(g is not None for g in dstates2):
            # a second addend per new state (window mode: the batched decoders' share, handed over beside autograd)
            dst2 = [None if g is None else g.to(torch.float32).contiguous() for g in dstates2]
            a_dst2 = (ctypes.c_void_p * n)(*[_p(g) for g in dst2])
            rc = _lib.lib().tef_net_pass_backward_part2(ctypes.byref(pl), part, _p(rec.xp), rec.states_arr, rec.tape.data_ptr(),
                                                        a_dfl, a_dst, a_dst2, 1 if want_dx else 0, gtape.data_ptr(),
                                                        ctypes.byref(ran), off, ctypes.byref(dxv), ws.data_ptr(), ws.numel(),
                                                        _lib.stream_ptr())
        else:
            rc = _lib.lib().tef_net_pass_backward_part(ctypes.byref(pl), part, _p(rec.xp), rec.states_arr, rec.tape.data_ptr(),
                                                       a_dfl, a_dst, 1 if want_dx else 0, gtape.data_ptr(), ctypes.byref(ran), off,
                                                       ctypes.byref(dxv), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_net_pass_backward")
        rec.ran |= ran.value
        shapes = [t_.shape for t_ in (rec.hn if rec.hn is not None else rec.states_in)]      # (a window's decoders: the stacked states)
        dh = [gtape[off[i]:off[i] + shapes[i].numel()].view(shapes[i]) if off[i] >= 0 else None for i in range(n)]
        dx = None
        if want_dx and (part & 1) and (levels is None or levels[0] == 0):
            ph, pw = rec.geom
            if dxv.value:
                dx = gtape[dxo:dxo + rec.xp.numel()].view(rec.xp.shape)[:, :, ph:, pw:]
            else:       # no gradient reached the first encoder: zeros, like autograd's own
                dx = torch.zeros(rec.x_shape, dtype=torch.float32, device=dev)
        if direct and a.deferred_wgrad and not rec.queued:
            same = [r_ for r_ in self._pending if r_.plan.dec_only == pl.dec_only]      # (a window's batched decoders sit beside its passes)
            if same and (same[0].plan.B, same[0].plan.H, same[0].plan.W) != (pl.B, pl.H, pl.W):
                self.flush_window()
            if a.auto_grads and a._bucket is not None and not self._callback_queued:
                # nobody calls flush_window for the drop-in loop: at the end of this backward() (autograd runs the queued
                # callbacks when every node is done).  The flag is the callback's own — not "the queue was empty": a
                # backward() that raised after queueing records leaves them behind, and every later backward still gets its
                # flush (the callback of a failed graph task never runs: the next forward drops the flag, run_pass)
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self._callback_queued = True
            self._pending.append(rec)       # (keeps the arenas alive until the flush)
            rec.queued = True
            sm._DEFERRED_ENGINES.add(self)
        if rec.queued and rec.hn is not None and self.wgrad_stream is not None and self.wgrad_group:
            # the batched decoder half of a window: its weight gradients (a third of the window's) run beside the encoders' BPTT.
            # ONLY this record: this backward runs on the decoders' stream, and the reduction stream waits for the stream the
            # flush is issued from — encoder records still queued belong to the caller's stream
            self.flush_window(self.wgrad_stream, only=[rec])
        if rec.queued and part == 1 and (levels is None or levels[0] == 0) and self.wgrad_stream is not None and self.wgrad_group:
            # (backward walks the passes last to first and a pass's encoder half is its last piece: when this record is
            # the newest one queued, every queued pass is complete)
            done = self._pending.index(rec) + 1
            if done >= self.wgrad_group and done == len(self._pending):
                self.flush_window(self.wgrad_stream)
        grads = [None] * len(params) if direct else [fresh.get(id(p)) for p in params]
        return dh, dx, grads

    def _end_of_backward(self):
        """arch.auto_grads: what train.Trainer does after loss.backward() — the window's remaining weight gradients, then wait
        for the reductions that ran beside BPTT on the weight-gradient stream."""
        self._callback_queued = False
        self.join()          # (pre-activation gradients of the decoder halves were formed on the side stream)
        self.flush_window()  # THIS engine's window; other models' engines and trainers flush their own
        if sm._DEFERRED:                   # packers of THIS network's layer-by-layer modules (normally none queued)
            if self._own_packers is None:
                self._own_packers = {id(v_) for m_ in self.arch.modules() for v_ in vars(m_).values()
                                     if isinstance(v_, sm.PackedWeights)}
            for pk in list(sm._DEFERRED):
                if id(pk) in self._own_packers:
                    sm.flush_deferred_wgrads(pk)
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def flush_window(self, stream=None, parts=(3,), between=None, keep=False, only=None):
        """The deferred weight gradients of every backward call since the last flush: one reduction per layer over all
        passes (tef_net_window_wgrads) — on the current stream, or on `stream` once it has caught up with the current
        one (the caller makes its stream wait for `stream` before the gradients are read).  `parts`: the layer halves in
        the order to reduce them (2 = residual blocks / decoders / heads, 1 = encoders, 3 = all); `between()` runs after
        every half but the last — a data-parallel caller starts the all-reduce of the finished half's gradients there.
        keep: leave the backward calls queued (the caller reduces the other half with a second call)."""
        if only is None:
            self.join_encoders()           # (the encoder halves' backward may still be running on the level streams)
        pend = self._pending
        if only is not None:               # these records alone; whatever else is queued stays queued
            pend = [r for r in pend if any(r is o_ for o_ in only)]
            self._pending = [r for r in self._pending if not any(r is o_ for o_ in only)]
            for r in pend:
                r.queued = False
            if not self._pending:
                sm._DEFERRED_ENGINES.discard(self)
        elif not keep:
            self._pending = []
            sm._DEFERRED_ENGINES.discard(self)
            for r in pend:
                r.queued = False
        if not pend:
            if between is not None:
                for _ in parts[:-1]:
                    between()
            return
        if self.enc_streams is not None and not torch.cuda.is_current_stream_capturing():
            # arenas made on the level streams, read by the reductions on this one
            cur = torch.cuda.current_stream()
            for r in pend:
                for t in (r.tape, r.gtape, r.xp):
                    if t is not None:
                        t.record_stream(cur)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                if self.debug_delay and len(self.debug_delay) > 2 and self.debug_delay[2]:
                    torch.cuda._sleep(int(self.debug_delay[2]))
                for r in pend:          # the arenas are released right after this call: not before `stream` has read them
                    for t in [r.tape, r.gtape, r.xp] + list(r.states_in or []) + list(r.hn or []):
                        if t is not None:
                            t.record_stream(stream)
                self._launch_wgrads(pend, parts, between)
            return
        self._launch_wgrads(pend, parts, between)

    def _launch_wgrads(self, pend, parts=(3,), between=None):
        # the passes' own records (encoder halves, or whole passes) and the records of windows' batched decoder halves: the
        # two kinds have different batch sizes, each goes through tef_net_window_wgrads with its own plan (a layer that did
        # not run in a record is skipped there by the record's `ran` bits)
        groups = [[r for r in pend if r.hn is None], [r for r in pend if r.hn is not None]]
        null = ctypes.POINTER(ctypes.c_void_p)()
        for k, part in enumerate(parts):
            for grp in groups:
                if not grp or (grp[0].hn is not None and not (int(part) & 2)):
                    continue
                npass = len(grp)
                xs = (ctypes.c_void_p * npass)(*[_p(r.xp) for r in grp])
                sts = (ctypes.POINTER(ctypes.c_void_p) * npass)(*[
                    ctypes.cast(r.states_arr, ctypes.POINTER(ctypes.c_void_p)) if r.states_arr is not None else null for r in grp])
                tapes = (ctypes.c_void_p * npass)(*[r.tape.data_ptr() for r in grp])
                gtapes = (ctypes.c_void_p * npass)(*[r.gtape.data_ptr() for r in grp])
                rans = (ctypes.c_ulonglong * npass)(*[r.ran for r in grp])
                if grp[0].hn is not None:        # one launch set per window record (each has its own stack of states in its plan)
                    for q in range(npass):
                        nws = _lib.lib().tef_net_window_wgrads_workspace(ctypes.byref(grp[q].plan), 1)
                        if nws:
                            ws_ = self.workspace(nws, grp[q].tape.device)
                            grp[q].plan.wgrad_ws, grp[q].plan.wgrad_ws_bytes = ws_.data_ptr(), ws_.numel()
                        rc = _lib.lib().tef_net_window_wgrads_part(ctypes.byref(grp[q].plan), 2, 1, (ctypes.c_void_p * 1)(xs[q]),
                                                                   (ctypes.POINTER(ctypes.c_void_p) * 1)(sts[q]), (ctypes.c_void_p * 1)(tapes[q]),
                                                                   (ctypes.c_void_p * 1)(gtapes[q]), (ctypes.c_ulonglong * 1)(rans[q]),
                                                                   _lib.stream_ptr())
                        _lib.check(rc, "tef_net_window_wgrads")
                else:
                    pl_ = grp[-1].plan
                    nws = _lib.lib().tef_net_window_wgrads_workspace(ctypes.byref(pl_), npass)
                    if nws:              # layers reduced as one copied batch (plan.copy_batch: the deepest stride-2 head)
                        ws_ = self.workspace(nws, grp[-1].tape.device)
                        pl_.wgrad_ws, pl_.wgrad_ws_bytes = ws_.data_ptr(), ws_.numel()
                    rc = _lib.lib().tef_net_window_wgrads_part(ctypes.byref(pl_), int(part), npass, xs, sts, tapes, gtapes, rans,
                                                               _lib.stream_ptr())
                    _lib.check(rc, "tef_net_window_wgrads")
            if between is not None and k + 1 < len(parts):
                between()


class _PassFn(torch.autograd.Function):
    """(input, 4 incoming states, parameters...) -> (4 flows, 4 new states)."""

    @staticmethod
    def forward(ctx, engine, nstates, x, *rest):
        states = list(rest[:nstates])
        flows, new_states, rec = engine.forward(x, states, keep=True)
        ctx.engine, ctx.rec, ctx.nstates = engine, rec, nstates
        ctx.params = rest[nstates:]
        ctx.state_given = [s is not None for s in states]
        ctx.want_dx = bool(x.requires_grad)
        return tuple(flows) + tuple(new_states)

    @staticmethod
    def backward(ctx, *grads):
        engine, n = ctx.engine, ctx.nstates
        nflow = len(grads) - n
        dh, dx, pg = engine.backward(ctx.rec, list(grads[:nflow]), list(grads[nflow:]), ctx.params, ctx.want_dx)
        ctx.rec = None               # (a deferred window keeps its own reference until the flush)
        dh = [g if given else None for g, given in zip(dh, ctx.state_given)]
        return (None, None, dx) + tuple(dh) + tuple(pg)


class _EncFn(torch.autograd.Function):
    """Two-stream pass, first node: (input, 4 incoming states, parameters...) -> 4 new states.  `holder` receives the
    pass's record for `_DecFn`."""

    @staticmethod
    def forward(ctx, engine, nstates, holder, x, *rest):
        states = list(rest[:nstates])
        _, new_states, rec = engine.forward(x, states, keep=True, part=1)
        holder.append(rec)
        ctx.engine, ctx.rec, ctx.nstates = engine, rec, nstates
        ctx.params = rest[nstates:]
        ctx.state_given = [s is not None for s in states]
        ctx.want_dx = bool(x.requires_grad)
        ctx.set_materialize_grads(False)      # (a state nobody differentiated arrives as None, not as a tensor of zeros)
        return tuple(new_states)

    @staticmethod
    def backward(ctx, *dstates):
        # window mode: the batched decoder half has left its share of the new states' gradients at the record (_DecWinFn);
        # `dstates` are then what the NEXT pass's encoder half sends (nothing at the window's last pass)
        dec = None
        if ctx.rec is not None:      # (None: a second backward through the pass — engine.backward says so)
            dec, ctx.rec.dec_dstates = ctx.rec.dec_dstates, None
        dh, dx, pg = ctx.engine.backward(ctx.rec, None, list(dstates), ctx.params, ctx.want_dx, part=1, dstates2=dec)
        ctx.rec = None
        dh = [g if given else None for g, given in zip(dh, ctx.state_given)]
        return (None, None, None, dx) + tuple(dh) + tuple(pg)


class _EncLowFn(torch.autograd.Function):
    """Pipelined encoder half, levels [0, split) of a pass, on the engine's low stream: (input, the incoming states of these
    levels, parameters...) -> their new states.  `all_states`: the incoming states of ALL levels (the pass's record holds
    them for both ranges); `holder` receives the record for `_EncHighFn`."""

    @staticmethod
    def forward(ctx, engine, split, all_states, holder, x, *rest):
        _, new_states, rec = engine.forward(x, list(all_states), keep=True, part=1, levels=(0, split))
        holder.append(rec)
        ctx.engine, ctx.rec, ctx.split = engine, rec, split
        ctx.params = rest[split:]
        ctx.state_given = [s is not None for s in rest[:split]]
        ctx.want_dx = bool(x.requires_grad)
        ctx.set_materialize_grads(False)
        return tuple(new_states[:split])

    @staticmethod
    def backward(ctx, *dstates):
        rec, n = ctx.rec, ctx.engine.plan.levels
        dec = None
        if rec is not None:
            cur = torch.cuda.current_stream()
            # the upper levels' backward (other stream: the last thing issued there — autograd walks the nodes newest first, the
            # upper range of pass t - 1 comes after this node) left a gradient for state split - 1 in the arena
            # (through the caller's stream, which idles during BPTT: on this stack hipStreamEndCapture crashes on a capture in
            # which the low stream waits for the high one after the high one has waited for the low one —
            # tools/experiments/capture_topology_probe.py)
            if rec.high_stream is not None:
                if rec.hub is not None and rec.hub != cur:
                    rec.hub.wait_stream(rec.high_stream)
                    cur.wait_stream(rec.hub)
                else:
                    cur.wait_stream(rec.high_stream)
            if rec.dec_stream is not None and ctx.engine.dec_wait_needed(cur, rec):
                cur.wait_stream(rec.dec_stream)
            dec = rec.dec_dstates
            if dec is not None and not torch.cuda.is_current_stream_capturing():
                for g in dec:
                    if g is not None:
                        g.record_stream(cur)
        dst = list(dstates) + [None] * (n - ctx.split)
        dh, dx, pg = ctx.engine.backward(rec, None, dst, ctx.params, ctx.want_dx, part=1, dstates2=dec, levels=(0, ctx.split))
        if rec is not None:
            rec.dec_dstates = None
        ctx.rec = None
        dh = [g if given else None for g, given in zip(dh[:ctx.split], ctx.state_given)]
        return (None, None, None, None, dx) + tuple(dh) + tuple(pg)


class _EncHighFn(torch.autograd.Function):
    """Pipelined encoder half, levels [split, n), on the engine's high stream: (the new state of level split - 1, the
    incoming states of these levels, parameters...) -> their new states.  Its backward leaves the gradient w.r.t. the state
    below in the pass's gradient arena (rec.above_valid) instead of handing it to autograd, where it would be summed with
    the next pass's by a launch of its own."""

    @staticmethod
    def forward(ctx, engine, split, rec, below, *rest):
        n = engine.plan.levels
        _, new_states, _ = engine.forward(None, None, keep=True, part=1, rec=rec, levels=(split, n))
        ctx.engine, ctx.rec, ctx.split = engine, rec, split
        ctx.params = rest[n - split:]
        ctx.state_given = [s is not None for s in rest[:n - split]]
        ctx.set_materialize_grads(False)
        return tuple(new_states[split:])

    @staticmethod
    def backward(ctx, *dstates):
        rec, n, split = ctx.rec, ctx.engine.plan.levels, ctx.split
        dec = None
        if rec is not None:
            cur = torch.cuda.current_stream()
            if rec.dec_stream is not None and ctx.engine.dec_wait_needed(cur, rec):
                cur.wait_stream(rec.dec_stream)
            dec = rec.dec_dstates
            if dec is not None and not torch.cuda.is_current_stream_capturing():
                for g in dec:
                    if g is not None:
                        g.record_stream(cur)
        dst = [None] * split + list(dstates)
        dh, _, pg = ctx.engine.backward(rec, None, dst, ctx.params, False, part=1, dstates2=dec, levels=(split, n))
        if rec is not None:
            rec.high_stream = torch.cuda.current_stream()      # (the stream to wait for; see _EncLowFn.backward)
        ctx.rec = None
        dh = [g if given else None for g, given in zip(dh[split:], ctx.state_given)]
        return (None, None, None, None) + tuple(dh) + tuple(pg)


class _DecFn(torch.autograd.Function):
    """Two-stream pass, second node, on the side stream: (4 new states, parameters...) -> 4 flows."""

    @staticmethod
    def forward(ctx, engine, rec, nstates, *rest):
        cur = torch.cuda.current_stream()
        if cur != rec.stream:
            rec.tape.record_stream(cur)
        flows, _, _ = engine.forward(None, None, keep=True, part=2, rec=rec)
        ctx.engine, ctx.rec, ctx.nstates = engine, rec, nstates
        ctx.params = rest[nstates:]
        return tuple(flows)

    @staticmethod
    def backward(ctx, *dflows):
        ds, _, pg = ctx.engine.backward(ctx.rec, list(dflows), None, ctx.params, False, part=2)
        ctx.rec = None
        return (None, None, None) + tuple(ds) + tuple(pg)


class _DecWinFn(torch.autograd.Function):
    """The decoder halves of the P passes of a loss window as ONE batch (PassEngine.decode_window):
    (P x 4 new states, parameters...) -> P x 4 flows.  Its backward runs once, before any encoder half's."""

    @staticmethod
    def forward(ctx, engine, recs, nstates, *rest):
        P = len(recs)
        states = [list(rest[t * nstates:(t + 1) * nstates]) for t in range(P)]
        flows, rec = engine.decode_window(recs, states)
        ctx.engine, ctx.rec, ctx.P, ctx.nstates = engine, rec, P, nstates
        ctx.recs = list(recs)
        ctx.params = rest[P * nstates:]
        return tuple(f for t in range(P) for f in flows[t])

    @staticmethod
    def backward(ctx, *dflows):
        engine, rec, P, n = ctx.engine, ctx.rec, ctx.P, ctx.nstates
        B = rec.plan.B // P
        dfl = []
        for k in range(n):          # per head: the passes' gradients stacked along the batch (one copy per head)
            gs = [dflows[t * n + k] for t in range(P)]
            have = [g for g in gs if g is not None]
            if not have:
                dfl.append(None)
            else:
                dfl.append(torch.cat([g if g is not None else torch.zeros_like(have[0]) for g in gs], 0))
        ds, _, pg = engine.backward(rec, dfl, None, ctx.params, False, part=2)
        ctx.rec = None
        # What this half sends to the new states goes to the passes' encoder halves DIRECTLY (their records), not through
        # autograd: there it would be added to the next pass's gradient by a launch per state and pass (36 per window);
        # the encoder half's first kernel sums its addends anyway.  The encoder nodes still run — each is reached through
        # the states it produced, with None where nothing else arrives.
        # (the encoder halves' backward runs on other streams: they wait for this one — the decoders' backward is the last
        # thing issued on it when they are)
        done = torch.cuda.current_stream() if engine.enc_streams is not None else None
        token = object()
        for t, r in enumerate(ctx.recs):
            r.dec_dstates = [None if ds[i] is None else ds[i][t * B:(t + 1) * B] for i in range(n)]
            r.dec_stream, r.dec_token = done, token
        ctx.recs = None
        return (None, None, None) + (None,) * (P * n) + tuple(pg)


def encode_pass(engine, x, states, first=True):
    """The encoder half of one pass as an autograd node of its own -> (new states, the pass's record for decode_passes).
    With engine.enc_streams: as two nodes on two streams, levels [0, n/2) and [n/2, n) — the lower levels of this pass run
    beside the upper levels of the previous one (the caller joins the streams: PassEngine.join_encoders).
    first: the first pass of its window (a captured window's later passes need no new dependency on the caller's stream:
    their inputs are static buffers written before the first one)."""
    params = [p for p in engine.arch.parameters()]
    holder = []
    n = len(states)
    if engine.enc_streams is None or n < 2 or any(s is None for s in states):
        # (a fresh sequence — None states — takes the single call: its zero states are made on the caller's stream)
        new_states = _EncFn.apply(engine, n, holder, x, *states, *params)
        return list(new_states), holder[0]
    low_s, high_s = engine.enc_streams
    split = n // 2
    main = torch.cuda.current_stream()
    # the input (loader stage), the weights (optimiser step), a window's first states.  (Every cross-stream dependency is a
    # node of a captured window: spelled out at every pass they cost the replay 0.3 ms)
    if first or not torch.cuda.is_current_stream_capturing():
        low_s.wait_stream(main)
    with torch.cuda.stream(low_s):
        low = _EncLowFn.apply(engine, split, tuple(states), holder, x, *states[:split], *params)
    rec = holder[0]
    rec.hub = main
    if isinstance(x, torch.Tensor) and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        x.record_stream(low_s)
    # everything issued on the low stream so far — this pass's lower levels are its last piece, and what the low stream has
    # waited for on the caller's stream comes with it
    high_s.wait_stream(low_s)
    with torch.cuda.stream(high_s):
        high = _EncHighFn.apply(engine, split, rec, low[split - 1], *states[split:], *params)
    return list(low) + list(high), rec


def decode_passes(engine, recs, states):
    """-> flows[t][k]: the decoder halves of the passes `recs` (encode_pass) as one batch."""
    params = [p for p in engine.arch.parameters()]
    n = len(states[0])
    out = _DecWinFn.apply(engine, list(recs), n, *[s for st in states for s in st], *params)
    return [list(out[t * n:(t + 1) * n]) for t in range(len(recs))]


def run_pass(engine, x, states):
    """Differentiable pass when gradients are enabled, plain launches otherwise."""
    params = [p for p in engine.arch.parameters()]
    engine.last_pass_split = False      # True: the flows of this pass live on the side stream (train.Trainer reads it)
    needs = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params) or
                                         any(s is not None and s.requires_grad for s in states))
    side = engine.side_stream
    if needs:
        if engine._callback_queued:
            # a backward() that queued its end-of-backward flush never finished (it raised: autograd drops the callbacks of a
            # failed graph task).  Its records are stale — their weight gradients belong to a window that was abandoned —
            # and the next backward must queue a flush of its own
            engine._callback_queued = False
            for r_ in engine._pending:
                r_.queued = False
            engine._pending = []
            sm._DEFERRED_ENGINES.discard(engine)
        engine.arch.own_gradients()          # (the drop-in loop: .grad views of the network's own flat buffer, see arch.py)
    if not needs:
        if side is None or not engine.defer_join:       # (without gradients the split only pays when the caller overlaps)
            flows, new_states, _ = engine.forward(x, list(states), keep=False)
            return flows, new_states
        engine.last_pass_split = True
        _, new_states, rec = engine.forward(x, list(states), keep=False, part=1)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rec.tape.record_stream(side)
            flows, _, _ = engine.forward(None, None, keep=False, part=2, rec=rec)
        return flows, new_states
    if side is not None and engine.arch.direct_grads and all(p.grad is not None and p.grad.is_contiguous()
                                                             for p in params if p.requires_grad):
        engine.last_pass_split = True
        holder = []
        delay = engine.debug_delay or (0, 0)
        if delay[0]:
            torch.cuda._sleep(int(delay[0]))
        new_states = _EncFn.apply(engine, len(states), holder, x, *states, *params)
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if delay[1]:
                torch.cuda._sleep(int(delay[1]))
            flows = _DecFn.apply(engine, holder[0], len(states), *new_states, *params)
        if engine.lazy_flows and not engine.defer_join:
            # the literal train_flow.py loop (no Trainer): the flows stay on the side stream as LazyFlow tensors — scaling
            # them and the loss container's update() run there, anything else joins first (models/lazy.py)
            return [LazyFlow.wrap(f, side) for f in flows], list(new_states)
        if not engine.defer_join:
            main.wait_stream(side)
        return list(flows), list(new_states)
    out = _PassFn.apply(engine, len(states), x, *states, *params)
    nflow = len(out) - len(states)
    return list(out[:nflow]), list(out[nflow:])
