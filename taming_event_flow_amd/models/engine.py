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
"""

import ctypes

import torch

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib

from . import submodules as sm

ACT = _lib.ACT


def _p(t):
    return t.data_ptr() if t is not None else None


class _Tape:
    """What one forward pass leaves for its backward: the plan it ran with, its inputs and its activation arena."""

    __slots__ = ("plan", "xp", "states_in", "states_arr", "tape", "geom", "x_shape")


class PassEngine:
    def __init__(self, arch):
        self.arch = arch
        self.plan = arch.plan
        self._ws = None
        self._zeros = {}
        self._layout = {}
        self._pending = []            # backward calls of the current window whose weight gradients are still deferred

    # ---- buffers -----------------------------------------------------------------------------------------------
    def workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty((max(nbytes, 1 << 20),), dtype=torch.uint8, device=device)
        return self._ws

    def zero_state(self, shape, device):
        key = (tuple(shape), device)
        z = self._zeros.get(key)
        if z is None:
            z = self._zeros[key] = torch.zeros(shape, dtype=torch.float32, device=device)
        return z

    # ---- the plan ----------------------------------------------------------------------------------------------
    def _conv_fields(self, nc, packer, weights, biases, desc):
        wp, wt = packer.get(weights, desc)
        bias = packer.bias(biases)
        nc.wp, nc.w2, nc.bias = wp.data_ptr(), wt.data_ptr(), _p(bias)

    def make_plan(self, B, Hp, Wp, ph, pw):
        """struct tef_net_plan for a (padded) input of B x bins x Hp x Wp: geometry + the packed weights / biases of
        every convolution (re-packed by PackedWeights only when a parameter changed)."""
        a, np_ = self.arch, self.plan
        pl = _lib.NetPlan()
        pl.B, pl.H, pl.W = B, Hp, Wp
        pl.bins, pl.levels, pl.nres, pl.nout = np_.num_bins, np_.levels, np_.nres, np_.nout
        pl.final_act = ACT[np_.final_activation]
        pl.crop_top, pl.crop_left, pl.flow_scale = ph, pw, float(a.flow_scale)
        dec_rows = np_.of("dec")
        for i, c in enumerate(np_.width):
            pl.width[i] = c
            pl.dec_out[i] = dec_rows[i].cout
        cin, h, w = np_.num_bins, Hp, Wp
        for i, enc in enumerate(a.encoders):
            c = np_.width[i]
            self._conv_fields(pl.head[i], enc.conv._packed, (enc.conv.conv2d.weight,), (enc.conv.conv2d.bias,),
                              _lib.ConvDesc(B, cin, 0, h, w, c, 3, np_.stride, ACT["relu"]))
            h, w = h // 2, w // 2
            g = enc.recurrent_block
            self._conv_fields(pl.gate_ur[i], g._packed_ur, (g.update_gate.weight, g.reset_gate.weight),
                              (g.update_gate.bias, g.reset_gate.bias), _lib.ConvDesc(B, c, c, h, w, 2 * c, 3, 1, ACT["sigmoid"]))
            self._conv_fields(pl.gate_o[i], g._packed_o, (g.out_gate.weight,), (g.out_gate.bias,),
                              _lib.ConvDesc(B, c, c, h, w, c, 3, 1, ACT["tanh"]))
            cin = c
        top = np_.width[-1]
        for j, rb in enumerate(a.resblocks):
            d = _lib.ConvDesc(B, top, 0, h, w, top, 3, 1, ACT["relu"])
            self._conv_fields(pl.res1[j], rb._packed1, (rb.conv1.weight,), (rb.conv1.bias,), d)
            self._conv_fields(pl.res2[j], rb._packed2, (rb.conv2.weight,), (rb.conv2.bias,), d)
        src = top
        for k, (dec, head) in enumerate(zip(a.decoders, a.preds)):
            h, w = h * 2, w * 2
            c0, c1 = (np_.nout, src) if k else (src, 0)
            out = dec_rows[k].cout
            self._conv_fields(pl.dec[k], dec._packed, (dec.conv2d.weight,), (dec.conv2d.bias,),
                              _lib.ConvDesc(B, c0, c1, h, w, out, 3, 1, ACT["relu"]))
            self._conv_fields(pl.pred[k], head._packed, (head.conv2d.weight,), (head.conv2d.bias,),
                              _lib.ConvDesc(B, out, 0, h, w, np_.nout, 1, 1, ACT[np_.final_activation]))
            src = out
        return pl

    def layout(self, pl):
        """(tape floats, gradient-arena floats, workspace bytes, flow / state / dstate offsets, dx offset) of a geometry."""
        key = (pl.B, pl.H, pl.W, pl.crop_top, pl.crop_left)
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
    def forward(self, x, states, keep):
        """x [B, bins, H, W] -> (flows: 4 x [B, 2, H, W], new states: 4 x [B, C_i, h_i, w_i], tape | None)."""
        plan = self.plan
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
        pl = self.make_plan(B, H + ph, W + pw, ph, pw)
        ntape, _, wsb, fo, so, _, _ = self.layout(pl)
        n = plan.levels
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
        arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in st])
        tape = torch.empty((ntape,), dtype=torch.float32, device=x.device)
        ws = self.workspace(wsb, x.device)
        rc = _lib.lib().tef_net_pass_forward(ctypes.byref(pl), xp.data_ptr(), arr, tape.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _lib.stream_ptr())
        _lib.check(rc, "tef_net_pass_forward")
        fshape = (B, plan.nout, H, W)
        flows = [tape[fo[k]:fo[k] + B * plan.nout * H * W].view(fshape) for k in range(n)]
        new_states = [tape[so[i]:so[i] + st[i].numel()].view(st[i].shape) for i in range(n)]
        rec = None
        if keep:
            rec = _Tape()
            rec.plan, rec.xp, rec.states_in, rec.states_arr, rec.tape = pl, xp, st, arr, tape
            rec.geom, rec.x_shape = (ph, pw), tuple(x.shape)
        return flows, new_states, rec

    def backward(self, rec, dflows, dstates, params, want_dx):
        """-> (gradients w.r.t. the incoming states, gradient w.r.t. the network input or None, parameter gradients for
        autograd).  Parameter gradients are added into the parameters' own .grad buffers when the training loop owns them
        (`direct_grads`: nothing is handed to autograd), otherwise into one fresh zero buffer whose views are returned."""
        a, plan = self.arch, self.plan
        n = plan.levels
        pl = rec.plan
        direct = a.direct_grads and all(p.grad is not None and p.grad.is_contiguous() for p in params if p.requires_grad)
        fresh = None
        if direct:
            self._grad_targets(pl, lambda p: p.grad if p.requires_grad else None, a.deferred_wgrad)
        else:
            tot = sum(p.numel() for p in params if p.requires_grad)
            flat = torch.zeros((tot,), dtype=torch.float32, device=rec.tape.device)
            fresh, o = {}, 0
            for p in params:
                if p.requires_grad:
                    fresh[id(p)] = flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
            self._grad_targets(pl, lambda p: fresh.get(id(p)), False)
        _, ngt, wsb, _, _, do, dxo = self.layout(pl)
        gtape = torch.empty((ngt,), dtype=torch.float32, device=rec.tape.device)
        dfl = [None if g is None else g.to(torch.float32).contiguous() for g in dflows]
        dst = [None if g is None else g.to(torch.float32).contiguous() for g in dstates]
        a_dfl = (ctypes.c_void_p * n)(*[_p(g) for g in dfl])
        a_dst = (ctypes.c_void_p * n)(*[_p(g) for g in dst])
        ran = ctypes.c_ulonglong(0)
        dsv = (ctypes.c_int * n)()
        dxv = ctypes.c_int(0)
        ws = self.workspace(wsb, rec.tape.device)
        rc = _lib.lib().tef_net_pass_backward(ctypes.byref(pl), rec.xp.data_ptr(), rec.states_arr, rec.tape.data_ptr(), a_dfl,
                                              a_dst, 1 if want_dx else 0, gtape.data_ptr(), ctypes.byref(ran), dsv,
                                              ctypes.byref(dxv), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_net_pass_backward")
        dh = [gtape[do[i]:do[i] + rec.states_in[i].numel()].view(rec.states_in[i].shape) if dsv[i] else None for i in range(n)]
        dx = None
        if want_dx:
            ph, pw = rec.geom
            if dxv.value:
                dx = gtape[dxo:dxo + rec.xp.numel()].view(rec.xp.shape)[:, :, ph:, pw:]
            else:       # no gradient reached the first encoder: zeros, like autograd's own
                dx = torch.zeros(rec.x_shape, dtype=torch.float32, device=rec.tape.device)
        if direct and a.deferred_wgrad:
            if self._pending and (self._pending[0][0].B, self._pending[0][0].H, self._pending[0][0].W) != (pl.B, pl.H, pl.W):
                self.flush_window()
            self._pending.append((pl, rec, gtape, ran.value))       # (keeps the arenas alive until the flush)
            sm._DEFERRED_ENGINES.add(self)
        grads = [None] * len(params) if direct else [fresh.get(id(p)) for p in params]
        return dh, dx, grads

    def flush_window(self):
        """The deferred weight gradients of every backward call since the last flush: one reduction per layer over all
        passes (tef_net_window_wgrads)."""
        pend, self._pending = self._pending, []
        sm._DEFERRED_ENGINES.discard(self)
        if not pend:
            return
        npass = len(pend)
        xs = (ctypes.c_void_p * npass)(*[r.xp.data_ptr() for _, r, _, _ in pend])
        sts = (ctypes.POINTER(ctypes.c_void_p) * npass)(*[ctypes.cast(r.states_arr, ctypes.POINTER(ctypes.c_void_p))
                                                          for _, r, _, _ in pend])
        tapes = (ctypes.c_void_p * npass)(*[r.tape.data_ptr() for _, r, _, _ in pend])
        gtapes = (ctypes.c_void_p * npass)(*[g.data_ptr() for _, _, g, _ in pend])
        rans = (ctypes.c_ulonglong * npass)(*[m for _, _, _, m in pend])
        rc = _lib.lib().tef_net_window_wgrads(ctypes.byref(pend[-1][0]), npass, xs, sts, tapes, gtapes, rans, _lib.stream_ptr())
        _lib.check(rc, "tef_net_window_wgrads")


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


def run_pass(engine, x, states):
    """Differentiable pass when gradients are enabled, plain launches otherwise."""
    params = [p for p in engine.arch.parameters()]
    needs = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params) or
                                         any(s is not None and s.requires_grad for s in states))
    if not needs:
        flows, new_states, _ = engine.forward(x, list(states), keep=False)
        return flows, new_states
    out = _PassFn.apply(engine, len(states), x, *states, *params)
    nflow = len(out) - len(states)
    return list(out[:nflow]), list(out[nflow:])
