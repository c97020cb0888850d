"""One recurrent pass of RecEVFlowNet as ONE autograd node with a hand-written backward.

The reference runs the pass as ~60 autograd nodes (models/arch.py:217-242, models/model.py:65-85) and lets autograd sum
the gradients of every tensor that has several consumers (encoder states: next encoder + decoder skip + next pass;
decoder outputs: prediction head + next decoder; head outputs: two gate convolutions; ...).  Here the pass is a static
sequence of libtef_hip.so launches over the layer table of `arch.NetPlan`:

  forward   [pad] -> 4 x (strided head conv, fused ConvGRU cell) -> 2 residual blocks -> 4 x (bilinear x2 of
            (features + encoder skip) [+ bilinear x2 of the previous prediction], conv over the two sources without
            materialising their concatenation, 1x1 tanh head) -> 4 x (bilinear to the input size, x 2^(3-k), crop)
  backward  the same table walked in reverse.  A gradient with several producers is never accumulated by a separate
            pass: the kernel that CONSUMES it (tef_grad_act, tef_convgru_cell_bwd) takes up to four addends.  Bias
            gradients come out of those same sweeps; weight gradients go straight into the parameters' .grad buffers
            (or, inside a BPTT window, into one tef_conv_wgrad_parts reduction per layer over all passes).

No ATen arithmetic runs inside a pass (padding a non-multiple-of-16 input is one strided copy); torch provides the
buffers and the stream.  All scratch lives in one workspace per engine, reused by every launch (stream order).
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


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr, len(tensors)


class _Tape:
    """Activations one pass keeps for its backward."""

    __slots__ = ("xp", "enc", "res", "dec", "geom", "dirty")

    def __init__(self):
        self.enc, self.res, self.dec = [], [], []


class PassEngine:
    def __init__(self, arch):
        self.arch = arch
        self.plan = arch.plan
        self._ws = None
        self._zeros = {}

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

    # ---- primitive launches ------------------------------------------------------------------------------------
    def conv_fwd(self, packer, weights, biases, x0, x1, stride, act):
        lib = _lib.lib()
        B, C0, H, W = x0.shape
        C1 = x1.shape[1] if x1 is not None else 0
        w = weights[0]
        N, k = sum(t.shape[0] for t in weights), w.shape[2]
        d = _lib.ConvDesc(B, C0, C1, H, W, N, k, stride, ACT[act])
        wp, _ = packer.get(weights, d)
        bias = packer.bias(biases)
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((B, N, Ho, Wo), dtype=torch.float32, device=x0.device)
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = self.workspace(nbytes, x0.device)
        rc = lib.tef_conv_forward(ctypes.byref(d), x0.data_ptr(), _p(x1), None, wp.data_ptr(), _p(bias), out.data_ptr(),
                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_conv_forward")
        return out

    def conv_bwd(self, packer, weights, g, x0, x1, stride, want_dx, sink):
        """Input gradients of a convolution whose pre-activation gradient `g` is already formed; weight gradients go to
        `sink` (immediately, or queued for the window's per-layer reduction)."""
        lib = _lib.lib()
        B, C0, H, W = x0.shape
        C1 = x1.shape[1] if x1 is not None else 0
        w = weights[0]
        N, k = sum(t.shape[0] for t in weights), w.shape[2]
        d = _lib.ConvDesc(B, C0, C1, H, W, N, k, stride, ACT[None])
        _, wt = packer.get(weights, d)
        dx0 = torch.empty_like(x0) if want_dx else None
        dx1 = torch.empty_like(x1) if (want_dx and x1 is not None) else None
        dws = sink.weight_targets(packer, weights, d, (g, x0, x1, None))
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = self.workspace(nbytes, x0.device)
        rc = lib.tef_conv_backward_keep(ctypes.byref(d), x0.data_ptr(), _p(x1), None, wt.data_ptr(), None, None,
                                        g.data_ptr(), None, N, _p(dx0), _p(dx1), _p(dws[0]), None, None, None, N, None,
                                        ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_conv_backward_keep")
        return dx0, dx1

    def grad_act(self, sources, out, act, dbias):
        B, C = out.shape[0], out.shape[1]
        HW = out.shape[2] * out.shape[3]
        g = torch.empty_like(out)
        arr, n = _ptr_array(sources)
        rc = _lib.lib().tef_grad_act(arr, n, out.data_ptr(), ACT[act], B, C, HW, g.data_ptr(), _p(dbias), _lib.stream_ptr())
        _lib.check(rc, "tef_grad_act")
        return g

    def upsample(self, x, x2, scale, mul=1.0, crop=(0, 0)):
        B, C, H, W = x.shape
        y = torch.empty((B, C, H * scale - crop[0], W * scale - crop[1]), dtype=torch.float32, device=x.device)
        rc = _lib.lib().tef_upsample_bilinear_crop(x.data_ptr(), _p(x2), B * C, H, W, scale, scale, float(mul), crop[0],
                                                   crop[1], y.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_upsample_bilinear_crop")
        return y

    def upsample_bwd(self, dy, shape, scale, mul=1.0, crop=(0, 0)):
        B, C, H, W = shape
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
        rc = _lib.lib().tef_upsample_bilinear_crop_backward(dy.data_ptr(), B * C, H, W, scale, scale, float(mul), crop[0],
                                                            crop[1], dx.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_upsample_bilinear_crop_backward")
        return dx

    def add_act(self, a, b, act):
        out = torch.empty_like(a)
        rc = _lib.lib().tef_add_act(a.data_ptr(), b.data_ptr(), ACT[act], a.numel(), out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_add_act")
        return out

    def cell_fwd(self, gru, x, h):
        lib = _lib.lib()
        B, C, H, W = x.shape
        d = _lib.GruDesc(B, C, H, W)
        d_ur = _lib.ConvDesc(B, C, C, H, W, 2 * C, 3, 1, ACT["sigmoid"])
        d_o = _lib.ConvDesc(B, C, C, H, W, C, 3, 1, ACT["tanh"])
        w_ur = (gru.update_gate.weight, gru.reset_gate.weight)
        wp_ur, _ = gru._packed_ur.get(w_ur, d_ur)
        wp_o, _ = gru._packed_o.get((gru.out_gate.weight,), d_o)
        b_ur = gru._packed_ur.bias((gru.update_gate.bias, gru.reset_gate.bias))
        b_o = gru._packed_o.bias((gru.out_gate.bias,))
        u, r, o, hn = (torch.empty_like(x) for _ in range(4))
        nbytes = lib.tef_convgru_workspace_bytes(ctypes.byref(d))
        ws = self.workspace(nbytes, x.device)
        rc = lib.tef_convgru_cell_fwd(ctypes.byref(d), x.data_ptr(), h.data_ptr(), wp_ur.data_ptr(), wp_o.data_ptr(),
                                      _p(b_ur), _p(b_o), u.data_ptr(), r.data_ptr(), o.data_ptr(), hn.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_convgru_cell_fwd")
        return u, r, o, hn

    def cell_bwd(self, gru, x, h, u, r, o, sources, sink):
        lib = _lib.lib()
        B, C, H, W = x.shape
        d = _lib.GruDesc(B, C, H, W)
        d_ur = _lib.ConvDesc(B, C, C, H, W, 2 * C, 3, 1, ACT[None])
        d_o = _lib.ConvDesc(B, C, C, H, W, C, 3, 1, ACT[None])
        w_ur = (gru.update_gate.weight, gru.reset_gate.weight)
        _, wt_ur = gru._packed_ur.get(w_ur, d_ur)
        _, wt_o = gru._packed_o.get((gru.out_gate.weight,), d_o)
        g_ur = torch.empty((B, 2 * C, H, W), dtype=torch.float32, device=x.device)
        g_o, dx, dh = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dw_ur = sink.weight_targets(gru._packed_ur, w_ur, d_ur, (g_ur, x, h, None))
        dw_o = sink.weight_targets(gru._packed_o, (gru.out_gate.weight,), d_o, (g_o, x, h, r))
        db = [sink.bias_target(b) for b in (gru.update_gate.bias, gru.reset_gate.bias, gru.out_gate.bias)]
        arr, n = _ptr_array(sources)
        nbytes = lib.tef_convgru_workspace_bytes(ctypes.byref(d))
        ws = self.workspace(nbytes, x.device)
        rc = lib.tef_convgru_cell_bwd(ctypes.byref(d), x.data_ptr(), h.data_ptr(), u.data_ptr(), r.data_ptr(), o.data_ptr(),
                                      arr, n, wt_ur.data_ptr(), wt_o.data_ptr(), g_ur.data_ptr(), g_o.data_ptr(),
                                      dx.data_ptr(), dh.data_ptr(), _p(dw_ur[0]), _p(dw_ur[1]), _p(dw_o[0]), _p(db[0]),
                                      _p(db[1]), _p(db[2]), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "tef_convgru_cell_bwd")
        return dx, dh

    # ---- the pass ----------------------------------------------------------------------------------------------
    def forward(self, x, states, keep):
        """x [B, bins, H, W] -> (flows: 4 x [B, 2, H, W], new states: 4 x [B, C_i, h_i, w_i], tape | None)."""
        a, plan = self.arch, self.plan
        _lib.require_device_tensor(x, "network input")
        x = x.contiguous()
        B, _, H, W = x.shape
        ph, pw = plan.padding(H, W)
        if ph or pw:        # E-RAFT style padding at the top / left (reference models/model_util.py:52-65)
            xp = x.new_zeros((B, x.shape[1], H + ph, W + pw))
            xp[:, :, ph:, pw:] = x
        else:
            xp = x
        tape = _Tape() if keep else None
        cur, hn_all = xp, []
        for i, enc in enumerate(a.encoders):
            e = self.conv_fwd(enc.conv._packed, (enc.conv.conv2d.weight,), (enc.conv.conv2d.bias,), cur, None, plan.stride, "relu")
            h = states[i]
            if h is None:
                h = self.zero_state(e.shape, e.device)
            u, r, o, hn = self.cell_fwd(enc.recurrent_block, e, h.contiguous())
            if keep:
                tape.enc.append((cur, e, h, u, r, o, hn))
            hn_all.append(hn)
            cur = hn
        for rb in a.resblocks:
            mid = self.conv_fwd(rb._packed1, (rb.conv1.weight,), (rb.conv1.bias,), cur, None, 1, "relu")
            lin = self.conv_fwd(rb._packed2, (rb.conv2.weight,), (rb.conv2.bias,), mid, None, 1, None)
            y = self.add_act(lin, cur, "relu")
            if keep:
                tape.res.append((cur, mid, y))
            cur = y
        flows, pred = [], None
        nlev = len(a.decoders)
        for k, (dec, head) in enumerate(zip(a.decoders, a.preds)):
            skip = hn_all[nlev - 1 - k]
            upx = self.upsample(cur, skip, 2)
            upp = self.upsample(pred, None, 2) if pred is not None else None
            x0, x1 = (upp, upx) if upp is not None else (upx, None)
            d = self.conv_fwd(dec._packed, (dec.conv2d.weight,), (dec.conv2d.bias,), x0, x1, 1, "relu")
            p = self.conv_fwd(head._packed, (head.conv2d.weight,), (head.conv2d.bias,), d, None, 1, plan.final_activation)
            s = 2 ** (nlev - 1 - k)
            mul = float(s) * a.flow_scale          # resolution factor (model.py:76-81) x the caller's flow scaling
            flows.append(self.upsample(p, None, s, mul=mul, crop=(ph, pw)))
            if keep:
                tape.dec.append((cur.shape, x0, x1, d, p, s, mul))
            cur, pred = d, p
        if keep:
            tape.geom = (ph, pw)
        return flows, hn_all, tape

    def backward(self, tape, dflows, dstates, sink, want_dx=False):
        """-> (gradients w.r.t. the incoming states (4), gradient w.r.t. the padded network input or None).  Parameter
        gradients go through `sink`.  want_dx: also run the first encoder's input-gradient contraction (the reference's
        autograd delivers d loss / d input when the caller asks for it, models/arch.py:217-227)."""
        a = self.arch
        ph, pw = tape.geom
        nlev = len(a.decoders)
        skip_grads = [None] * nlev          # d loss / d (features + encoder skip) of decoder k, shared by both addends
        d_prev_pred = None                  # gradient arriving at prediction k from decoder k + 1
        d_feat = None                       # gradient arriving at decoder k's output from decoder k + 1
        for k in range(nlev - 1, -1, -1):
            dec, head = a.decoders[k], a.preds[k]
            src_shape, x0, x1, d, p, s, mul = tape.dec[k]
            srcs = []
            if dflows[k] is not None:
                srcs.append(self.upsample_bwd(dflows[k].contiguous(), p.shape, s, mul=mul, crop=(ph, pw)))
            if d_prev_pred is not None:
                srcs.append(d_prev_pred)
            feat_srcs = [d_feat] if d_feat is not None else []
            if srcs:
                gp = self.grad_act(srcs, p, self.plan.final_activation, sink.bias_target(head.conv2d.bias))
                dd, _ = self.conv_bwd(head._packed, (head.conv2d.weight,), gp, d, None, 1, True, sink)
                feat_srcs.insert(0, dd)
            if not feat_srcs:           # nothing reaches this level (all its flow gradients absent): the chain is dead here
                skip_grads[k] = None
                d_prev_pred = d_feat = None
                continue
            gd = self.grad_act(feat_srcs, d, "relu", sink.bias_target(dec.conv2d.bias))
            dx0, dx1 = self.conv_bwd(dec._packed, (dec.conv2d.weight,), gd, x0, x1, 1, True, sink)
            dupp, dupx = (dx0, dx1) if x1 is not None else (None, dx0)
            skip_grads[k] = self.upsample_bwd(dupx, src_shape, 2)
            d_feat = skip_grads[k]
            d_prev_pred = self.upsample_bwd(dupp, tape.dec[k - 1][4].shape, 2) if dupp is not None else None
        # residual blocks, last first; `extra` = gradients reaching the block input besides its first convolution
        srcs = [skip_grads[0]] if skip_grads[0] is not None else []
        for rb, (xin, mid, y) in zip(reversed(list(a.resblocks)), reversed(tape.res)):
            if not srcs:
                break
            gy = self.grad_act(srcs, y, "relu", sink.bias_target(rb.conv2.bias))
            dmid, _ = self.conv_bwd(rb._packed2, (rb.conv2.weight,), gy, mid, None, 1, True, sink)
            gmid = self.grad_act([dmid], mid, "relu", sink.bias_target(rb.conv1.bias))
            dxin, _ = self.conv_bwd(rb._packed1, (rb.conv1.weight,), gmid, xin, None, 1, True, sink)
            srcs = [dxin, gy]
        # encoders, deepest first
        dh_in = [None] * nlev
        from_above = srcs                   # gradient of the deepest state through the residual blocks
        for i in range(nlev - 1, -1, -1):
            enc = a.encoders[i]
            xin, e, h, u, r, o, hn = tape.enc[i]
            sources = list(from_above)
            sg = skip_grads[nlev - 1 - i]
            # (the deepest state is decoder 0's skip addend AND the input of the residual blocks; without residual blocks
            # `from_above` is that same gradient once more — features + skip = 2 x the state — and it is added twice)
            if sg is not None:
                sources.append(sg)
            if dstates[i] is not None:
                sources.append(dstates[i].contiguous())
            if not sources:
                from_above = []
                continue
            de, dh_in[i] = self.cell_bwd(enc.recurrent_block, e, h, u, r, o, sources, sink)
            ge = self.grad_act([de], e, "relu", sink.bias_target(enc.conv.conv2d.bias))
            dxin, _ = self.conv_bwd(enc.conv._packed, (enc.conv.conv2d.weight,), ge, xin, None, self.plan.stride,
                                    i > 0 or want_dx, sink)
            from_above = [dxin] if dxin is not None else []
        # (what is left in `from_above` after level 0 is the gradient of the padded input)
        return dh_in, (from_above[0] if want_dx and from_above else None)


class GradSink:
    """Where one backward pass puts its parameter gradients.

    `direct`: the parameters own pre-allocated .grad buffers (train.Trainer's flat bucket) and the kernels add into them;
    otherwise fresh zero tensors are filled and handed back to autograd.  `deferred` (with direct): weight gradients are
    queued per layer and reduced over all passes of the window by submodules.flush_deferred_wgrads()."""

    def __init__(self, params, direct, deferred):
        self.direct, self.deferred = direct, deferred and direct
        self.fresh = {}
        self.params = params

    def _target(self, p):
        if self.direct:
            return p.grad
        t = self.fresh.get(id(p))
        if t is None:
            t = self.fresh[id(p)] = torch.zeros_like(p)
        return t

    def bias_target(self, b):
        if b is None or not b.requires_grad:
            return None
        return self._target(b)

    def weight_targets(self, packer, weights, desc, part):
        """-> [dw, dw2] device tensors to accumulate into now, or [None, None] when the layer's gradient is queued."""
        if not all(w.requires_grad for w in weights):
            return [None, None]
        if self.deferred and _lib.lib().tef_conv_wgrad_parts_supported(ctypes.byref(desc)):
            meta = (desc, [w.grad for w in weights], weights[0].shape[0] if len(weights) == 2 else desc.N)
            if packer.pending and (packer.pending_meta[0].B, packer.pending_meta[0].H, packer.pending_meta[0].W) != (desc.B, desc.H, desc.W):
                sm.flush_deferred_wgrads(packer)
            packer.pending_meta = meta
            packer.pending.append(part)
            sm._DEFERRED.add(packer)
            if len(packer.pending) == sm._MAX_PARTS:
                sm.flush_deferred_wgrads(packer)
            return [None, None]
        t = [self._target(w) for w in weights]
        return t + [None] * (2 - len(t))

    def grads(self):
        if self.direct:
            return [None] * len(self.params)
        return [self.fresh.get(id(p)) for p in self.params]


class _PassFn(torch.autograd.Function):
    """(input, 4 incoming states, parameters...) -> (4 flows, 4 new states)."""

    @staticmethod
    def forward(ctx, engine, nstates, x, *rest):
        states = list(rest[:nstates])
        flows, new_states, tape = engine.forward(x, states, keep=True)
        ctx.engine, ctx.tape, ctx.nstates = engine, tape, nstates
        ctx.params = rest[nstates:]
        ctx.state_given = [s is not None for s in states]
        ctx.x_shape = tuple(x.shape) if x.requires_grad else None
        return tuple(flows) + tuple(new_states)

    @staticmethod
    def backward(ctx, *grads):
        engine, n = ctx.engine, ctx.nstates
        nflow = len(grads) - n
        params = ctx.params
        direct = all(p.grad is not None and p.grad.is_contiguous() for p in params if p.requires_grad) and engine.arch.direct_grads
        sink = GradSink(params, direct, engine.arch.deferred_wgrad)
        ph, pw = ctx.tape.geom
        dh, dxp = engine.backward(ctx.tape, list(grads[:nflow]), list(grads[nflow:]), sink, want_dx=ctx.x_shape is not None)
        ctx.tape = None
        dh = [g if given else None for g, given in zip(dh, ctx.state_given)]
        dx = None
        if ctx.x_shape is not None:
            # no gradient reached the first encoder (every flow / state gradient absent): zeros, like autograd's own
            dx = dxp[:, :, ph:, pw:] if dxp is not None else torch.zeros(ctx.x_shape, dtype=torch.float32, device=params[0].device)
        return (None, None, dx) + tuple(dh) + tuple(sink.grads())


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
