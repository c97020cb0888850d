"""Building blocks of RecEVFlowNet — drop-in for the reference's ``models/submodules.py``.

Same classes, constructor arguments, parameter names / shapes (so ``state_dict`` keys match) and initialisation
(ConvLayer :33-39 uniform(+-sqrt(1/Cin)) or ``w_scale``; ConvGRU :127-132 orthogonal weights, zero bias).
Every convolution (forward, input gradient, weight gradient) runs on the fp32 MFMA GEMM of libtef_hip.so
(tef_conv.hip); only tensor plumbing (padding, concatenation, bilinear resize, additions) is left to torch.
"""

import ctypes
import math

import torch
import torch.nn as nn

try:
    from .. import _lib
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else None


class PackedWeights:
    """GEMM operands of one convolution's weight (include/tef.h tef_conv_pack_weight), re-packed only when a
    parameter changed (optimizer step / load_state_dict), i.e. once per loss window instead of once per call."""

    def __init__(self):
        self.key = None
        self.wp = self.wt = None
        # training loops that own pre-allocated .grad buffers (train.Trainer's flat bucket) switch this on: the weight /
        # bias gradients are then accumulated by the kernel straight into .grad (no temporary, no autograd add)
        self.direct_grads = False
        # with direct_grads: weight gradients of the backward calls of one BPTT window are computed together, as one long
        # pixel reduction per layer (flush_deferred_wgrads) instead of one short, atomics-heavy reduction per pass
        self.defer_wgrad = False
        self.pending = []          # (g, x0, x1, gate1) tensors of the queued backward calls
        self.pending_meta = None   # (desc, weight .grad tensors, rows of the first weight)
        self.bias_key = self.bias_cat = None

    def invalidate(self):
        """Forget the packed copies (after parameters were written behind autograd's back, e.g. through `.data`)."""
        self.key = self.bias_key = None

    def __getstate__(self):
        """Checkpoints (torch.save(model), reference utils/utils.py:60-61) carry parameters only: packed copies, queued
        gradient parts and training-loop switches are rebuilt on use."""
        return {"packed": None}     # (a non-empty state, so that __setstate__ runs when the checkpoint is loaded)

    def __setstate__(self, state):
        self.__init__()

    def bias(self, biases):
        """The bias vector of a (row-concatenated) convolution: the parameter itself, or a cached concatenation."""
        if not biases or biases[0] is None:
            return None
        if len(biases) == 1:
            return biases[0]
        key = tuple((b.data_ptr(), b._version) for b in biases)
        if key != self.bias_key:
            self.bias_cat = torch.cat([b.detach() for b in biases])
            self.bias_key = key
        return self.bias_cat

    def get(self, weights, desc, batch=None):
        """-> (wp, w2).  `batch` (a list): a stale layer only queues its pack jobs there — the caller runs them all in ONE
        launch with run_pack_jobs(batch) before anything reads the operands (models/engine.py: a network after an optimiser
        step)."""
        key = tuple((w.data_ptr(), w._version) for w in weights) + (desc.C0 + desc.C1, desc.N, desc.ksize)
        if key != self.key:
            lib = _lib.lib()
            np_, nt = ctypes.c_size_t(), ctypes.c_size_t()
            lib.tef_conv_packed_weight_floats(ctypes.byref(desc), ctypes.byref(np_), ctypes.byref(nt))
            dev = weights[0].device
            if self.wp is None or self.wp.numel() != np_.value or self.wt.numel() != nt.value or self.wp.device != dev:
                self.wp = torch.empty((np_.value,), dtype=torch.float32, device=dev)
                self.wt = torch.empty((nt.value,), dtype=torch.float32, device=dev)
            row0, jobs = 0, []
            for w in weights:
                wc = w.detach().contiguous()
                jobs.append((desc, wc, wc.shape[0], row0, self.wp, self.wt))
                row0 += wc.shape[0]
            if batch is None:
                run_pack_jobs(jobs)
            else:
                batch.extend(jobs)
            self.key = key
        return self.wp, self.wt


def run_pack_jobs(jobs):
    """The queued weight parts (PackedWeights.get) in one launch on the current stream: include/tef.h tef_conv_pack_weights."""
    if not jobs:
        return
    arr = (_lib.PackJob * len(jobs))()
    for a, (desc, wc, rows, row0, wp, wt) in zip(arr, jobs):
        a.desc = desc
        a.weight, a.rows, a.row0, a.wp, a.w2 = wc.data_ptr(), rows, row0, wp.data_ptr(), wt.data_ptr()
    _lib.check(_lib.lib().tef_conv_pack_weights(arr, len(jobs), _lib.stream_ptr()), "tef_conv_pack_weights")
    del jobs[:]          # (the caller's list: a second call packs only what was queued since)


class _ConvFn(torch.autograd.Function):
    """act(conv2d(cat[x0, x1 * gate1], cat(weights, 0), cat(biases, 0))) with padding k//2 (include/tef.h
    tef_conv_forward).  `nw` weight tensors are row-concatenated without materialising the concatenation."""

    @staticmethod
    def forward(ctx, packer, stride, act, nw, out_split, x0, x1, gate1, *wb):
        lib = _lib.lib()
        _lib.require_device_tensor(x0, "conv input")
        weights, biases = wb[:nw], wb[nw:]
        x0 = x0.contiguous()
        x1 = x1.contiguous() if x1 is not None else None
        gate1 = gate1.contiguous() if gate1 is not None else None
        B, C0, H, W = x0.shape
        C1 = x1.shape[1] if x1 is not None else 0
        N = sum(w.shape[0] for w in weights)
        Ct, k = weights[0].shape[1], weights[0].shape[2]
        assert Ct == C0 + C1, (Ct, C0, C1)
        d = _lib.ConvDesc(B, C0, C1, H, W, N, k, stride, _lib.ACT[act])
        wp, wt = packer.get(weights, d)
        bias = None
        if biases and biases[0] is not None:
            bias = biases[0] if len(biases) == 1 else torch.cat([b.detach() for b in biases])
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        split = N if out_split is None else int(out_split)
        out = torch.empty((B, split, Ho, Wo), dtype=torch.float32, device=x0.device)
        out2 = torch.empty((B, N - split, Ho, Wo), dtype=torch.float32, device=x0.device) if split < N else None
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x0.device)
        rc = lib.tef_conv_forward_split(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), wp.data_ptr(), _ptr(bias),
                                        out.data_ptr(), _ptr(out2), split, ws.data_ptr(), nbytes, _lib.stream_ptr())
        _lib.check(rc, "tef_conv_forward_split")
        ctx.io_split = split
        ctx.desc, ctx.nw, ctx.wt = d, nw, wt
        ctx.packer, ctx.params = packer, (weights, biases)
        ctx.rows = [w.shape[0] for w in weights]
        ctx.wshape = tuple(weights[0].shape[1:])
        ctx.has = (x1 is not None, gate1 is not None, bias is not None)
        ctx.save_for_backward(x0, x1, gate1, out if act is not None else None,
                              out2 if (act is not None and out2 is not None) else None)
        return out if out2 is None else (out, out2)

    @staticmethod
    def backward(ctx, dout, dout2=None):
        lib = _lib.lib()
        x0, x1, gate1, out, out2 = ctx.saved_tensors
        d, nw = ctx.desc, ctx.nw
        has_x1, has_gate, has_bias = ctx.has
        need = (ctx.needs_input_grad[:4] + ctx.needs_input_grad[5:])   # drop out_split: (packer, stride, act, nw, x0, x1, gate1, *w, *b)
        io_split = ctx.io_split
        if io_split < d.N:        # an unused half of a split output arrives as None
            Ho, Wo = (dout if dout is not None else dout2).shape[2:]
            if dout is None:
                dout = torch.zeros((d.B, io_split, Ho, Wo), dtype=torch.float32, device=dout2.device)
            if dout2 is None:
                dout2 = torch.zeros((d.B, d.N - io_split, Ho, Wo), dtype=torch.float32, device=dout.device)
            dout2 = dout2.contiguous()
        dout = dout.contiguous()
        dev = dout.device
        need_dx = need[4] or (has_x1 and (need[5] or need[6]))      # the kernel produces dx0 and dx1 together
        dx0 = torch.empty_like(x0) if need_dx else None
        dxg = torch.empty_like(x1) if (has_x1 and need_dx) else None
        need_w = any(need[7:7 + nw])
        need_b = has_bias and any(need[7 + nw:])
        nbytes = lib.tef_conv_workspace_bytes(ctypes.byref(d))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        weights, biases = ctx.params
        direct = (ctx.packer.direct_grads and nw <= 2 and need_w and all(need[7:7 + nw])
                  and all(w.grad is not None and w.grad.is_contiguous() for w in weights)
                  and (not has_bias or all(b.grad is not None and b.grad.is_contiguous() for b in biases)))
        packer = ctx.packer
        if direct and packer.defer_wgrad and lib.tef_conv_wgrad_parts_supported(ctypes.byref(d)):
            # input / bias gradients now; g = dY * act'(out) is kept and the weight gradient joins the layer's queue
            dbs = ([b.grad.data_ptr() for b in biases] + [None]) if has_bias else [None, None]
            g_formed = d.act != _lib.ACT[None] or io_split < d.N
            g = torch.empty((d.B, d.N) + tuple(dout.shape[2:]), dtype=torch.float32, device=dev) if g_formed else None
            rc = lib.tef_conv_backward_keep(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), ctx.wt.data_ptr(),
                                            _ptr(out), _ptr(out2), dout.data_ptr(), _ptr(dout2), io_split, _ptr(dx0),
                                            _ptr(dxg), None, None, dbs[0], dbs[1],
                                            ctx.rows[0] if nw == 2 else d.N, _ptr(g), ws.data_ptr(), nbytes,
                                            _lib.stream_ptr())
            _lib.check(rc, "tef_conv_backward_keep")
            meta = (d, [w.grad for w in weights], ctx.rows[0] if nw == 2 else d.N)
            if packer.pending and (packer.pending_meta[0].B, packer.pending_meta[0].H, packer.pending_meta[0].W) != (d.B, d.H, d.W):
                flush_deferred_wgrads(packer)
            packer.pending_meta = meta
            packer.pending.append((g if g_formed else dout, x0, x1, gate1))
            if getattr(packer, "auto_flush", False) and not _DEFERRED:
                # a network that owns its gradient buffers (arch.own_gradients: the drop-in loop, nobody flushes): at the end
                # of this backward()
                torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_wgrads)
            _DEFERRED.add(packer)
            if len(packer.pending) == _MAX_PARTS:
                flush_deferred_wgrads(packer)
            dw = db = None
        elif direct:      # the kernel adds into the parameters' own .grad buffers
            dws = [w.grad.data_ptr() for w in weights] + [None]
            dbs = ([b.grad.data_ptr() for b in biases] + [None]) if has_bias else [None, None]
            rc = lib.tef_conv_backward_split(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), ctx.wt.data_ptr(),
                                             _ptr(out), _ptr(out2), dout.data_ptr(), _ptr(dout2), io_split, _ptr(dx0),
                                             _ptr(dxg), dws[0], dws[1], dbs[0], dbs[1],
                                             ctx.rows[0] if nw == 2 else d.N, ws.data_ptr(), nbytes, _lib.stream_ptr())
            _lib.check(rc, "tef_conv_backward_split")
            dw = db = None
        else:
            dw = torch.zeros((d.N,) + ctx.wshape, dtype=torch.float32, device=dev) if need_w else None
            db = torch.zeros((d.N,), dtype=torch.float32, device=dev) if need_b else None
            rc = lib.tef_conv_backward_split(ctypes.byref(d), x0.data_ptr(), _ptr(x1), _ptr(gate1), ctx.wt.data_ptr(),
                                             _ptr(out), _ptr(out2), dout.data_ptr(), _ptr(dout2), io_split, _ptr(dx0),
                                             _ptr(dxg), _ptr(dw), None, _ptr(db), None, d.N, ws.data_ptr(), nbytes,
                                             _lib.stream_ptr())
            _lib.check(rc, "tef_conv_backward_split")
        if not need[4]:
            dx0 = None
        dx1 = dgate = None
        if dxg is not None:
            if has_gate:
                dx1 = dxg * gate1 if need[5] else None
                dgate = dxg * x1 if need[6] else None
            else:
                dx1 = dxg
        gw, gb, r0 = [], [], 0
        for r in ctx.rows:
            gw.append(dw[r0:r0 + r] if dw is not None else None)
            gb.append(db[r0:r0 + r] if db is not None else None)
            r0 += r
        if not has_bias:
            gb = [None] * (len(need) - 7 - nw)
        return (None, None, None, None, None, dx0, dx1, dgate) + tuple(gw) + tuple(gb)


_MAX_PARTS = 16        # TEF_CONV_MAX_PARTS (include/tef.h)
_DEFERRED = set()      # packers with queued weight-gradient parts
_DEFERRED_ENGINES = set()      # pass engines (models/engine.py) holding the backward calls of an unfinished window


def flush_deferred_wgrads(packer=None):
    """Run the queued weight gradients (all layers, or one): one tef_conv_wgrad_parts launch per layer over every backward
    call since the last flush.  Call after loss.backward() and before the gradients are read (all-reduce, clip, step)."""
    if packer is None:
        for eng in list(_DEFERRED_ENGINES):      # the fused passes of the window: one tef_net_window_wgrads per engine
            eng.flush_window()
    todo = [packer] if packer is not None else list(_DEFERRED)
    lib = _lib.lib()
    for pk in todo:
        parts = pk.pending
        if not parts:
            _DEFERRED.discard(pk)
            continue
        d, grads, split_rows = pk.pending_meta
        n = len(parts)
        arr = lambda col: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in col])  # noqa: E731
        gs, x0s, x1s, gts = (arr([p[i] for p in parts]) for i in range(4))
        rc = lib.tef_conv_wgrad_parts(ctypes.byref(d), n, gs, x0s, x1s, gts, grads[0].data_ptr(),
                                      grads[1].data_ptr() if len(grads) > 1 else None, split_rows, _lib.stream_ptr())
        _lib.check(rc, "tef_conv_wgrad_parts")
        pk.pending = []
        _DEFERRED.discard(pk)


def invalidate_packed(module):
    """Forget the packed GEMM operands of every convolution of `module`: for optimisers that update the parameters with
    their own kernels (parallel.FusedAdam), which does not bump the version counters the packed copies are keyed on."""
    for m in module.modules():
        for v in vars(m).values():
            if isinstance(v, PackedWeights):
                v.invalidate()


def enable_deferred_wgrad(module, on=True):
    """With enable_direct_grads: queue the weight gradients of the convolutions of `module` during backward and compute
    them per layer in flush_deferred_wgrads() (the caller must call it after every backward)."""
    n = 0
    for m in module.modules():
        for v in vars(m).values():
            if isinstance(v, PackedWeights):
                v.defer_wgrad = on
                n += 1
    return n


def enable_direct_grads(module, on=True):
    """Let every convolution of `module` accumulate its parameter gradients straight into the pre-allocated .grad
    buffers (the caller owns them and runs loss.backward(); not for torch.autograd.grad on the parameters)."""
    n = 0
    for m in module.modules():
        for v in vars(m).values():
            if isinstance(v, PackedWeights):
                v.direct_grads = on
                n += 1
    return n


def conv2d(packer, x0, weights, biases, stride=1, act=None, x1=None, gate1=None, out_split=None):
    """weights / biases: a parameter or a tuple of parameters to be row-concatenated (same input, one GEMM).
    out_split: return the output channels as two tensors, [:out_split] and [out_split:]."""
    if not isinstance(weights, (tuple, list)):
        weights, biases = (weights,), (biases,)
    return _ConvFn.apply(packer, stride, act, len(weights), out_split, x0, x1, gate1, *weights, *biases)


class _UpsampleFn(torch.autograd.Function):
    """mul * F.interpolate(x, scale_factor=(sh, sw), mode="bilinear", align_corners=False) for integer factors."""

    @staticmethod
    def forward(ctx, x, sh, sw, mul):
        _lib.require_device_tensor(x, "upsample input")
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty((B, C, H * sh, W * sw), dtype=torch.float32, device=x.device)
        rc = _lib.lib().tef_upsample_bilinear(x.data_ptr(), B * C, H, W, sh, sw, mul, y.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_upsample_bilinear")
        ctx.geom = (B, C, H, W, sh, sw, mul)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, sh, sw, mul = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        rc = _lib.lib().tef_upsample_bilinear_backward(dy.data_ptr(), B * C, H, W, sh, sw, mul, dx.data_ptr(),
                                                       _lib.stream_ptr())
        _lib.check(rc, "tef_upsample_bilinear_backward")
        return dx, None, None, None


def upsample_bilinear(x, scale_h, scale_w, mul=1.0):
    """Integer-factor bilinear up-sampling (align_corners=False) on the HIP kernel; other factors are not used by
    RecEVFlowNet (power-of-two pyramid on a padded input)."""
    sh, sw = int(round(scale_h)), int(round(scale_w))
    if sh != scale_h or sw != scale_w or sh < 1 or sw < 1:
        raise NotImplementedError(f"non-integer bilinear scale ({scale_h}, {scale_w})")
    if sh == 1 and sw == 1:
        return x * mul if mul != 1.0 else x
    return _UpsampleFn.apply(x, sh, sw, float(mul))


class _GruBlendFn(torch.autograd.Function):
    """new_state = prev_state * (1 - update) + out_inputs * update (reference submodules.py:150)."""

    @staticmethod
    def forward(ctx, h, u, o):
        h, u, o = h.contiguous(), u.contiguous(), o.contiguous()
        out = torch.empty_like(h)
        rc = _lib.lib().tef_gru_blend(h.data_ptr(), u.data_ptr(), o.data_ptr(), h.numel(), out.data_ptr(),
                                      _lib.stream_ptr())
        _lib.check(rc, "tef_gru_blend")
        ctx.save_for_backward(h, u, o)
        return out

    @staticmethod
    def backward(ctx, dhn):
        h, u, o = ctx.saved_tensors
        dhn = dhn.contiguous()
        dh, du, do = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        rc = _lib.lib().tef_gru_blend_backward(dhn.data_ptr(), h.data_ptr(), u.data_ptr(), o.data_ptr(), h.numel(),
                                               dh.data_ptr(), du.data_ptr(), do.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_gru_blend_backward")
        return dh, du, do


def _act_name(activation):
    if activation is None:
        return None
    if activation not in ("relu", "tanh", "sigmoid"):
        raise NotImplementedError(f"activation '{activation}' has no fused HIP epilogue (relu/tanh/sigmoid/None)")
    return activation


def _no_norm(norm):
    if norm is not None:
        raise NotImplementedError("norm layers are unused by RecEVFlowNet (norm=None, reference models/model.py:28)")


class ConvLayer(nn.Module):
    """Convolutional layer. Default: bias, ReLU, no downsampling, no batch norm (reference submodules.py:8-62)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None,
                 BN_momentum=0.1, w_scale=None, padding=None, bias=None):
        super().__init__()
        _no_norm(norm)
        if padding is not None and padding != kernel_size // 2:
            raise NotImplementedError("only padding = kernel_size // 2 is implemented")
        if bias is None:
            bias = True
        if w_scale is None:
            w_scale = math.sqrt(1 / in_channels)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=bias)
        nn.init.uniform_(self.conv2d.weight, -w_scale, w_scale)
        if bias:
            nn.init.zeros_(self.conv2d.bias)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm
        self._packed = PackedWeights()

    def forward(self, x):
        return conv2d(self._packed, x, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation)


class ConvGRU(nn.Module):
    """Convolutional GRU cell (reference submodules.py:111-152)."""

    def __init__(self, input_size, hidden_size, kernel_size, activation=None):
        super().__init__()
        padding = kernel_size // 2
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        assert activation is None, "ConvGRU activation cannot be set (just for compatibility)"
        nn.init.orthogonal_(self.reset_gate.weight)
        nn.init.orthogonal_(self.update_gate.weight)
        nn.init.orthogonal_(self.out_gate.weight)
        nn.init.constant_(self.reset_gate.bias, 0.0)
        nn.init.constant_(self.update_gate.bias, 0.0)
        nn.init.constant_(self.out_gate.bias, 0.0)
        self._packed_ur, self._packed_o = PackedWeights(), PackedWeights()

    def forward(self, input_, prev_state):
        if prev_state is None:
            prev_state = torch.zeros((input_.shape[0], self.hidden_size) + tuple(input_.shape[2:]),
                                     dtype=input_.dtype, device=input_.device)
        C = self.hidden_size
        # update and reset gates share their input: one GEMM with 2C output channels
        update, reset = conv2d(self._packed_ur, input_, (self.update_gate.weight, self.reset_gate.weight),
                               (self.update_gate.bias, self.reset_gate.bias), 1, "sigmoid", x1=prev_state, out_split=C)
        # tanh(out_gate(cat[input_, prev_state * reset])): the product is formed inside the im2col gather
        out_inputs = conv2d(self._packed_o, input_, self.out_gate.weight, self.out_gate.bias, 1, "tanh",
                            x1=prev_state, gate1=reset)
        new_state = _GruBlendFn.apply(prev_state, update, out_inputs)
        return new_state, new_state


class RecurrentConvLayer(nn.Module):
    """Convolution followed by a recurrent convolutional block (reference submodules.py:65-108)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, recurrent_block_type="convgru",
                 activation_ff="relu", activation_rec=None, norm=None, BN_momentum=0.1):
        super().__init__()
        assert recurrent_block_type in ["convgru"]
        self.recurrent_block_type = recurrent_block_type
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, activation_ff, norm,
                              BN_momentum=BN_momentum)
        self.recurrent_block = ConvGRU(input_size=out_channels, hidden_size=out_channels, kernel_size=3,
                                       activation=activation_rec)

    def forward(self, x, prev_state):
        x = self.conv(x)
        x, state = self.recurrent_block(x, prev_state)
        return x, state


class ResidualBlock(nn.Module):
    """Residual block (reference submodules.py:155-227): conv-act-conv, += x, act; returns (out2, out1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation="relu", downsample=None,
                 norm=None, BN_momentum=0.1):
        super().__init__()
        _no_norm(norm)
        if downsample is not None:
            raise NotImplementedError("downsample is unused by RecEVFlowNet")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=kernel_size // 2, bias=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=kernel_size // 2, bias=True)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm
        self.downsample = downsample
        self._packed1, self._packed2 = PackedWeights(), PackedWeights()

    def forward(self, x):
        out1 = conv2d(self._packed1, x, self.conv1.weight, self.conv1.bias, self.stride, self.activation)
        out2 = conv2d(self._packed2, out1, self.conv2.weight, self.conv2.bias, self.stride, None)
        out2 = out2 + x
        if self.activation is not None:
            out2 = getattr(torch, self.activation)(out2)
        return out2, out1


class UpsampleConvLayer(nn.Module):
    """Bilinear x2 (align_corners=False) + Conv2d + activation (reference submodules.py:230-273)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu", norm=None):
        super().__init__()
        _no_norm(norm)
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2, bias=True)
        self.activation = _act_name(activation)
        self.stride = stride
        self.norm = norm
        self._packed = PackedWeights()

    def forward(self, x):
        return self.conv_after_upsample(upsample_bilinear(x, 2, 2))

    def conv_after_upsample(self, up, up_first=None):
        """The convolution on already up-sampled features; `up_first` supplies the leading input channels from a second
        tensor (the previous prediction, reference arch.py:238) without concatenating."""
        if up_first is None:
            return conv2d(self._packed, up, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation)
        return conv2d(self._packed, up_first, self.conv2d.weight, self.conv2d.bias, self.stride, self.activation, x1=up)


class TransposedConvLayer(nn.Module):
    """Present in the reference (submodules.py:276-325) but never built by RecEVFlowNet (use_upsample_conv=True)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("TransposedConvLayer is unused by RecEVFlowNet (use_upsample_conv=True)")
