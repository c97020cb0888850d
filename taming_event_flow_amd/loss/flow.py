"""Contrast-maximisation loss modules — drop-in for the reference's ``loss/flow.py``.

Same public surface as the reference (`BaseEventWarping` loss/flow.py:14, `Linear` :216, `Iterative` :415):
``L = Iterative(config, device)``, ``L.update(flow_list, event_list, pol_mask, d_event_list, d_pol_mask)``,
``L.num_passes``, ``loss = L()``, ``L.reset()``; the same config keys are read
(``loader.resolution|batch_size``, ``loss.flow_spat_smooth_weight|flow_temp_smooth_weight|round_ts|iterative_mode``,
``data.passes_loss|scales_loss``) and ``update`` keeps the reference's in-place time shift of the caller's event
lists (loss/flow.py:457-458).

All arithmetic runs in the hand-written HIP kernels of libtef_hip.so (include/tef.h); this file only owns
buffers, bookkeeping and the autograd boundary.  There is no PyTorch/CPU fallback.
"""

import ctypes
import weakref

import torch

try:
    from .. import _lib
    from ..models.lazy import LazyFlow as _LazyFlow, plain_of as _plain_of
except ImportError:      # drop-in mode: this package's directory itself is on sys.path
    import _lib
    from models.lazy import LazyFlow as _LazyFlow, plain_of as _plain_of

__all__ = ["BaseEventWarping", "Linear", "Iterative"]


class _SoA:
    """Growable structure-of-arrays event store for one list (grad or detached) of one window."""

    def __init__(self, B, device, res, cap_hint=0):
        self.B, self.device, self.res = B, device, res
        self.cap = 0
        self.n = 0
        self.cap_hint = cap_hint      # slots the previous window of this module needed: allocated in one go
        self.off = [0]
        self.ts = self.y = self.x = self.mp = self.mn = self.bin = None
        self._ref = None
        # per (sample, pass): ends of the pos-only / neg-only / both-polarity runs (written by tef_pack_events)
        self.cls = torch.zeros((B, _lib.TEF_MAX_PASSES, 3), dtype=torch.int32, device=device)

    def _grow(self, need):
        cap = max(need, 2 * self.cap, 1024, self.cap_hint)
        new = [torch.empty((self.B, cap), dtype=torch.float32, device=self.device) for _ in range(5)]
        nbin = torch.empty((cap,), dtype=torch.uint8, device=self.device)
        if self.n:
            for dst, src in zip(new, (self.ts, self.y, self.x, self.mp, self.mn)):
                dst[:, : self.n].copy_(src[:, : self.n])
            nbin[: self.n].copy_(self.bin[: self.n])
        self.ts, self.y, self.x, self.mp, self.mn = new
        self.bin = nbin
        self.cap = cap
        self._ref = None

    def append(self, ev, pm, pass_idx, ts_override):
        """Pack one pass ([B,N,4], [B,N,2]); shifts ev[:, :, 0] in place by pass_idx."""
        B, N = ev.shape[0], ev.shape[1]
        if B != self.B:
            raise RuntimeError(f"event list batch {B} != config loader.batch_size {self.B}")
        Np = (N + 63) & ~63                 # passes start at multiples of 64 slots (include/tef.h tef_pack_events)
        if self.n + Np > self.cap:
            self._grow(self.n + Np)
        if N:
            in_place = ev.is_contiguous() and ev.dtype == torch.float32 and ev.data_ptr() % 16 == 0
            if in_place:
                src, shift = ev, float(pass_idx)
            else:  # exotic views: keep the side effect with a strided add, pack a contiguous copy
                ev[:, :, 0:1] += pass_idx
                src, shift = ev.to(torch.float32).contiguous(), 0.0
            pmc = pm.to(torch.float32).contiguous()
            rc = _lib.lib().tef_pack_events(
                src.data_ptr(), pmc.data_ptr(), B, N, shift, None if ts_override is None else ts_override.data_ptr(),
                pass_idx, self.n, self.cap,
                self.res[0], self.res[1],
                self.ts.data_ptr(), self.y.data_ptr(), self.x.data_ptr(), self.mp.data_ptr(), self.mn.data_ptr(),
                self.bin.data_ptr(), self.cls.data_ptr(), _lib.stream_ptr(),
            )
            _lib.check(rc, "tef_pack_events")
        self.n += Np
        self.off.append(self.n)

    def reserve(self, N):
        """Room for a pass of N events (padded to a multiple of 64 slots); -> its first slot."""
        Np = (N + 63) & ~63
        if self.n + Np > self.cap:
            self._grow(self.n + Np)
        return self.n

    def commit(self, N):
        self.n += (N + 63) & ~63
        self.off.append(self.n)

    def struct(self):
        if self.cap == 0:
            self._grow(1)
        return _lib.Events(self.ts.data_ptr(), self.y.data_ptr(), self.x.data_ptr(), self.mp.data_ptr(),
                           self.mn.data_ptr(), self.bin.data_ptr(), self.cls.data_ptr(), self.cap)

    def struct_ref(self):
        """byref() of a struct kept until the buffers change (update() hands it to the library once per pass)."""
        if self._ref is None or self._ref[0] != self.cap or self.cap == 0:
            st = self.struct()
            self._ref = (self.cap, st, ctypes.byref(st))
        return self._ref[2]


class _Window:
    """Device state of one loss window (everything the kernels read)."""

    def __init__(self, B, device, res, hints=(0, 0)):
        self.flows = None          # planar      [P][F][B][2][H][W]  (smoothing terms, gradient layout)
        self.flows_yx = None       # interleaved [P][F][B][H][W][2]  (flow_y, flow_x) for the lookups
        self.flow_refs = []        # autograd handles, flow_refs[t][i]
        self.grad = _SoA(B, device, res, hints[0])
        self.det = _SoA(B, device, res, hints[1])
        self.workspace = None
        self.scratch = None
        self.pass_args = None      # ctypes argument arrays of tef_update_pass, refilled per pass
        self.pending = []          # passes recorded by a deferred update() (BaseEventWarping.defer_update)
        self.pending_ptrs = set()  # storage of the event lists recorded so far (aliasing check of the deferred update)
        self.leases = []          # weak references to the tokens of evaluations whose autograd graph still reads the buffers
        self.cfg = None


class _Token:
    """Lives exactly as long as the autograd context of one loss evaluation (weakly referenced by the window)."""

    __slots__ = ("__weakref__",)


class _CMLossFn(torch.autograd.Function):
    """Autograd boundary: inputs are the F*P flow tensors of the window, output the scalar loss."""

    @staticmethod
    def forward(ctx, module, win, *flows):
        lib = _lib.lib()
        cfg = win.cfg
        loss = torch.empty((), dtype=torch.float32, device=win.flows.device)      # (written, not accumulated: no fill launch)
        g, d = win.grad.struct(), win.det.struct()
        rc = lib.tef_loss_forward(ctypes.byref(cfg), win.flows_yx.data_ptr(), ctypes.byref(g), ctypes.byref(d),
                                  win.workspace.data_ptr(), win.workspace.numel(), loss.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "tef_loss_forward")
        ws, wt = module._smooth_weights(cfg.P)
        if ws >= 0 or wt >= 0:
            rc = lib.tef_smoothing_forward(ctypes.byref(cfg), win.flows.data_ptr(), ws, wt, win.scratch.data_ptr(),
                                           loss.data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "tef_smoothing_forward")
        # the context owns everything backward reads, so a later forward on the same window cannot clobber it: while this
        # evaluation's graph is alive its token is, and forward() takes fresh buffers instead of these (explicit ownership)
        ctx.win, ctx.cfg, ctx.workspace, ctx.scratch = win, cfg, win.workspace, win.scratch
        ctx.token = _Token()
        win.leases.append(weakref.ref(ctx.token))
        ctx.smooth = (ws, wt)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.lib()
        win, cfg, workspace, scratch = ctx.win, ctx.cfg, ctx.workspace, ctx.scratch
        go = grad_out.to(torch.float32).contiguous()
        dflows = torch.empty_like(win.flows)
        g, d = win.grad.struct(), win.det.struct()
        rc = lib.tef_loss_backward(ctypes.byref(cfg), win.flows_yx.data_ptr(), ctypes.byref(g), ctypes.byref(d),
                                   workspace.data_ptr(), workspace.numel(), go.data_ptr(), dflows.data_ptr(),
                                   _lib.stream_ptr())
        _lib.check(rc, "tef_loss_backward")
        ws, wt = ctx.smooth
        if ws >= 0 or wt >= 0:
            rc = lib.tef_smoothing_backward(ctypes.byref(cfg), win.flows.data_ptr(), ws, wt, scratch.data_ptr(),
                                            go.data_ptr(), dflows.data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "tef_smoothing_backward")
        P, F = cfg.P, cfg.F
        grads = tuple(dflows[t, i] for t in range(P) for i in range(F))
        return (None, None) + grads


class BaseEventWarping(torch.nn.Module):
    """Base class for the contrast maximization loss (reference loss/flow.py:14-213)."""

    _kind = None

    def __init__(self, config, device, loss_scaling=True, border_compensation=True):
        super().__init__()
        self.device = torch.device(device)
        self.config = config
        self.loss_scaling = loss_scaling
        self.border_compensation = border_compensation
        self.res = config["loader"]["resolution"]
        self.batch_size = config["loader"]["batch_size"]
        self.flow_spat_smooth_weight = config["loss"]["flow_spat_smooth_weight"]
        self.flow_temp_smooth_weight = config["loss"]["flow_temp_smooth_weight"]

        self._passes = 0
        self._num_flows = None
        self._win = None
        self._side = None       # the stream update() ran on when it was handed flows that live on the network's side stream
        # Not in the reference: update() only records the pass and the whole window is packed in ONE launch when the loss
        # is evaluated (tef_update_window) — for callers to whom ten launches and ten host calls per window matter (a
        # loss-only caller; inside a training window update() hides behind the network on a side stream anyway).  The
        # in-place shift of the callers' time stamps (loss/flow.py:457-458) then happens at the evaluation, and the lists
        # must not be changed between update() and the evaluation (a list whose storage was already recorded in this window
        # — one buffer reused for every pass — raises; a pass that takes the converting path, and reset(), pack the recorded
        # passes first).  Off by default: reference behaviour.
        self.defer_update = bool(config["loss"].get("defer_update", False))

        # timescales for loss computation (loss/flow.py:42-44)
        self.passes_loss = []
        for s in range(config["data"]["scales_loss"]):
            self.passes_loss.append(config["data"]["passes_loss"] // (2**s))

    # ---- reference API -------------------------------------------------------------------------
    @property
    def num_passes(self):
        return self._passes

    def update_base(self, flow_list):
        """Append the flow maps of one pass (reference loss/flow.py:46-66)."""
        if self._num_flows is None:
            self._num_flows = len(flow_list)
        if len(flow_list) != self._num_flows:
            raise RuntimeError("number of flow maps changed inside a loss window")
        P = max(self.passes_loss)
        if self._passes >= P:
            raise RuntimeError(f"update() called more than data.passes_loss={P} times without reset()")
        H, W = self.res
        B, F = self.batch_size, self._num_flows
        win = self._win
        if win is None:
            win = self._win = _Window(B, self.device, self.res, getattr(self, "_hints", (0, 0)))
        if win.flows is None:
            _lib.require_device_tensor(flow_list[0], "flow map")
            win.flows = torch.empty((P, F, B, 2, H, W), dtype=torch.float32, device=flow_list[0].device)
            win.flows_yx = torch.empty((P, F, B, H, W, 2), dtype=torch.float32, device=flow_list[0].device)
        refs, srcs = [], []
        for i, flow in enumerate(flow_list):
            if tuple(flow.shape) != (B, 2, H, W):
                raise RuntimeError(f"flow map {i} has shape {tuple(flow.shape)}, expected {(B, 2, H, W)}")
            src = _lib.require_device_tensor(flow.detach(), "flow map")
            if src.dtype != torch.float32 or src.stride(3) != 1 or src.stride(2) != W:
                src = src.to(torch.float32).contiguous()
            srcs.append(src)
            refs.append(flow)
        if F <= 16:       # all heads of the pass in one launch
            ptrs = (ctypes.c_void_p * F)(*[t.data_ptr() for t in srcs])
            sb = (ctypes.c_long * F)(*[t.stride(0) for t in srcs])
            sc = (ctypes.c_long * F)(*[t.stride(1) for t in srcs])
            rc = _lib.lib().tef_pack_flows(ptrs, sb, sc, F, B, H, W, win.flows[self._passes].data_ptr(),
                                           win.flows_yx[self._passes].data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "tef_pack_flows")
        else:
            for i, src in enumerate(srcs):
                rc = _lib.lib().tef_pack_flow(src.data_ptr(), src.stride(0), src.stride(1), B, H, W,
                                              win.flows[self._passes, i].data_ptr(), win.flows_yx[self._passes, i].data_ptr(),
                                              _lib.stream_ptr())
                _lib.check(rc, "tef_pack_flow")
        win.flow_refs.append(refs)

    def reset_base(self):
        if self._win is not None and self._win.pending:
            # a window dropped with recorded passes (a new sequence in the middle of it): the reference has shifted the
            # callers' time stamps in place at every update() (loss/flow.py:457-458) — pack them now, which applies the shift
            self._flush_updates()
        if self._win is not None:      # the next window starts with stores of the size this one ended with
            self._hints = (max(self._win.grad.n, getattr(self, "_hints", (0, 0))[0]), max(self._win.det.n, getattr(self, "_hints", (0, 0))[1]))
        self._passes = 0
        self._win = None
        self._side = None

    def _update_events(self, event_list, pol_mask, d_event_list, d_pol_mask):
        """Shared tail of Linear.update / Iterative.update (loss/flow.py:246-263, :456-476)."""
        win = self._win
        for name, t in (("event_list", event_list), ("pol_mask", pol_mask), ("d_event_list", d_event_list),
                        ("d_pol_mask", d_pol_mask)):
            _lib.require_device_tensor(t, name)
        ovr = d_ovr = None
        if self.config["loss"]["round_ts"]:
            # event_ts[...] = event_ts.min() + 0.5 over the shifted list (loss/flow.py:461-463), in fp32 on the device like
            # the reference: min(ts + passes) = fl(min(ts) + passes) (rounding is monotone), then + 0.5.  No host sync.
            def rounded(lst):
                return ((lst[:, :, 0].min().to(torch.float32) + float(self._passes)) + 0.5).reshape(1).contiguous()

            if event_list.shape[1] > 0:
                ovr = rounded(event_list)
            if d_event_list.shape[1] > 0:
                d_ovr = rounded(d_event_list)
            win.keep = getattr(win, "keep", []) + [ovr, d_ovr]       # alive until the pack kernels have run
        win.grad.append(event_list, pol_mask, self._passes, ovr)
        win.det.append(d_event_list, d_pol_mask, self._passes, d_ovr)
        self._passes += 1

    def _update_pass(self, flow_list, event_list, pol_mask, d_event_list, d_pol_mask):
        """update() of one pass.  Flow maps that are still on the network's side stream (models/lazy.py: the literal
        train_flow.py loop on this package's model) are packed THERE, beside the next pass's encoders; the evaluation of the
        loss waits for that stream."""
        side = None
        if any(type(f) is _LazyFlow for f in flow_list):
            plains = []
            for f in flow_list:
                pl, st = _plain_of(f)
                plains.append(pl)
                side = side or st
            flow_list = plains
        if side is None:
            return self._update_pass_here(flow_list, event_list, pol_mask, d_event_list, d_pol_mask)
        main = torch.cuda.current_stream()
        side.wait_stream(main)               # (the loader's tensors of this pass are produced on the caller's stream)
        for t_ in (event_list, pol_mask, d_event_list, d_pol_mask):
            if isinstance(t_, torch.Tensor) and t_.is_cuda:
                t_.record_stream(side)       # (the caller may drop them before the side stream has read them)
        self._side = side
        with torch.cuda.stream(side):
            return self._update_pass_here(flow_list, event_list, pol_mask, d_event_list, d_pol_mask)

    def _update_pass_here(self, flow_list, event_list, pol_mask, d_event_list, d_pol_mask):
        """One pass of update(): flow maps + both event lists handed to the library in ONE call (tef_update_pass) when the
        caller's tensors are in the layout the reference's loader produces (contiguous fp32); anything else takes the
        call-by-call path, which converts."""
        F = len(flow_list)
        H, W = self.res

        # (written for few attribute look-ups: this check is a third of the host time of a call)
        B, f32 = self.batch_size, torch.float32
        es, ds = event_list.shape, d_event_list.shape
        fast = F <= 16 and len(es) == 3 and len(ds) == 3
        if fast:
            N, Nd = es[1], ds[1]
            fshape, ftail = (B, 2, H, W), (W, 1)
            fast = (es == (B, N, 4) and pol_mask.shape == (B, N, 2) and ds == (B, Nd, 4) and d_pol_mask.shape == (B, Nd, 2)
                    and event_list.dtype is f32 and pol_mask.dtype is f32 and d_event_list.dtype is f32 and d_pol_mask.dtype is f32
                    and event_list.is_cuda and pol_mask.is_cuda and d_event_list.is_cuda and d_pol_mask.is_cuda
                    and event_list.is_contiguous() and pol_mask.is_contiguous() and d_event_list.is_contiguous()
                    and d_pol_mask.is_contiguous()
                    and not ((event_list.data_ptr() | pol_mask.data_ptr() | d_event_list.data_ptr() | d_pol_mask.data_ptr()) & 15)
                    and all(f.shape == fshape and f.dtype is f32 and f.is_cuda and f.stride()[2:] == ftail for f in flow_list))
        if not fast or (self._num_flows is not None and F != self._num_flows) or self._passes >= max(self.passes_loss):
            self._flush_updates()                # (recorded passes first: the stores are filled in pass order)
            self.update_base(flow_list)          # (also where malformed calls get their error messages)
            self._update_events(event_list, pol_mask, d_event_list, d_pol_mask)
            return
        if self._num_flows is None:
            self._num_flows = F
        P, B = max(self.passes_loss), self.batch_size
        win = self._win
        if win is None:
            win = self._win = _Window(B, self.device, self.res, getattr(self, "_hints", (0, 0)))
        if win.flows is None:
            win.flows = torch.empty((P, F, B, 2, H, W), dtype=torch.float32, device=flow_list[0].device)
            win.flows_yx = torch.empty((P, F, B, H, W, 2), dtype=torch.float32, device=flow_list[0].device)
        t = self._passes
        N, Nd = event_list.shape[1], d_event_list.shape[1]
        ovr = d_ovr = None
        if self.config["loss"]["round_ts"]:
            if N > 0:
                ovr = ((event_list[:, :, 0].min() + float(t)) + 0.5).reshape(1).contiguous()
            if Nd > 0:
                d_ovr = ((d_event_list[:, :, 0].min() + float(t)) + 0.5).reshape(1).contiguous()
            win.keep = getattr(win, "keep", []) + [ovr, d_ovr]       # alive until the pack kernels have run
        slot0, dslot0 = win.grad.reserve(N), win.det.reserve(Nd)
        if self.defer_update and win.pending is not None:
            # the pass is only recorded (its tensors stay referenced, nothing is launched, the caller's time stamps are not
            # shifted yet): _flush_updates() hands the whole window to the library in one call when the loss is evaluated.
            # A caller that REUSES one device buffer for the event lists of successive passes (a common loader pattern) would
            # have every recorded pass packed from the buffer's last contents — and by the time the second pass names the
            # buffer the first one's data is gone: nothing to fall back to.  Detected and refused.
            ptrs_ = (event_list.data_ptr() if N else 0, d_event_list.data_ptr() if Nd else 0)
            seen = win.pending_ptrs
            if (ptrs_[0] and ptrs_[0] in seen) or (ptrs_[1] and ptrs_[1] in seen):
                raise RuntimeError("defer_update: this pass's event list lives in storage that an earlier pass of the window "
                                   "recorded and that has not been packed yet (one buffer reused for every pass?).  A "
                                   "deferred update() needs the lists of all passes alive and unchanged until the loss is "
                                   "evaluated: hand in a tensor per pass, or switch defer_update off")
            seen.update(p_ for p_ in ptrs_ if p_)
        if self.defer_update and win.pending is not None:
            win.pending.append((list(flow_list), event_list, pol_mask, d_event_list, d_pol_mask, N, Nd, ovr, d_ovr, t, slot0, dslot0))
            win.grad.commit(N)
            win.det.commit(Nd)
            win.flow_refs.append(list(flow_list))
            self._passes += 1
            return
        # host time per call matters to a loss-only caller (ten calls per window, the kernel takes ~15 us): the argument
        # arrays and the two event-store structs are kept per window and refilled, no view tensors are made
        args = win.pass_args
        if args is None or len(args[0]) != F:
            args = win.pass_args = ((ctypes.c_void_p * F)(), (ctypes.c_long * F)(), (ctypes.c_long * F)())
        ptrs, sb, sc = args
        for i, f in enumerate(flow_list):
            ptrs[i] = f.data_ptr()
            sb[i] = f.stride(0)
            sc[i] = f.stride(1)
        per_pass = F * B * 2 * H * W * 4
        rc = _lib.lib().tef_update_pass(ptrs, sb, sc, F, B, H, W, win.flows.data_ptr() + t * per_pass,
                                        win.flows_yx.data_ptr() + t * per_pass,
                                        event_list.data_ptr(), pol_mask.data_ptr(), N, None if ovr is None else ovr.data_ptr(),
                                        d_event_list.data_ptr(), d_pol_mask.data_ptr(), Nd,
                                        None if d_ovr is None else d_ovr.data_ptr(), t, slot0, dslot0, win.grad.struct_ref(),
                                        win.det.struct_ref(), _lib.stream_ptr())
        _lib.check(rc, "tef_update_pass")
        win.grad.commit(N)
        win.det.commit(Nd)
        win.flow_refs.append(list(flow_list))
        self._passes += 1

    def _flush_updates(self):
        """Deferred update() calls (`defer_update`): all recorded passes of the window in one tef_update_window call."""
        win = self._win
        if win is None or not win.pending:
            return
        if self._side is not None:           # (recorded passes whose flows live on the network's side stream)
            torch.cuda.current_stream().wait_stream(self._side)
        F, B = self._num_flows, self.batch_size
        H, W = self.res
        n = len(win.pending)
        descs = (_lib.UpdateDesc * n)()
        keep = []
        for k, (flows, ev, pm, dev, dpm, N, Nd, ovr, d_ovr, t, slot0, dslot0) in enumerate(win.pending):
            ptrs = (ctypes.c_void_p * F)(*[f.data_ptr() for f in flows])
            sb = (ctypes.c_long * F)(*[f.stride(0) for f in flows])
            sc = (ctypes.c_long * F)(*[f.stride(1) for f in flows])
            keep += [ptrs, sb, sc]
            d = descs[k]
            d.flows = ctypes.cast(ptrs, ctypes.c_void_p)
            d.stride_b = ctypes.cast(sb, ctypes.c_void_p)
            d.stride_c = ctypes.cast(sc, ctypes.c_void_p)
            d.ev, d.pm = ev.data_ptr(), pm.data_ptr()
            d.ts_override = None if ovr is None else ovr.data_ptr()
            d.dev, d.dpm = dev.data_ptr(), dpm.data_ptr()
            d.dts_override = None if d_ovr is None else d_ovr.data_ptr()
            d.N, d.Nd, d.pass_idx, d.slot0, d.dslot0 = N, Nd, t, slot0, dslot0
        rc = _lib.lib().tef_update_window(descs, n, F, B, H, W, win.flows.data_ptr(), win.flows_yx.data_ptr(),
                                          win.grad.struct_ref(), win.det.struct_ref(), _lib.stream_ptr())
        _lib.check(rc, "tef_update_window")
        win.pending = []
        win.pending_ptrs = set()

    def _smooth_weights(self, P):
        ws = -1.0 if self.flow_spat_smooth_weight is None else float(self.flow_spat_smooth_weight)
        wt = -1.0 if (self.flow_temp_smooth_weight is None or P < 2) else float(self.flow_temp_smooth_weight)
        return ws, wt

    def _make_cfg(self):
        win = self._win
        P = max(self.passes_loss)
        cfg = _lib.LossCfg()
        cfg.kind = self._kind
        cfg.B, cfg.H, cfg.W = self.batch_size, self.res[0], self.res[1]
        cfg.P, cfg.F, cfg.S = P, self._num_flows, len(self.passes_loss)
        cfg.mode_div = getattr(self, "_mode_div", 1)
        cfg.M, cfg.Md = win.grad.n, win.det.n
        cfg.loss_scaling = 1 if self.loss_scaling else 0
        # read when the loss is evaluated, like the reference does (loss/flow.py:324, :671): its Linear / Iterative
        # constructors always leave True, the attribute can be changed afterwards
        cfg.border_compensation = 1 if self.border_compensation else 0
        for t in range(P + 1):
            cfg.off[t] = win.grad.off[t]
            cfg.doff[t] = win.det.off[t]
        return cfg

    def forward(self):
        win = self._win
        P = max(self.passes_loss)
        if win is None or self._passes != P:
            raise RuntimeError(f"loss called after {self._passes} update() calls; data.passes_loss={P} are required")
        lib = _lib.lib()
        if self._side is not None:           # update() ran on the network's side stream (LazyFlow inputs)
            torch.cuda.current_stream().wait_stream(self._side)
        self._flush_updates()
        win.cfg = cfg = self._make_cfg()
        nbytes = lib.tef_loss_workspace_bytes(ctypes.byref(cfg))
        if nbytes == 0:
            _lib.check(-1, "tef_loss_workspace_bytes")
        # the window keeps its workspace (0.7 GB at the BASELINE size) across evaluations; an autograd graph of an earlier
        # evaluation that is still alive holds a reference to the one it ran on, and then a fresh one is taken
        win.leases = [r for r in win.leases if r() is not None]
        leased = bool(win.leases)
        if win.workspace is None or win.workspace.numel() != nbytes or leased:
            win.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=win.flows.device)
        ws, wt = self._smooth_weights(P)
        if ws >= 0 or wt >= 0:
            nscr = lib.tef_smoothing_scratch_bytes(ctypes.byref(cfg))
            if win.scratch is None or win.scratch.numel() != nscr or leased:
                win.scratch = torch.empty((nscr,), dtype=torch.uint8, device=win.flows.device)
        flat = [f for refs in win.flow_refs for f in refs]
        return _CMLossFn.apply(self, win, *flat)


class Linear(BaseEventWarping):
    """Contrast maximization loss from Hagenaars and Paredes-Valles et al. (NeurIPS 2021) — reference loss/flow.py:216."""

    _kind = _lib.KIND_LINEAR

    def __init__(self, config, device, loss_scaling=True):
        super().__init__(config, device, loss_scaling=loss_scaling)

    def update(self, flow_list, event_list, pol_mask, d_event_list, d_pol_mask):
        """reference loss/flow.py:233-288.  The per-event flow lookup of :268-283 happens inside the HIP forward
        (the map of the event's own pass is the "latest" map at update time)."""
        self._update_pass(flow_list, event_list, pol_mask, d_event_list, d_pol_mask)

    def reset(self):
        self.reset_base()


class Iterative(BaseEventWarping):
    """CM loss with iterative event warping, loss at all intermediate times and multiple temporal scales (ICCV 2023)
    — reference loss/flow.py:415."""

    _kind = _lib.KIND_ITERATIVE

    def __init__(self, config, device, loss_scaling=True):
        mode = config["loss"]["iterative_mode"]
        if mode == "four":
            # the reference doubles passes_loss here (loss/flow.py:422-423) and then fails with a TypeError in
            # forward (:674-692, shared mask list holds None); there is no behaviour to reproduce
            raise NotImplementedError("iterative_mode 'four' raises TypeError in the reference implementation itself")
        if mode not in ("one", "two"):
            raise ValueError(f"Unknown iterative_mode: {mode}")
        super().__init__(config, device, loss_scaling=loss_scaling)
        self._mode_div = {"one": 1, "two": 2}[mode]
        self.delta_passes = [p // self._mode_div for p in self.passes_loss]   # loss/flow.py:434-441

    def update(self, flow_list, event_list, pol_mask, d_event_list, d_pol_mask):
        """reference loss/flow.py:443-476"""
        self._update_pass(flow_list, event_list, pol_mask, d_event_list, d_pol_mask)

    def reset(self):
        self.reset_base()
