"""ctypes/numpy front end of the CPU oracle (oracle/tef_oracle.c).

*** TEST INFRASTRUCTURE ONLY *** — imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.  Parity of the oracle is pinned
against golden vectors recorded from the reference itself (tests/golden/make_golden.py).
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "tef_oracle.c")
_LIB = os.path.join(_HERE, "libtef_oracle.so")

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


class _Window(ctypes.Structure):
    _fields_ = [
        ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("P", ctypes.c_int),
        ("F", ctypes.c_int), ("S", ctypes.c_int), ("mode_div", ctypes.c_int),
        ("M", ctypes.c_int), ("Md", ctypes.c_int),
        ("off", _i), ("doff", _i), ("flows", _f),
        ("ts", _f), ("y", _f), ("x", _f), ("mp", _f), ("mn", _f), ("loss_scaling", ctypes.c_int),
        ("border_compensation", ctypes.c_int), ("dmass", _f),
    ]


def build(force=False):
    """Compile the C restatement (gcc only; building the checker is not using it)."""
    import hashlib

    with open(_SRC, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    stamp = _LIB + ".srchash"      # content hash, not mtime: mtimes do not survive the copy to the GPU box
    fresh = os.path.exists(_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest
    if force or not fresh:
        subprocess.check_call(
            ["gcc", "-O2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", "-o", _LIB, _SRC, "-lm"]
        )
        with open(stamp, "w") as f:
            f.write(digest)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("TEF_ORACLE_LIB")      # a sanitizer build of the same source (tools/oracle_asan.sh)
        if not override:
            build()
        _lib = ctypes.CDLL(override or _LIB)
        _lib.tef_oracle_iterative.restype = ctypes.c_float
        _lib.tef_oracle_iterative.argtypes = [ctypes.POINTER(_Window), _f, ctypes.c_float]
        _lib.tef_oracle_linear.restype = ctypes.c_float
        _lib.tef_oracle_linear.argtypes = [ctypes.POINTER(_Window), _f, ctypes.c_float]
        for n in ("tef_oracle_spatial_smoothing", "tef_oracle_temporal_smoothing"):
            fn = getattr(_lib, n)
            fn.restype = ctypes.c_float
            fn.argtypes = [ctypes.POINTER(_Window), ctypes.c_float, _f, ctypes.c_float]
        _lib.tef_oracle_focus_loss.restype = ctypes.c_float
    return _lib


def threads(n=0):
    """Set (n > 0) / query the number of OpenMP threads the oracle uses for the (head, sample) loop."""
    return int(lib().tef_oracle_threads(int(n)))


def _p(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Window:
    """A loss window in the oracle's layout, built from the reference-style per-pass inputs.

    flows[t][i]: [B,2,H,W]; ev[t]: [B,N_t,4] (ts,y,x,p) with ts in [0,1] (NOT yet shifted);
    pm[t]: [B,N_t,2]; dev/dpm likewise.  Mirrors what Iterative/Linear.update do
    (reference loss/flow.py:443-476): ts += pass index, optional round_ts.
    """

    def __init__(self, flows, ev, pm, dev, dpm, S=1, mode="two", round_ts=False, loss_scaling=True,
                 border_compensation=True):
        P = len(flows)
        F = len(flows[0])
        B, _, H, W = flows[0][0].shape
        self.P, self.F, self.B, self.H, self.W, self.S = P, F, B, H, W, S
        self.mode_div = {"one": 1, "two": 2}[mode]
        self.flows = _c(np.stack([np.stack(flows[t]) for t in range(P)]))
        n = [ev[t].shape[1] for t in range(P)]
        nd = [dev[t].shape[1] for t in range(P)]
        self.off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
        self.doff = np.concatenate([[0], np.cumsum(nd)]).astype(np.int32)
        self.M, self.Md = int(self.off[-1]), int(self.doff[-1])

        def cat(lists, col, shift):
            parts = []
            for t in range(P):
                a = lists[t][:, :, col].astype(np.float32)
                if shift:
                    a = a + np.float32(t)
                    if round_ts and a.size:
                        a = np.full_like(a, a.min() + np.float32(0.5))
                parts.append(a)
            return np.concatenate(parts, axis=1) if parts else np.zeros((B, 0), np.float32)

        def both(g, d, col, shift=False):
            return _c(np.concatenate([cat(g, col, shift), cat(d, col, shift)], axis=1))

        self.ts = both(ev, dev, 0, True)
        self.y = both(ev, dev, 1)
        self.x = both(ev, dev, 2)
        self.mp = both(pm, dpm, 0)
        self.mn = both(pm, dpm, 1)
        self._w = _Window(B, H, W, P, F, S, self.mode_div, self.M, self.Md,
                          self.off.ctypes.data_as(_i), self.doff.ctypes.data_as(_i), _p(self.flows),
                          _p(self.ts), _p(self.y), _p(self.x), _p(self.mp), _p(self.mn), 1 if loss_scaling else 0,
                          1 if border_compensation else 0, None)
        self.mass = None

    def _run(self, fn, backward, grad_out):
        d = np.zeros_like(self.flows) if backward else None
        loss = fn(ctypes.byref(self._w), _p(d), ctypes.c_float(grad_out))
        return np.float32(loss), d

    def gradient_mass(self, kind="Iterative"):
        """-> [P,F,B,2,H,W]: per pixel of d loss / d flow, the sum over the events of |contribution| (grad_out = 1; the
        smoothing terms not included).  |gradient| <= mass; where contributions cancel, mass >> |gradient| and two fp32
        evaluations cannot agree to 1e-4 of the pixel's own value: tests/conftest.py::elementwise_excess compares
        gradients element by element against this local scale."""
        self.mass = np.zeros_like(self.flows)
        self._w.dmass = _p(self.mass)
        try:
            (self.iterative if kind == "Iterative" else self.linear)(True, 1.0)
        finally:
            self._w.dmass = None
        return self.mass

    def iterative(self, backward=True, grad_out=1.0):
        """-> (loss, dflows [P,F,B,2,H,W]) of loss/flow.py:588 Iterative.forward (no smoothing terms)."""
        return self._run(lib().tef_oracle_iterative, backward, grad_out)

    def linear(self, backward=True, grad_out=1.0):
        return self._run(lib().tef_oracle_linear, backward, grad_out)

    def smoothing(self, spat=None, temp=None, backward=True, grad_out=1.0):
        """-> (term, dflows) for the optional priors (loss/flow.py:739-744)."""
        d = np.zeros_like(self.flows) if backward else None
        total = np.float32(0)
        if spat is not None:
            total += lib().tef_oracle_spatial_smoothing(ctypes.byref(self._w), ctypes.c_float(spat), _p(d),
                                                        ctypes.c_float(grad_out))
        if temp is not None and self.P > 1:
            total += lib().tef_oracle_temporal_smoothing(ctypes.byref(self._w), ctypes.c_float(temp), _p(d),
                                                         ctypes.c_float(grad_out))
        return np.float32(total), d

    def loss(self, kind="Iterative", spat=None, temp=None, backward=True, grad_out=1.0):
        """Full module output: CM loss + optional priors, and d/dflows."""
        l, d = (self.iterative if kind == "Iterative" else self.linear)(backward, grad_out)
        if spat is not None or temp is not None:
            ls, ds = self.smoothing(spat, temp, backward, grad_out)
            l = np.float32(l + ls)
            if backward:
                d = d + ds
        return l, d


def get_event_flow(fx, fy, loc, gout=None):
    fx, fy, loc = _c(fx), _c(fy), _c(loc)
    B, H, W = fx.shape
    N = loc.shape[1]
    out = np.zeros((B, N, 2), np.float32)
    if gout is None:
        lib().tef_oracle_get_event_flow(_p(fx), _p(fy), B, H, W, _p(loc), N, _p(out), None, None, None, None)
        return out
    gout = _c(gout)
    dfx, dfy, dloc = np.zeros_like(fx), np.zeros_like(fy), np.zeros_like(loc)
    lib().tef_oracle_get_event_flow(_p(fx), _p(fy), B, H, W, _p(loc), N, _p(out), _p(gout), _p(dfx), _p(dfy), _p(dloc))
    return out, dfx, dfy, dloc


def get_interpolation(pos, H, W, r=None):
    pos = _c(pos)
    B, N, _ = pos.shape
    idx = np.zeros((B, 4 * N, 1), np.float32)
    w = np.zeros((B, 4 * N, 1), np.float32)
    if r is None:
        lib().tef_oracle_get_interpolation(_p(pos), B, N, H, W, _p(idx), _p(w), None, None)
        return idx, w
    r = _c(r)
    dpos = np.zeros_like(pos)
    lib().tef_oracle_get_interpolation(_p(pos), B, N, H, W, _p(idx), _p(w), _p(r), _p(dpos))
    return idx, w, dpos


def iwe_formatting(pos, mask, ts, H, W, tref, scale):
    pos, mask, ts = _c(pos), _c(mask), _c(ts)
    B, N, _ = pos.shape
    iwe = np.zeros((B, 2, H, W), np.float32)
    iwe_ts = np.zeros((B, 2, H, W), np.float32)
    lib().tef_oracle_iwe_formatting(_p(pos), _p(mask), _p(ts), B, N, H, W, ctypes.c_float(tref), ctypes.c_float(scale),
                                    _p(iwe), _p(iwe_ts))
    return iwe, iwe_ts


def focus_loss(iwe, iwe_ts):
    iwe, iwe_ts = _c(iwe), _c(iwe_ts)
    B, _, H, W = iwe.shape
    return np.float32(lib().tef_oracle_focus_loss(_p(iwe), _p(iwe_ts), B, H, W))


def events_to_channels(xs, ys, ps, H, W):
    out = np.zeros((2, H, W), np.float32)
    lib().tef_oracle_events_to_channels(_p(_c(xs)), _p(_c(ys)), _p(_c(ps)), len(xs), H, W, _p(out))
    return out


def events_to_voxel(xs, ys, ts, ps, bins, H, W):
    out = np.zeros((bins, H, W), np.float32)
    lib().tef_oracle_events_to_voxel(_p(_c(xs)), _p(_c(ys)), _p(_c(ts)), _p(_c(ps)), len(xs), bins, H, W, _p(out))
    return out


def events_to_image(xs, ys, ps, H, W):
    out = np.zeros((H, W), np.float32)
    lib().tef_oracle_events_to_image(_p(_c(xs)), _p(_c(ys)), _p(_c(ps)), len(xs), H, W, _p(out))
    return out
