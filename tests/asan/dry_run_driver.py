#!/usr/bin/env python3
"""Host-side sanitizer run of libtef_hip.so WITHOUT a GPU (tests/asan/run_host_asan.sh sets the environment up):

  * the library is the ASan + UBSan build of tools/build_asan_host.sh (host code instrumented, device code as usual);
  * the HIP runtime is tests/asan/hip_stub.c: launches do nothing, hipMemsetAsync writes for real;
  * "device" tensors are host tensors: the product's device checks and stream look-ups are patched out HERE, in the test
    process only (the product refuses host tensors).

What runs is every host path of the library with the plans the real callers build: train.Trainer windows (RecEVFlowNet pass
forward / backward / deferred weight gradients through models/engine.py, loss update / forward / backward, clip + Adam) at
the training shape, a padded odd shape, several window lengths up to TEF_MAX_PASSES and weight-gradient group sizes; the
loader stage; the encodings; the validation metrics.  The numbers are garbage (no kernel ran); the point is that the host code
indexes, sizes and writes nothing out of bounds.  Any sanitizer report aborts the process.
"""
import contextlib
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

assert "TEF_HIP_LIB" in os.environ, "run through tests/asan/run_host_asan.sh"
from taming_event_flow_amd import _lib  # noqa: E402


class _Stream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass

    def synchronize(self):
        pass

    def __eq__(self, other):
        return isinstance(other, _Stream)

    def __hash__(self):
        return 0


_lib.require_device_tensor = lambda t, name: t
_lib.stream_ptr = lambda: None
torch.cuda.current_stream = lambda *a, **k: _Stream()
torch.cuda.Stream = lambda *a, **k: _Stream()
torch.cuda.stream = lambda s: contextlib.nullcontext()
torch.Tensor.record_stream = lambda self, s: None
torch.cuda.is_current_stream_capturing = lambda: False

from taming_event_flow_amd import synth, train  # noqa: E402
from taming_event_flow_amd.dataloader import base as dl_base  # noqa: E402
from taming_event_flow_amd.dataloader import encodings  # noqa: E402
from taming_event_flow_amd.loss import flow as loss_flow  # noqa: E402
from taming_event_flow_amd.loss import flow_val  # noqa: E402
from taming_event_flow_amd.models import engine as eng_mod  # noqa: E402
from taming_event_flow_amd.models import submodules  # noqa: E402

for mod in (loss_flow, flow_val, eng_mod, submodules, dl_base, encodings):      # modules that bound the names at import
    for name in ("require_device_tensor", "stream_ptr"):
        if hasattr(mod, name):
            setattr(mod, name, getattr(_lib, name))

dev = torch.device("cpu")
lib = _lib.lib()
raw = __import__("ctypes").CDLL(os.environ["TEF_HIP_STUB"])
print("library:", _lib.LIB_PATH, flush=True)

if "--self-test" in sys.argv:
    # the sanitizer must see a host overrun by the library: tef_net_layout writes `levels` offsets into each array it is
    # given; hand it one that is one entry short -> AddressSanitizer aborts the process (the caller checks for that)
    import ctypes

    from taming_event_flow_amd.models.model import RecEVFlowNet

    net = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2)
    e = net.arch.engine
    pl = e.make_plan(1, 32, 32, 0, 0)
    n = e.plan.levels
    short = (ctypes.c_size_t * (n - 1))()
    ok_ = (ctypes.c_size_t * n)()
    dx = ctypes.c_size_t()
    lib.tef_net_layout(ctypes.byref(pl), short, ok_, ok_, ctypes.byref(dx))
    print("SELF-TEST FAILED: the overrun went unnoticed", flush=True)
    sys.exit(0)


def window(cfg, passes, wgrad_group, streams, new_seq_at=None, events=600, windows=2, window_decode=False):
    os.environ["TEF_WGRAD_GROUP"] = str(wgrad_group)
    torch.manual_seed(0)
    tr = train.Trainer(cfg, dev, streams=streams, window_decode=window_decode)
    if streams:       # the side streams are dummies: the control flow (split passes, grouped flushes) is the real one
        e = tr.model.arch.engine
        e.wgrad_stream, e.wgrad_group = _Stream(), wgrad_group
        tr.wgrad_stream = e.wgrad_stream
        if not window_decode:
            e.side_stream = _Stream()
            tr.dec_stream = e.side_stream
        else:         # the encoder levels of consecutive passes by level range (tef_net_pass_forward_levels / _backward_levels)
            e.enc_streams = (_Stream(), _Stream())
    src = train.SyntheticSequences(cfg, dev, events, seq_len=10 ** 9, seed=3, jitter=50)
    tr.reset()
    steps = 0
    for t in range(passes * windows + (new_seq_at or 0)):
        if tr.step(src.next(), new_seq=(new_seq_at is not None and t == new_seq_at)):
            steps += 1
    assert steps == windows, (steps, windows)
    tr.close()


def cfg_for(B, H, W, P, warping="Iterative", scales=1, smooth=False, max_grad=400):
    cfg = copy.deepcopy(train.DEFAULT_CONFIG)
    cfg["loader"].update(batch_size=B, resolution=[H, W], max_num_grad_events=max_grad)
    cfg["data"].update(passes_loss=P, scales_loss=scales)
    cfg["loss"].update(warping=warping)
    if smooth:
        cfg["loss"].update(flow_spat_smooth_weight=0.001, flow_temp_smooth_weight=0.1)
    return cfg


cases = [
    ("training shape B=8 128x128 P=10, groups of 3", cfg_for(8, 128, 128, 10), 10, 3, True, None),
    ("the same on one stream", cfg_for(8, 128, 128, 10), 10, 3, False, None),
    ("padded shape 40x52, B=2, P=4, groups of 1", cfg_for(2, 40, 52, 4), 4, 1, True, None),
    ("groups of 10 (one flush)", cfg_for(2, 64, 64, 10), 10, 10, True, None),
    ("window cut short by new_seq", cfg_for(2, 64, 64, 4), 4, 3, True, 2),
    ("Linear, two scales, smoothing", cfg_for(2, 48, 64, 8, "Linear", 2, True), 8, 3, True, None),
    ("Iterative, three scales", cfg_for(1, 32, 32, 8, "Iterative", 3, True), 8, 3, False, None),
    ("TEF_MAX_PASSES passes", cfg_for(1, 32, 32, 64, max_grad=100), 64, 3, True, None),
]
for name, cfg, P, group, streams, ns in cases:
    window(cfg, P, group, streams, ns, windows=1 if P == 64 else 2)
    print("ok:", name, flush=True)
# window mode (round 6): encoder halves pass by pass, the decoder halves of the window as one batch
for name, cfg, P, group, streams, ns in cases:
    window(cfg, P, group, streams, ns, windows=1 if P == 64 else 2, window_decode=True)
    print("ok (window decode):", name, flush=True)

# loss module on its own: ragged / empty lists, general masks, detached events
rng = np.random.default_rng(0)
for kind, S, P in (("Iterative", 1, 4), ("Iterative", 2, 8), ("Linear", 1, 4), ("Linear", 3, 8)):
    B, H, W, F = 2, 24, 28, 2
    win = synth.make_window(rng, B, H, W, P, F, [50, 0, 70, 40] * (P // 4), [20, 0, 0, 10] * (P // 4), sigma=1.0, ragged=True)
    cfg = {"loader": {"resolution": [H, W], "batch_size": B},
           "loss": {"flow_spat_smooth_weight": 0.01, "flow_temp_smooth_weight": 0.1, "round_ts": False, "iterative_mode": "two"},
           "data": {"passes_loss": P, "scales_loss": S}}
    L = getattr(loss_flow, kind)(cfg, dev)
    flows = [[torch.tensor(win["flows"][t][i], requires_grad=True) for i in range(F)] for t in range(P)]
    for t in range(P):
        L.update(flows[t], torch.tensor(win["ev"][t]), torch.tensor(win["pm"][t]), torch.tensor(win["dev"][t]), torch.tensor(win["dpm"][t]))
    L().backward()
    L.reset()
print("ok: loss modules", flush=True)

# the one-call forms of update() straight through the C ABI (the Python fast path wants device tensors): one pass
# (tef_update_pass) and a whole window with the most heads and passes the records can describe (tef_update_window: the
# records travel in the kernel arguments, several launches)
import ctypes  # noqa: E402

for F, P, N, Nd in ((2, 6, 500, 120), (16, 64, 70, 0), (1, 3, 0, 40), (4, 10, 20000, 5000)):
    B, H, W = 2, 24, 28
    flows = [[torch.zeros(B, 2, H, W) for _ in range(F)] for _ in range(P)]
    planar, yx = torch.zeros(P, F, B, 2, H, W), torch.zeros(P, F, B, H, W, 2)
    g, d = loss_flow._SoA(B, dev, (H, W)), loss_flow._SoA(B, dev, (H, W))
    evs = [(torch.rand(B, N, 4), torch.ones(B, N, 2), torch.rand(B, Nd, 4), torch.ones(B, Nd, 2)) for _ in range(P)]
    descs = (_lib.UpdateDesc * P)()
    keep = []
    for t in range(P):
        slot0, dslot0 = g.reserve(N), d.reserve(Nd)
        ptrs = (ctypes.c_void_p * F)(*[f.data_ptr() for f in flows[t]])
        sb = (ctypes.c_long * F)(*[f.stride(0) for f in flows[t]])
        sc = (ctypes.c_long * F)(*[f.stride(1) for f in flows[t]])
        keep += [ptrs, sb, sc]
        u = descs[t]
        u.flows, u.stride_b, u.stride_c = (ctypes.cast(x_, ctypes.c_void_p) for x_ in (ptrs, sb, sc))
        ev, pm, dv, dpm = evs[t]
        u.ev, u.pm, u.dev, u.dpm = ev.data_ptr(), pm.data_ptr(), dv.data_ptr(), dpm.data_ptr()
        u.N, u.Nd, u.pass_idx, u.slot0, u.dslot0 = N, Nd, t, slot0, dslot0
        g.commit(N)
        d.commit(Nd)
    _lib.check(lib.tef_update_window(descs, P, F, B, H, W, planar.data_ptr(), yx.data_ptr(), g.struct_ref(), d.struct_ref(), None),
               "tef_update_window")
    t = P - 1
    _lib.check(lib.tef_update_pass(keep[-3], keep[-2], keep[-1], F, B, H, W, planar[t].data_ptr(), yx[t].data_ptr(),
                                   evs[t][0].data_ptr(), evs[t][1].data_ptr(), N, None, evs[t][2].data_ptr(), evs[t][3].data_ptr(), Nd,
                                   None, t, descs[t].slot0, descs[t].dslot0, g.struct_ref(), d.struct_ref(), None), "tef_update_pass")
print("ok: tef_update_pass / tef_update_window", flush=True)

# encodings + validation metrics (batch 1)
H, W = 40, 52
ev, pm = synth.make_event_pass(rng, 1, 3000, H, W)
encodings.event_list_to_channels(torch.tensor(ev), (H, W))
encodings.events_to_voxel(torch.tensor(ev[0, :, 2]), torch.tensor(ev[0, :, 1]), torch.tensor(ev[0, :, 0]), torch.tensor(ev[0, :, 3]), 5, (H, W))
V = flow_val.Iterative({"loader": {"resolution": [H, W]}, "loss": {"round_ts": False}, "vis": {"mask_output": True}, "metrics": {}}, dev)
for d in synth.make_eval_window(5, H, W, 3, 2000):
    V.update([torch.tensor(d["low"]), torch.tensor(d["flow"])], torch.tensor(d["ev"]), torch.tensor(d["pm"]), torch.tensor(d["mask"]))
    V.rsat(), V.fwl()
V.window_iwe(mode="forward", round_idx=False), V.window_flow(mode="backward", mask=True), V.window_events(round_idx=True)
print("ok: encodings, validation metrics", flush=True)
print("launches swallowed by the stub:", raw.tef_dry_run_launches(), flush=True)
print("HOST SANITIZER RUN CLEAN", flush=True)
