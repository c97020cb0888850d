#!/usr/bin/env python3
"""Accuracy anchor for the BPTT parameter gradients — build container only.

Window 0 of tests/golden/train_trace.npz (same seed, same inputs, same weights) is run through the reference's
RecEVFlowNet + Iterative loss twice: in float32 (what train_trace.npz recorded) and in FLOAT64 (model.double(),
default dtype float64 so that every buffer the loss allocates is double).  Of every parameter's gradient a fixed,
seeded subset of at most SAMPLES elements is stored in both precisions, plus the float64 norm of the whole gradient:
tests/test_train_gpu.py::test_bptt_gradient_accuracy_anchor holds the HIP path's distance to the float64 gradient to
1.5 x the distance of the reference's own float32 run, per parameter and globally.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from taming_event_flow_amd import synth  # noqa: E402

sys.path.insert(0, "/root/reference")
import warnings  # noqa: E402

warnings.filterwarnings("ignore")
from dataloader.encodings import events_to_channels  # noqa: E402
from loss.flow import Iterative  # noqa: E402
from models.model import RecEVFlowNet  # noqa: E402

torch.set_num_threads(8)
SAMPLES = 2048


def run(dtype, trace):
    torch.set_default_dtype(dtype)
    H, W, B, P = int(trace["H"]), int(trace["W"]), int(trace["B"]), int(trace["P"])
    config = {
        "loader": {"resolution": [H, W], "batch_size": B},
        "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False,
                 "iterative_mode": "two", "flow_scaling": 32, "clip_grad": float(trace["clip"])},
        "data": {"passes_loss": P, "scales_loss": 1},
    }
    model = RecEVFlowNet({"name": "RecEVFlowNet", "final_w_scale": 0.01}, 2)
    sd = model.state_dict()
    wts = synth.make_model_weights([(k, v.shape) for k, v in sd.items()], int(trace["seed"]))
    model.load_state_dict({k: torch.tensor(v, dtype=dtype) for k, v in wts.items()})
    model.train()
    loss_function = Iterative(config, torch.device("cpu"))
    loss_function.reset()
    model.reset_states()
    for t in range(P):
        ev, pm = trace[f"ev0_{t}"], trace[f"pm0_{t}"]
        dev, dpm = trace[f"dev0_{t}"], trace[f"dpm0_{t}"]
        net_input = torch.tensor(trace[f"inp0_{t}"], dtype=dtype)
        x = model(net_input)
        for i in range(len(x["flow"])):
            x["flow"][i] = x["flow"][i] * config["loss"]["flow_scaling"]
        loss_function.update(x["flow"], torch.tensor(ev, dtype=dtype), torch.tensor(pm, dtype=dtype),
                             torch.tensor(dev, dtype=dtype), torch.tensor(dpm, dtype=dtype))
    loss = loss_function()
    loss.backward()
    names = [n for n, _ in model.named_parameters()]
    return float(loss.item()), names, [p.grad.detach().double().numpy().ravel().copy() for p in model.parameters()]


def main():
    trace = np.load(os.path.join(HERE, "train_trace.npz"))
    l32, names, g32 = run(torch.float32, trace)
    l64, _, g64 = run(torch.float64, trace)
    assert np.float32(l32) == trace["loss0"], (l32, trace["loss0"])      # the float32 run IS the recorded trace
    rng = np.random.default_rng(77)
    out = dict(loss32=np.float64(l32), loss64=np.float64(l64), names=np.array(names))
    idx, s32, s64, n64, e32 = [], [], [], [], []
    for a, b in zip(g32, g64):
        k = np.sort(rng.choice(a.size, min(SAMPLES, a.size), replace=False)).astype(np.int64)
        idx.append(k)
        s32.append(a[k].astype(np.float32))
        s64.append(b[k])
        n64.append(np.sqrt((b * b).sum()))
        e32.append(np.sqrt(((a - b) ** 2).sum()))
    out["offsets"] = np.cumsum([0] + [len(k) for k in idx])
    out["index"] = np.concatenate(idx)
    out["g32"] = np.concatenate(s32)
    out["g64"] = np.concatenate(s64)
    out["norm64"] = np.array(n64)
    out["err32_full"] = np.array(e32)        # || reference fp32 - reference fp64 || over the WHOLE parameter
    path = os.path.join(HERE, "train_trace_f64.npz")
    np.savez_compressed(path, **out)
    g = np.sqrt(sum(e * e for e in e32)) / np.sqrt(sum(n * n for n in n64))
    worst = max(e / max(n, 1e-30) for e, n in zip(e32, n64))
    print(f"loss fp32 {l32:.8f} fp64 {l64:.8f} rel {abs(l32 - l64) / abs(l64):.2e}; gradient fp32 vs fp64: global {g:.2e}, "
          f"worst parameter {worst:.2e}; {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
