#!/usr/bin/env python3
"""Outputs and input / parameter gradients of a few convolution layers whose forward and input-gradient kernels are eligible
for TEF_CONV_MATH=bf16x3 (rows of 32 / 64 / 128 pixels, no split over k), on seeded inputs -> an .npz.  Run once per math mode
(the switch is read once per process); tests/test_conv_math_gpu.py compares the two files.

    python tools/conv_math_check.py OUT.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from taming_event_flow_amd.models import submodules as sm  # noqa: E402

dev = torch.device("cuda:0")
out = {}
torch.manual_seed(0)


def run(name, module, inputs):
    module = module.to(dev)
    xs = [x.to(dev).requires_grad_(True) for x in inputs]
    y = module(*xs)
    y = y[0] if isinstance(y, (tuple, list)) else y
    w = torch.randn_like(y)
    (y * w).sum().backward()
    out[name + ".y"] = y.detach().cpu().numpy()
    for i, x in enumerate(xs):
        out[f"{name}.dx{i}"] = x.grad.cpu().numpy()
    for k, p in module.named_parameters():
        out[f"{name}.d{k}"] = p.grad.cpu().numpy()


run("gru64", sm.ConvGRU(64, 64, 3), [torch.randn(8, 64, 64, 64), torch.randn(8, 64, 64, 64) * 0.5])      # gates (two sources) + gated out gate
# (smooth activations: with relu a pre-activation 4e-6 from zero flips its mask, and the gradients of the two modes then differ by
# whole elements — as they would between any two summation orders)
run("conv128_tanh_64", sm.ConvLayer(128, 128, 3, activation="tanh"), [torch.randn(8, 128, 64, 64) * 0.1])
run("conv64_tanh_128", sm.ConvLayer(64, 64, 3, activation="tanh"), [torch.randn(2, 64, 128, 128) * 0.2])
run("conv144_32", sm.ConvLayer(144, 128, 3, activation="sigmoid"), [torch.randn(32, 144, 32, 32) * 0.1])
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays; TEF_CONV_MATH =", os.environ.get("TEF_CONV_MATH", "(fp32)"))
