#!/usr/bin/env python3
"""Experiment: per-workgroup phase clocks of the implicit-GEMM kernel (needs a -DTEF_CONV_STAMP variant library,
tools/build_variant.sh STAMP -DTEF_CONV_STAMP; run with TEF_HIP_LIB pointing at it)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taming_event_flow_amd import _lib                      # noqa: E402
from taming_event_flow_amd.models import submodules as sm   # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(_lib.LIB_PATH)
stamps = torch.zeros((1 << 16, 4), dtype=torch.int64, device=dev)
assert lib.tef_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
for name, c0, c1, n, res in (("enc0 gates", 64, 64, 128, 64), ("enc1 gates", 128, 128, 256, 32), ("enc2 gates", 256, 256, 512, 16),
                             ("enc3 gates", 512, 512, 1024, 8), ("dec3", 66, 0, 32, 128)):
    x0 = torch.randn(8, c0, res, res, device=dev)
    x1 = torch.randn(8, c1, res, res, device=dev) if c1 else None
    w = torch.randn(n, c0 + c1, 3, 3, device=dev) * 0.05
    b = torch.zeros(n, device=dev)
    pk = sm.PackedWeights()
    for _ in range(3):
        stamps.zero_()
        sm.conv2d(pk, x0, w, b, stride=1, act="relu", x1=x1)
        torch.cuda.synchronize()
    s = stamps.cpu()
    s = s[s[:, 0] > 0].double()
    t0 = s[:, 0].min()
    print(f"{name:12s} WGs {len(s):5d}  prologue {float((s[:,1]-s[:,0]).mean()):9.0f}  main loop {float((s[:,2]-s[:,1]).mean()):9.0f}  "
          f"epilogue {float((s[:,3]-s[:,2]).mean()):9.0f}  | start spread {float((s[:,0]-t0).max()):9.0f}  kernel span {float(s[:,3].max()-t0):9.0f} clocks")
