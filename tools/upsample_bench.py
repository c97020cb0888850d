"""Times of the bilinear up-sampling kernels at the shapes of one RecEVFlowNet pass (B = 8, 128 x 128): the four decoder
inputs (x2) and the four flow heads (x8 / x4 / x2 / x1), forward and backward, against the HBM floor of their bytes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g  # noqa: E402

g.build()
from taming_event_flow_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("dec0", 8 * 512, 8, 8, 2), ("dec1", 8 * 256, 16, 16, 2), ("dec2", 8 * 128, 32, 32, 2), ("dec3", 8 * 64, 64, 64, 2),
          ("flow0", 16, 16, 16, 8), ("flow1", 16, 32, 32, 4), ("flow2", 16, 64, 64, 2), ("flow3", 16, 128, 128, 1)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':8s} {'fwd us':>8s} {'floor':>7s} {'bwd us':>8s} {'floor':>7s}")
for name, planes, H, W, s in shapes:
    x = torch.randn(planes, H, W, device=dev)
    y = torch.empty(planes, H * s, W * s, device=dev)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    fwd = timed(lambda: lib.tef_upsample_bilinear_crop(p(x), None, planes, H, W, s, s, 1.0, 0, 0, p(y), st))
    bwd = timed(lambda: lib.tef_upsample_bilinear_crop_backward(p(dy), planes, H, W, s, s, 1.0, 0, 0, p(dx), st))
    floor = (x.numel() + y.numel()) * 4 / 4e12 * 1e6        # at 4 TB/s
    print(f"{name:8s} {fwd:8.1f} {floor:7.1f} {bwd:8.1f} {floor:7.1f}")
