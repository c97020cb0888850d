"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol include/tef.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from taming_event_flow_amd import _lib

    return _lib


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(tef_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(built.LIB_PATH)
    decl = declared_symbols()
    assert decl, "no declarations found"
    for name in sorted(decl):
        assert hasattr(handle, name), f"{name} declared in include/ but not exported"
    # and the Python binding table covers the same set
    assert set(built.SIGNATURES) == decl


def test_version_and_size_queries(built):
    lib = built.lib()
    assert lib.tef_version() == 1
    cfg = built.LossCfg()
    cfg.kind, cfg.B, cfg.H, cfg.W, cfg.P, cfg.F, cfg.S, cfg.mode_div = 0, 2, 16, 20, 4, 2, 1, 2
    cfg.M, cfg.Md = 256, 0
    for t in range(5):
        cfg.off[t] = 64 * t
    assert lib.tef_loss_workspace_bytes(ctypes.byref(cfg)) > 0
    # invalid configurations are rejected with a message, not a crash
    cfg.mode_div = 4
    assert lib.tef_loss_workspace_bytes(ctypes.byref(cfg)) == 0
    assert b"iterative_mode" in lib.tef_last_error()
    cfg.mode_div, cfg.S = 2, 4   # 4 >> 3 = 0 passes at the last scale
    assert lib.tef_loss_workspace_bytes(ctypes.byref(cfg)) == 0


def test_product_path_refuses_cpu_tensors(built):
    import torch

    from taming_event_flow_amd.loss.flow import Iterative

    cfg = {"loader": {"resolution": [16, 20], "batch_size": 1},
           "loss": {"flow_spat_smooth_weight": None, "flow_temp_smooth_weight": None, "round_ts": False,
                    "iterative_mode": "two"},
           "data": {"passes_loss": 2, "scales_loss": 1}}
    L = Iterative(cfg, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.update([torch.zeros(1, 2, 16, 20)], torch.zeros(1, 4, 4), torch.zeros(1, 4, 2), torch.zeros(1, 0, 4),
                 torch.zeros(1, 0, 2))
    cfg["loss"]["iterative_mode"] = "four"
    with pytest.raises(NotImplementedError):
        Iterative(cfg, torch.device("cpu"))
