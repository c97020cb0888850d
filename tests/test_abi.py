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


def _header_struct_fields(name):
    """Field names (in order) and total int-sized element count of `struct name` in include/tef.h."""
    text = open(os.path.join(ROOT, "include", "tef.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#define\s+(\w+)\s+(\d+)", lambda m: m.group(0), text)
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(TEF_\w+)\s+(\d+)", text)}
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        for item in names.split(","):
            item = item.strip().lstrip("*").strip()
            m = re.match(r"(\w+)\s*(?:\[(.*?)\])?$", item)
            count = 1
            if m.group(2):
                expr = m.group(2)
                for k, v in defines.items():
                    expr = expr.replace(k, str(v))
                count = eval(expr)
            fields.append((m.group(1), count))
    return fields


def test_integration_doc_structs_match_the_header(built):
    """INTEGRATION.md section 2 shows a host binding to copy from: its ctypes structures must lay out like include/tef.h
    (a missing trailing field hands the library a struct 4 bytes short)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes\nlib = ctypes\.CDLL.*?)```", doc, flags=re.S).group(1)
    code = "\n".join(l for l in block.splitlines() if not l.startswith("lib"))      # (do not load a library here)
    ns = {}
    exec(code, ns)
    for cname in ("tef_events", "tef_loss_cfg"):
        hdr = _header_struct_fields(cname)
        doc_fields = [(n, getattr(t, "_length_", 1)) for n, t in ns[cname]._fields_]
        assert doc_fields == hdr, (cname, doc_fields, hdr)
    assert ctypes.sizeof(ns["tef_loss_cfg"]) == ctypes.sizeof(built.LossCfg)
    assert ctypes.sizeof(ns["tef_events"]) == ctypes.sizeof(built.Events) if hasattr(built, "Events") else True
