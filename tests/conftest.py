import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    """Load a golden loss case -> (meta dict, window lists, loss, dflows)."""
    import json

    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    P, F = meta["P"], meta["F"]
    win = {
        "flows": [[z["flows"][t, i] for i in range(F)] for t in range(P)],
        "ev": [z[f"ev{t}"] for t in range(P)],
        "pm": [z[f"pm{t}"] for t in range(P)],
        "dev": [z[f"dev{t}"] for t in range(P)],
        "dpm": [z[f"dpm{t}"] for t in range(P)],
    }
    return meta, win, np.float32(z["loss"]), z["dflows"]


ITERATIVE_CASES = [
    "it_two_s1_p6", "it_one_s1_p4", "it_two_s2_p8", "it_two_s3_p8", "it_two_s1_p10_f4", "it_two_iid",
    "it_two_zero_flow", "it_two_smooth_terms", "it_two_round_ts", "it_two_float_xy", "it_two_p5_odd",
]
LINEAR_CASES = ["lin_s1_p6", "lin_s2_p8", "lin_smooth_terms", "lin_zero_flow"]


def rel_err(a, b):
    """max-norm relative error of arrays (or scalars)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
