"""TEF_CONV_MATH=bf16x3 (csrc/tef_conv.hip conv3x3_x3_kernel, opt-in): forward and input gradient of the eligible 3x3 layers on
error-compensated bf16 splits against the exact-fp32 MFMA path — each mode in a process of its own (the switch is read once)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16x3_mode_against_fp32_mode(tmp_path):
    files = {}
    for mode in ("fp32", "bf16x3"):
        files[mode] = str(tmp_path / f"{mode}.npz")
        env = dict(os.environ, TEF_CONV_MATH=mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_math_check.py"), files[mode]], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = np.load(files["fp32"]), np.load(files["bf16x3"])
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 20
    differ = 0
    for k in a.files:
        scale = np.abs(a[k]).max()
        err = np.abs(a[k] - b[k]).max() / max(scale, 1e-30)
        # weight gradients stay on the fp32 kernels: summation-order noise of their atomics only; outputs and input gradients
        # carry the dropped lo * lo terms and the 16-bit significand of hi + lo (2^-17 per operand): 3e-6 ... 1e-5 of the
        # largest element measured, inside the 1e-4 bar
        assert err <= 5e-5, (k, err)
        differ += int(not np.array_equal(a[k], b[k]) and (k.endswith(".y") or ".dx" in k))
    assert differ >= 6, "the bf16x3 kernels were not taken (outputs and input gradients bit-identical to the fp32 mode)"
