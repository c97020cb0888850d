"""Host-side AddressSanitizer + UndefinedBehaviorSanitizer run of libtef_hip.so WITHOUT a GPU (tests/asan/): the library's
host code instrumented (tools/build_asan_host.sh; device code as usual), the HIP runtime replaced by a dry-run stub, and
every host path driven with the real callers' plans — Trainer windows at the training shape, a padded shape, group sizes 1 / 3 /
10, a window cut short, Linear / multi-scale / smoothing losses, TEF_MAX_PASSES passes, loader, encodings, validation
metrics.  The first run builds the instrumented library (about 90 s of hipcc); later runs take 20 s."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "asan", "run_host_asan.sh")

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc to build the instrumented library")


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("TEF_HIP_LIB", "LD_PRELOAD")}
    return subprocess.run([SCRIPT, *args], env=env, capture_output=True, text=True, timeout=1500)


def test_sanitizer_sees_a_host_overrun():
    """The run must be able to fail: tef_net_layout given an array one entry short is a heap-buffer-overflow report."""
    r = _run("--self-test")
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in out and "tef_net_layout" in out, out[-2000:]
    assert "SELF-TEST FAILED" not in out


def test_host_paths_clean_under_asan_ubsan():
    r = _run()
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "HOST SANITIZER RUN CLEAN" in out, out[-3000:]
    assert "ok: tef_update_pass / tef_update_window" in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
