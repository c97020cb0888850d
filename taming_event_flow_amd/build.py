"""Build libtef_hip.so (the C-ABI library of include/tef.h) for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The .so is built IN-TREE (taming_event_flow_amd/libtef_hip.so):
it is git-ignored but travels to the GPU box with the snapshot.
"""

import glob
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libtef_hip.so")

# -ffp-contract=off: keep the reference's (unfused) fp32 op order on the parity-sensitive paths.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-munsafe-fp-atomics"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


STAMP = LIB + ".srchash"


def source_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy to the GPU box)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    for p in deps:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(PKG, "build", os.path.basename(src) + ".o")
        cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
