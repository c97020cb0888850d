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
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17"]

# -munsafe-fp-atomics (hardware floating-point atomic adds instead of compare-and-swap loops) only where a unit HAS
# floating-point atomics, and what they are.  None of them is on the integer paths whose results are bitwise reproducible
# (the IWE scatter K2 and the flow-gradient scatter K7 accumulate Q17.46 / block-floating INTEGERS in LDS):
UNSAFE_FP_ATOMICS = {
    "tef_conv": "split-K / bias epilogues of the weight-gradient kernels (EPI_ATOMIC): parameter gradients are summed in "
                "arrival order, i.e. reproducible only to fp32 summation order",
    "tef_cell": "bias gradient of the fused ConvGRU sweeps (one atomic per workgroup)",
    "tef_prims": "scatter_add / grid-sample backward of the stand-alone primitives (tests, metrics)",
    "tef_smooth": "backward of the smoothing priors (off in the reference configuration)",
    "tef_val": "validation metrics (flow_val): global float scatters like the reference's index_put_",
    "tef_encode": "event encodings: fp64 accumulators in LDS",
    "tef_loss": "the fp64 fall-back accumulators in LDS (general masks / >= 2^17 events): ds_add_f64",
}


def unit_flags(src):
    unit = os.path.splitext(os.path.basename(src))[0]
    return FLAGS + (["-munsafe-fp-atomics"] if unit in UNSAFE_FP_ATOMICS else [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


STAMP = LIB + ".srchash"


def source_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy to the GPU box)."""
    import hashlib

    h = hashlib.sha256((" ".join(FLAGS) + " " + " ".join(sorted(UNSAFE_FP_ATOMICS))).encode())
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
    force = force or os.environ.get("TEF_FORCE_BUILD", "0") == "1"      # (one full compile on the box: `TEF_FORCE_BUILD=1 python -c "import __graft_entry__ as g; g.build()"`)
    if not force and not needs_build():
        return LIB
    import hashlib

    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    hdr = hashlib.sha256(b"")
    for p in sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        with open(p, "rb") as f:
            hdr.update(f.read())
    procs = []
    for src in sources():
        obj = os.path.join(PKG, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        # one object per translation unit, recompiled only when the unit, a header or the flags changed
        h = hdr.copy()
        h.update(" ".join(unit_flags(src)).encode())
        with open(src, "rb") as f:
            h.update(f.read())
        stamp, digest = obj + ".srchash", h.hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
            continue
        cmd = [hipcc] + unit_flags(src) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd, stamp, digest))
        if len(procs) >= 4:                      # a few units at a time (the container has 8 cores)
            pr, c, st, dg = procs.pop(0)
            if pr.wait():
                raise subprocess.CalledProcessError(pr.returncode, c)
            with open(st, "w") as f:
                f.write(dg)
    for pr, c, st, dg in procs:
        if pr.wait():
            raise subprocess.CalledProcessError(pr.returncode, c)
        with open(st, "w") as f:
            f.write(dg)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
