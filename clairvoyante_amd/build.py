"""Builds csrc/libclairvoyante_hip.so (hipcc, gfx950 only) in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libclairvoyante_hip.so")
SOURCES = ["cv_api.hip", "cv_kernels_ref.hip", "cv_kernels_mfma.hip", "cv_kernels_wgrad.hip", "cv_post.hip", "cv_train.hip", "cv_pileup.hip", "cv_hostio.cpp", "cv_bam.cpp", "cv_inflate.cpp"]
# -ffp-contract=off: the canonical arithmetic fuses only where fmaf()/MFMA say so
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("CV_EXTRA_FLAGS", "").split()       # development: e.g. -DCV_SETPRIO
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "clairvoyante_amd.h"))
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
