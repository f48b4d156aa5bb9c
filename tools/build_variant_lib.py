"""Development: a second build of the library with extra compile flags, next to the in-tree one, for same-box A/B runs
through CV_HIP_LIB (tools/gpu_lib_ab.sh, tools/gpu_fast_selu_ab.sh):
    python tools/build_variant_lib.py fast -DCV_FAST_SELU     ->  clairvoyante_amd/csrc/libclairvoyante_hip_fast.so
Objects go to a temporary directory; the in-tree objects and library are not touched."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import build as b

tag, extra = sys.argv[1], sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="cv_variant_")
objs, procs = [], []
for s in b.SOURCES:
    obj = os.path.join(tmp, os.path.splitext(s)[0] + ".o")
    objs.append(obj)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + b.FLAGS + extra + ["-c", os.path.join(b.CSRC, s), "-o", obj]))
assert all(p.wait() == 0 for p in procs)
out = os.path.join(b.CSRC, "libclairvoyante_hip_%s.so" % tag)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lz"])
print(out)
