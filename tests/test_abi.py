"""The C-ABI shared library loads on a CPU-only box and exports every entry point that
include/clairvoyante_amd.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "clairvoyante_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", src)))


def test_header_is_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "clairvoyante_amd.h"\nint main(void){cv_arch a; (void)a; return CV_NUM_PARAMS == 18 ? 0 : 1;}\n')
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c),
                           "-o", str(tmp_path / "t.o")])


def test_library_exports_every_declared_symbol():
    from clairvoyante_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    # the ctypes binding covers the same set
    assert sorted(_lib.EXPORTS) == names
    _lib.load()


def test_every_option_key_is_documented_in_the_header():
    """cv_set_option / cv_get_option take string keys: every key the library accepts is described in the header's option
    list (the dbg0..dbg7 development switches are named there as a family), and both functions know the same keys"""
    api = open(os.path.join(ROOT, "clairvoyante_amd", "csrc", "cv_api.hip")).read()
    header = open(os.path.join(ROOT, "include", "clairvoyante_amd.h")).read()
    setter = api[api.index('extern "C" int cv_set_option'):api.index('extern "C" int cv_get_option')]
    getter = api[api.index('extern "C" int cv_get_option'):]
    getter = getter[:getter.index("\n}\n")]
    set_keys = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', setter))
    get_keys = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', getter))
    assert len(set_keys) >= 10 and set_keys <= get_keys, sorted(set_keys - get_keys)
    for k in sorted(set_keys):
        assert '"%s"' % k in header, "option %s is not described in include/clairvoyante_amd.h" % k
    assert '"dbg0".."dbg7"' in header


def test_error_reporting_without_gpu():
    """bad arguments fail with a message instead of crashing"""
    from clairvoyante_amd import _lib
    lib = _lib.load()
    assert lib.cv_set_option(None, b"impl", 1) != 0
    assert b"null" in lib.cv_last_error()
    h = ctypes.c_void_p()
    arch = _lib.CvArch()
    arch.kh[:] = [1, 2, 3]; arch.cout[:] = [16, 32, 48]; arch.pool[:] = [5, 4, 3]; arch.fc4, arch.fc5 = 336, 168
    import torch
    if not torch.cuda.is_available():
        assert lib.cv_create(ctypes.byref(arch), 0, ctypes.byref(h)) != 0     # no device: loud failure
        assert len(lib.cv_last_error()) > 0
        with pytest.raises(_lib.CvError):
            from clairvoyante_amd import clairvoyante_v3
            clairvoyante_v3.Clairvoyante()


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under clairvoyante_amd/ may reference it"""
    pkg = os.path.join(ROOT, "clairvoyante_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "cv_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: no module of the package (nor the C sources) may refer to it"""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clairvoyante_amd")
    bad = []
    for dirpath, _dirs, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|from\s+\.\.?\s*oracle|cv_oracle|cvo_", text, re.M):
                    bad.append(f)
    assert bad == []


def test_host_batch_parts_cover_the_batch():
    """model._pipe_cuts: the parts predict(numpy) sends a large host batch to the device in -- contiguous, complete,
    boundaries on whole groups of 16 candidates, the first part the smallest (its copy is the one nothing hides), no part
    more than ~1.6 x the one before except the last (which takes a short tail)"""
    from clairvoyante_amd.model import Clairvoyante as C
    for n in (24576, 24577, 30000, 40003, 65536, 70001, 140000, 1000003):
        c = C._pipe_cuts(n)
        assert c[0] == 0 and c[-1] == n and all(b > a for a, b in zip(c[:-1], c[1:]))
        assert all(v % 16 == 0 for v in c[:-1])
        sizes = [b - a for a, b in zip(c[:-1], c[1:])]
        assert sizes[0] == min(sizes[:-1] or sizes) and len(sizes) >= 2
        assert all(b <= 1.61 * a + 16 for a, b in zip(sizes[:-2], sizes[1:-1]))
