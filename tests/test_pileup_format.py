"""Host-only pieces of the pileup front end (no GPU): the row formatter of the C ABI reproduces the
reference's rows character for character, the CLI carries the reference's options and defaults."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from test_pileup_oracle import load_case  # noqa: E402


def test_format_row_reproduces_reference_rows():
    from clairvoyante_amd import _lib
    lib = _lib.load()
    _, _, _, _, want = load_case("noisy")
    buf = ctypes.create_string_buffer(1 << 14)
    for row in want[:40]:
        f = row.split(" ")
        counts = np.asarray(f[3:], dtype=np.float32)
        assert counts.size == 528
        n = lib.cv_format_tensor_row(f[0].encode(), int(f[1]), f[2].encode(), len(f[2]),
                                     counts.ctypes.data_as(ctypes.c_void_p), buf, len(buf))
        assert buf.raw[:n].decode() == row
    # values "%0.1f" has to round, negative values, a buffer that is too small
    odd = np.zeros(528, dtype=np.float32); odd[0] = 2.25; odd[1] = -3.0; odd[2] = 1e9
    n = lib.cv_format_tensor_row(b"c", 7, b"ACG", 3, odd.ctypes.data_as(ctypes.c_void_p), buf, len(buf))
    assert buf.raw[:n].decode().split(" ")[:6] == ["c", "7", "ACG", "2.2", "-3.0", "1000000000.0"]
    assert lib.cv_format_tensor_row(b"c", 7, b"ACG", 3, odd.ctypes.data_as(ctypes.c_void_p), buf, 100) == -1


def test_createtensor_cli_defaults():
    from clairvoyante_amd import CreateTensor
    a = CreateTensor.build_parser().parse_args(["--ctgName", "chr1"])
    assert (a.bam_fn, a.ref_fn, a.can_fn, a.tensor_fn) == ("input.bam", "ref.fa", "PIPE", "PIPE")
    assert (a.minMQ, a.dcov, a.minCoverage, a.considerleftedge, a.samtools) == (0, 250, 0, True, "samtools")
    assert a.ctgStart is None and a.ctgEnd is None
    a = CreateTensor.build_parser().parse_args(["--ctgStart", "99", "--ctgEnd", "500", "--considerleftedge", "False"])
    assert CreateTensor.region_of(a) == (100, 500, 1, 1000500) and a.considerleftedge is False
