"""Native `samtools view` (csrc/cv_bam.cpp through clairvoyante_amd/bam.py), host only: BAM files written by
tests/bam_writer.py from the golden SAM text must come back as that text -- whole contig and regions, with and
without the .bai, records straddling BGZF blocks, several contigs, the -F mask, optional QUAL; `faidx` against the
test stand-in for samtools."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, ".."))
from bam_writer import write_bam  # noqa: E402

G = os.path.join(HERE, "golden", "pileup")
_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def sam_records(name):
    return [l.rstrip("\n") for l in open(os.path.join(G, name + ".sam")) if not l.startswith("@")]


def expected(lines, ctg, start, end, mask=2308, with_qual=False):
    out = []
    for l in lines:
        f = l.split("\t")
        if f[2] != ctg or (int(f[1]) & mask):
            continue
        pos = int(f[3])
        span = sum(int(n) for n, op in _CIG.findall(f[5]) if op in "MDN=X") or 1
        if start is not None and (pos + span - 1 < start or pos > end):
            continue
        q = f[10] if with_qual and len(f[10]) == len(f[9]) else "*"
        out.append("\t".join(f[:10] + [q]))
    return out


def view_text(path, ctg, start=None, end=None, **kw):
    from clairvoyante_amd.bam import BamFile
    bf = BamFile(path, threads=kw.pop("threads", 3))
    text = b"".join(bf.view(ctg, start, end, **kw)).decode()
    bf.close()
    return text.splitlines()


@pytest.mark.parametrize("case", ["plain", "noisy", "eqx", "handmade"])
@pytest.mark.parametrize("payload,index", [(60000, True), (777, True), (4093, False)])
def test_native_view_equals_the_sam_text(case, payload, index, tmp_path):
    recs = sam_records(case)
    L = 3000 if case != "handmade" else 320
    bam = str(tmp_path / "a.bam")
    write_bam(bam, recs, [("ctgA", max(L, 4000)), ("other", 10)], block_payload=payload, index=index)
    from clairvoyante_amd.bam import BamFile
    bf = BamFile(bam)
    assert bf.references() == [("ctgA", max(L, 4000)), ("other", 10)] and bf.has_index() == index
    bf.close()
    assert view_text(bam, "ctgA") == expected(recs, "ctgA", None, None)
    for start, end in ((1, 50), (601, 2900), (1500, 1500), (2990, 100000), (17, 33)):
        assert view_text(bam, "ctgA", start, end) == expected(recs, "ctgA", start, end), (start, end)
    assert view_text(bam, "other") == [] and view_text(bam, "nope") == []
    assert view_text(bam, "ctgA", exclude_flags=16) == expected(recs, "ctgA", None, None, mask=16)
    assert view_text(bam, "ctgA", 1, 400, with_qual=True, threads=1) == expected(recs, "ctgA", 1, 400, with_qual=True)


def test_native_view_on_several_contigs_and_a_long_file(tmp_path):
    from clairvoyante_amd import synth_pileup as sp
    recs = []
    refs = []
    for k, (name, L) in enumerate((("c1", 40000), ("c2", 70000), ("c3", 1000))):
        _ref, lines = sp.make_alignments(seed=40 + k, ref_len=L, n_reads=3000 if L > 1000 else 20, ctg=name,
                                         profile=sp.NOISY_PROFILE, read_len=(30, 400))
        recs += lines
        refs.append((name, L))
    bam = str(tmp_path / "m.bam")
    write_bam(bam, recs, refs, block_payload=30011)
    for ctg, L in refs:
        assert view_text(bam, ctg, threads=5) == expected(recs, ctg, None, None)
        for start in (1, L // 3, L - 500):
            assert view_text(bam, ctg, start, start + 2000) == expected(recs, ctg, start, start + 2000)
    nobai = str(tmp_path / "n.bam")
    write_bam(nobai, recs, refs, block_payload=65000, index=False)
    assert view_text(nobai, "c2", 30000, 31000) == expected(recs, "c2", 30000, 31000)


def test_native_faidx_equals_the_stand_in(tmp_path):
    from clairvoyante_amd.bam import faidx
    rng = np.random.RandomState(4)
    fa = str(tmp_path / "r.fa")
    seqs = {"ctgA": "".join("ACGTNacgt"[i] for i in rng.randint(0, 9, 1234)), "b": "".join("ACGT"[i] for i in rng.randint(0, 4, 61))}
    off = 0
    with open(fa, "w") as fh, open(fa + ".fai", "w") as fi:
        for name, s in seqs.items():
            hdr = ">%s test\n" % name
            fh.write(hdr); off += len(hdr)
            fi.write("%s\t%d\t%d\t60\t61\n" % (name, len(s), off))
            for i in range(0, len(s), 60):
                fh.write(s[i:i + 60] + "\n"); off += len(s[i:i + 60]) + 1
    fake = [sys.executable, os.path.join(HERE, "golden", "fake_samtools.py")]
    for ctg, a, b in (("ctgA", None, None), ("ctgA", 1, 60), ("ctgA", 61, 61), ("ctgA", 59, 183), ("ctgA", 1200, 5000), ("b", 2, 61)):
        region = ctg if a is None else "%s:%d-%d" % (ctg, a, b)
        want = b"".join(subprocess.check_output(fake + ["faidx", fa, region]).split(b"\n")[1:])
        assert faidx(fa, ctg, a, b) == want, region
    assert faidx(fa, "missing") == b""


def test_errors_are_reported(tmp_path):
    from clairvoyante_amd import _lib
    from clairvoyante_amd.bam import BamFile
    with pytest.raises(_lib.CvError):
        BamFile(str(tmp_path / "nope.bam"))
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"this is not a BGZF file" * 10)
    with pytest.raises(_lib.CvError):
        BamFile(str(bad))
    recs = sam_records("plain")
    bam = str(tmp_path / "t.bam")
    write_bam(bam, recs, [("ctgA", 4000)], block_payload=5000)
    blob = bytearray(open(bam, "rb").read())
    blob[len(blob) // 2] ^= 0x55                         # a flipped byte inside some block: CRC / inflate failure
    open(bam, "wb").write(bytes(blob))
    os.remove(bam + ".bai")
    with pytest.raises(_lib.CvError):
        view_text(bam, "ctgA")


def view_records(path, ctg, start=None, end=None, window=65536, threads=3, mask=2308):
    """(QNAME, FLAG, POS, CIGAR, SEQ) of the records cv_bam_view_records selects, decoded here from the raw records"""
    import ctypes
    import struct
    from clairvoyante_amd import _lib
    from clairvoyante_amd.bam import BamFile
    bf = BamFile(path, threads=threads)
    lib = bf.lib
    _lib.check(lib.cv_bam_view_begin(bf.h, ctg.encode(), int(start or 0), int(end or 0), mask, 0))
    base = ctypes.c_void_p(); offs = ctypes.c_void_p(); done = ctypes.c_int(0)
    out = []
    while not done.value:
        n = lib.cv_bam_view_records(bf.h, window, ctypes.byref(base), ctypes.byref(offs), ctypes.byref(done))
        assert n >= 0, _lib.last_error() if hasattr(_lib, "last_error") else "error"
        if n == 0:
            continue
        o = np.ctypeslib.as_array(ctypes.cast(offs, ctypes.POINTER(ctypes.c_uint32)), shape=(n,)).copy()
        for off in o:
            bs = struct.unpack("<i", ctypes.string_at(base.value + int(off) - 4, 4))[0]
            r = ctypes.string_at(base.value + int(off), bs)
            _tid, pos, l_name, _mq, _bin, n_cig, flag, l_seq = struct.unpack("<iiBBHHHi", r[:20])
            name = r[32:32 + l_name - 1].decode()
            cig = struct.unpack("<%dI" % n_cig, r[32 + l_name:32 + l_name + 4 * n_cig])
            cigar = "".join("%d%s" % (c >> 4, "MIDNSHP=X"[c & 15]) for c in cig) or "*"
            sq = r[32 + l_name + 4 * n_cig:32 + l_name + 4 * n_cig + (l_seq + 1) // 2]
            seq = "".join("=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(l_seq)) or "*"
            out.append((name, flag, pos + 1, cigar, seq))
    bf.close()
    return out


@pytest.mark.parametrize("case", ["plain", "noisy", "eqx", "handmade"])
@pytest.mark.parametrize("payload,window", [(60000, 65536), (777, 65536), (4093, 1 << 20)])
def test_raw_record_view_selects_what_the_text_view_prints(case, payload, window, tmp_path):
    """cv_bam_view_records (the feed of cv_pileup_add_bam) hands out exactly the records of the text view, in order"""
    recs = sam_records(case)
    L = 3000 if case != "handmade" else 320
    bam = str(tmp_path / "a.bam")
    write_bam(bam, recs, [("ctgA", max(L, 4000)), ("other", 10)], block_payload=payload, index=True)
    for start, end in ((None, None), (1, 50), (601, 2900), (1500, 1500), (2990, 100000)):
        want = [(f[0], int(f[1]), int(f[3]), f[5], f[9]) for f in (l.split("\t") for l in expected(recs, "ctgA", start, end))]
        assert view_records(bam, "ctgA", start, end, window=window) == want
    assert view_records(bam, "nope") == [] and view_records(bam, "other") == []


def long_cigar_records(n_ops=70001, start=100):
    """a read whose CIGAR has more than 65535 operations (1M 1I 1M 1D ...), between two ordinary reads"""
    rng = np.random.RandomState(4)
    ops = []
    while len(ops) < n_ops:
        ops += ["1M", "1I", "2M", "1D"]
    ops = ops[:n_ops]
    qlen = sum(int(o[:-1]) for o in ops if o[-1] in "MI")
    seq = "".join("ACGT"[i] for i in rng.randint(0, 4, qlen))
    return ["r0\t0\tctgA\t%d\t60\t30M\t*\t0\t0\t%s\t*" % (start - 20, seq[:30]),
            "long\t0\tctgA\t%d\t60\t%s\t*\t0\t0\t%s\t*" % (start, "".join(ops), seq),
            "r2\t0\tctgA\t%d\t60\t12M3S\t*\t0\t0\t%s\t*" % (start + 500, seq[:15])]


def test_long_cigar_comes_from_the_cg_tag(tmp_path):
    """more than 65535 operations: the record holds <l_seq>S<span>N inline and the real CIGAR in CG:B,I (SAMv1 4.2.2);
    text and record views give the real one; a placeholder without the tag is an error"""
    import ctypes
    from clairvoyante_amd import _lib
    from clairvoyante_amd.bam import BamFile
    recs = long_cigar_records()
    bam = str(tmp_path / "long.bam")
    write_bam(bam, recs, [("ctgA", 200000)], block_payload=60000, index=True)
    assert view_text(bam, "ctgA") == expected(recs, "ctgA", None, None)
    assert view_text(bam, "ctgA", 50000, 50010) == expected(recs, "ctgA", 50000, 50010) and len(view_text(bam, "ctgA", 50000, 50010)) == 1
    got = view_records(bam, "ctgA", window=1 << 20)
    assert [g[0] for g in got] == ["r0", "long", "r2"] and got[1][3].endswith("N")      # raw view: the placeholder
    # the helper resolves it
    bf = BamFile(bam)
    lib = bf.lib
    lib.cv_bam_record_cigar.restype = ctypes.c_int
    lib.cv_bam_record_cigar.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
    _lib.check(lib.cv_bam_view_begin(bf.h, b"ctgA", 0, 0, 2308, 0))
    base = ctypes.c_void_p(); offs = ctypes.c_void_p(); done = ctypes.c_int(0)
    n = lib.cv_bam_view_records(bf.h, 1 << 22, ctypes.byref(base), ctypes.byref(offs), ctypes.byref(done))
    assert n == 3
    o = np.ctypeslib.as_array(ctypes.cast(offs, ctypes.POINTER(ctypes.c_uint32)), shape=(n,))
    ops = ctypes.c_void_p(); cnt = ctypes.c_int64()
    assert lib.cv_bam_record_cigar(ctypes.c_void_p(base.value + int(o[1])), ctypes.byref(ops), ctypes.byref(cnt)) == 0
    words = np.ctypeslib.as_array(ctypes.cast(ops, ctypes.POINTER(ctypes.c_uint32)), shape=(cnt.value,))
    assert "".join("%d%s" % (w >> 4, "MIDNSHP=X"[w & 15]) for w in words) == recs[1].split("\t")[5]
    bf.close()
    # strip the tags: block_size shrinks by the aux length, the placeholder stays -> rejected
    import struct
    from bam_writer import encode_record
    blob, _t, _b, _e = encode_record(recs[1].split("\t"), {"ctgA": 0})
    bs = struct.unpack("<i", blob[:4])[0]
    l_seq = struct.unpack("<i", blob[4 + 16:4 + 20])[0]
    core = 32 + blob[4 + 8] + 8 + (l_seq + 1) // 2 + l_seq
    assert core < bs
    bad = struct.pack("<i", core) + blob[4:4 + core]
    import bam_writer
    orig = bam_writer.encode_record
    try:
        bam_writer.encode_record = lambda f, t: (bad, 0, int(f[3]) - 1, int(f[3]) + 10) if f[0] == "long" else orig(f, t)
        write_bam(str(tmp_path / "bad.bam"), recs, [("ctgA", 200000)], index=False)
    finally:
        bam_writer.encode_record = orig
    bf = BamFile(str(tmp_path / "bad.bam"))
    with pytest.raises(Exception):
        b"".join(bf.view("ctgA"))
    bf.close()


def _rewrite_with_patched_stream(src_bam, dst_bam, patch):
    """inflate every BGZF block of src, let `patch(bytearray stream)` edit the BAM stream, write it back as valid
    (CRC-correct) BGZF blocks: corrupt RECORDS behind intact block checksums"""
    import struct
    import zlib
    from bam_writer import _bgzf_block, _EOF
    blob = open(src_bam, "rb").read()
    stream = bytearray()
    off = 0
    while off < len(blob):
        xlen = struct.unpack("<H", blob[off + 10:off + 12])[0]
        bsize = struct.unpack("<H", blob[off + 16:off + 18])[0] + 1
        stream += zlib.decompress(blob[off + 12 + xlen:off + bsize - 8], -15)
        off += bsize
    patch(stream)
    out = bytearray()
    for o in range(0, len(stream), 60000):
        out += _bgzf_block(bytes(stream[o:o + 60000]))
    open(dst_bam, "wb").write(bytes(out + _EOF))


def _first_record_offset(stream):
    import struct
    l_text = struct.unpack("<i", stream[4:8])[0]
    n_ref = struct.unpack("<i", stream[8 + l_text:12 + l_text])[0]
    pos = 12 + l_text
    for _ in range(n_ref):
        l_name = struct.unpack("<i", stream[pos:pos + 4])[0]
        pos += 8 + l_name
    return pos


@pytest.mark.parametrize("l_seq", [-1, -2 ** 31, -2])
def test_negative_l_seq_is_rejected_not_written_past_the_buffer(l_seq, tmp_path):
    """a record whose l_seq is negative passed the layout check, made the `worst` size estimate of
    the text view negative and let name / CIGAR text run past the caller's buffer"""
    import struct
    from clairvoyante_amd import _lib
    recs = sam_records("plain")
    # give the first record a CIGAR of many operations so that its text is long
    f = recs[0].split("\t")
    f[5] = "1M1I" * 20000 + "1M"; f[9] = "A" * 40001; f[10] = "*"
    recs = ["\t".join(f)] + recs[1:]
    good = str(tmp_path / "good.bam"); bad = str(tmp_path / "bad.bam")
    write_bam(good, recs, [("ctgA", 50000)], index=False)

    def patch(stream):
        p = _first_record_offset(stream)
        stream[p + 4 + 16:p + 4 + 20] = struct.pack("<i", l_seq)
    _rewrite_with_patched_stream(good, bad, patch)
    with pytest.raises(_lib.CvError, match="corrupt record"):
        view_text(bad, "ctgA")
    with pytest.raises((_lib.CvError, AssertionError)):
        view_records(bad, "ctgA")


def test_bsize_smaller_than_its_own_header_is_rejected(tmp_path):
    """BSIZE < header + trailer made the trailer reads land before the block"""
    import struct
    from clairvoyante_amd import _lib
    from clairvoyante_amd.bam import BamFile
    bam = str(tmp_path / "t.bam")
    write_bam(bam, sam_records("plain"), [("ctgA", 4000)], index=False)
    blob = bytearray(open(bam, "rb").read())
    blob[16:18] = struct.pack("<H", 2)                   # BSIZE = 3 bytes in total
    open(bam, "wb").write(bytes(blob))
    with pytest.raises(_lib.CvError):
        BamFile(bam)
