"""Native producer of the alignments text (optional: `--samtools native`): what `samtools view -F 2308 BAM CTG[:S-E]`
and `samtools faidx REF CTG[:S-E]` print, read directly from the BAM (+ .bai) and the FASTA (+ .fai) -- no external
process, BGZF inflated by several host threads (csrc/cv_bam.cpp).  The default producer remains the samtools pipe
of the reference (dataPrepScripts/CreateTensor.py:101-130)."""
import ctypes
import os

from . import _lib

NATIVE = "native"


class BamFile(object):
    def __init__(self, path, threads=None):
        self.lib = _lib.load()
        self.h = ctypes.c_void_p()
        t = min(_lib.usable_cores(), 16) if threads is None else int(threads)
        _lib.check(self.lib.cv_bam_open(os.fsencode(path), t, ctypes.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cv_bam_close(self.h)
            self.h = None

    __del__ = close

    def references(self):
        out = []
        for i in range(self.lib.cv_bam_nref(self.h)):
            name = ctypes.c_char_p(); ln = ctypes.c_int64()
            _lib.check(self.lib.cv_bam_ref(self.h, i, ctypes.byref(name), ctypes.byref(ln)))
            out.append((name.value.decode(), ln.value))
        return out

    def has_index(self):
        return bool(self.lib.cv_bam_has_index(self.h))

    def view(self, ref, start=None, end=None, exclude_flags=2308, with_qual=False, chunk=8 << 20):
        """generator of bytes: whole SAM lines of the records of `ref` overlapping [start, end] (1-based inclusive)"""
        _lib.check(self.lib.cv_bam_view_begin(self.h, ref.encode(), int(start or 0), int(end or 0), int(exclude_flags),
                                              int(bool(with_qual))))
        buf = ctypes.create_string_buffer(chunk)
        done = ctypes.c_int(0)
        while not done.value:
            n = self.lib.cv_bam_view_read(self.h, buf, chunk, ctypes.byref(done))
            if n < 0:
                _lib.check(1)
            if n:
                yield buf.raw[:n]


def faidx(ref_fn, ctg, start=None, end=None):
    """the bases `samtools faidx REF CTG[:S-E]` prints (1-based inclusive), through REF.fai"""
    with open(ref_fn + ".fai") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if f[0] == ctg:
                length, offset, linebases, linewidth = int(f[1]), int(f[2]), int(f[3]), int(f[4])
                break
        else:
            return b""
    s = 1 if start is None else max(int(start), 1)
    e = length if end is None else min(int(end), length)
    if e < s:
        return b""
    b0, b1 = s - 1, e
    with open(ref_fn, "rb") as fh:
        fh.seek(offset + b0 // linebases * linewidth + b0 % linebases)
        raw = fh.read((b1 // linebases - b0 // linebases) * linewidth + (b1 % linebases) - (b0 % linebases) + linewidth)
    seq = raw.replace(b"\n", b"").replace(b"\r", b"")
    return seq[:b1 - b0]
