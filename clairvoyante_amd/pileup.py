"""Host handle of the on-GPU pileup (include/clairvoyante_amd.h, "pileup front end"): SAM text in,
[n,33,4,4] count tensors in HBM out.  Behaviour of /root/reference/dataPrepScripts/CreateTensor.py
(OutputAlnTensor :93-246, GenerateTensor :23-54); see csrc/cv_pileup.hip for the decomposition.
"""
import ctypes

import numpy as np

from . import _lib

FLANK = 16
WIDTH = 2 * FLANK + 1
FLUSH_COLUMNS = 1 << 26         # queue at most this many alignment columns on the host before a scatter launch


class Pileup(object):
    def __init__(self, device=None, minMQ=0, dcov=250, considerleftedge=True, evc=False, retain=False, evc_minMQ=0,
                 contig=None, threads=None):
        import torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.CvError("the pileup kernels need an MI355X (no GPU visible); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.cv_pileup_create(self.device.index, int(minMQ), int(dcov), int(bool(considerleftedge)),
                                             ctypes.byref(self.h)))
        if evc:
            _lib.check(self.lib.cv_pileup_set_option(self.h, b"evc", 1))
            _lib.check(self.lib.cv_pileup_set_option(self.h, b"evc_min_mq", int(evc_minMQ)))
            if contig is not None:
                _lib.check(self.lib.cv_pileup_set_contig(self.h, contig.encode()))
        if retain:
            _lib.check(self.lib.cv_pileup_set_option(self.h, b"retain", 1))
        # SAM text is parsed by several host threads (chunks of >= 1 MiB); the result does not depend on the count
        self.threads = min(_lib.usable_cores(), 16) if threads is None else int(threads)
        _lib.check(self.lib.cv_pileup_set_option(self.h, b"threads", self.threads))
        self.n = 0
        self.centers = np.zeros(0, dtype=np.int64)
        self._tail = b""
        self.reads_kept = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.cv_pileup_destroy(self.h)
            self.h = None

    __del__ = close

    def _stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_reference(self, seq, first_pos0=0):
        """seq: the bases `samtools faidx` printed (str/bytes); seq[0] is 0-based position first_pos0"""
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        _lib.check(self.lib.cv_pileup_set_reference(self.h, b, len(b), int(first_pos0)))

    def set_candidates(self, centers):
        """1-based candidate positions; sorted and de-duplicated here"""
        c = np.unique(np.asarray(centers, dtype=np.int64))
        self.centers = np.ascontiguousarray(c)
        self.n = len(c)
        _lib.check(self.lib.cv_pileup_set_candidates(self.h, self.centers.ctypes.data_as(ctypes.c_void_p), self.n))

    def _feed(self, data, off, final):
        """parse data[off:] in place (no copy); returns the bytes consumed"""
        n = len(data) - off
        if n <= 0:
            return 0
        consumed = ctypes.c_int64(0)
        kept = ctypes.c_int64(0)
        base = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value        # data stays referenced by the caller
        _lib.check(self.lib.cv_pileup_add_sam(self.h, ctypes.c_void_p(base + off), n, int(final), ctypes.byref(consumed),
                                              ctypes.byref(kept)))
        self.reads_kept += kept.value
        self._kept_now += kept.value
        return consumed.value

    def add_sam(self, chunk, final=False):
        """feed SAM text (bytes) in arbitrary chunks; an incomplete last line is kept for the next call"""
        self._kept_now = 0
        off = 0
        if self._tail:
            nl = chunk.find(b"\n")
            if nl < 0 and not final:
                self._tail += chunk
                return 0
            head = self._tail + (chunk if nl < 0 else chunk[:nl + 1])           # one line: the only bytes copied
            self._tail = b""
            self._feed(head, 0, final and nl < 0)
            off = len(chunk) if nl < 0 else nl + 1
        off += self._feed(chunk, off, final)
        if off < len(chunk):
            self._tail = chunk[off:]
        if self.lib.cv_pileup_pending(self.h) >= FLUSH_COLUMNS:
            _lib.check(self.lib.cv_pileup_flush(self.h, self._stream()))
        return self._kept_now

    def add_bam(self, bam, ref, start=None, end=None, exclude_flags=2308, contig_ok=True, window=64 << 20):
        """feed the records `samtools view -F exclude_flags BAM ref[:start-end]` would print, straight from the
        BAM (bam: clairvoyante_amd.bam.BamFile) -- no SAM text in between; same result as add_sam on that text"""
        self._kept_now = 0
        if self._tail:
            self.add_sam(b"", final=True)
        _lib.check(self.lib.cv_bam_view_begin(bam.h, ref.encode(), int(start or 0), int(end or 0), int(exclude_flags), 0))
        base = ctypes.c_void_p(); offs = ctypes.c_void_p(); done = ctypes.c_int(0); kept = ctypes.c_int64(0)
        total = 0
        while not done.value:
            n = self.lib.cv_bam_view_records(bam.h, int(window), ctypes.byref(base), ctypes.byref(offs), ctypes.byref(done))
            if n < 0:
                _lib.check(1)
            if n:
                _lib.check(self.lib.cv_pileup_add_bam(self.h, base, offs, n, int(bool(contig_ok)), ctypes.byref(kept)))
                self.reads_kept += kept.value
                total += kept.value
                if self.lib.cv_pileup_pending(self.h) >= FLUSH_COLUMNS:
                    _lib.check(self.lib.cv_pileup_flush(self.h, self._stream()))
        return total

    def finish(self, subtract=False, want_tensors=True):
        """-> (tensors [n,33,4,4] fp32 on the device, depth [n] int32, touched [n] bool)"""
        import torch
        if self._tail:
            self.add_sam(b"", final=True)
        t = torch.empty((self.n, WIDTH, 4, 4), dtype=torch.float32, device=self.device) if want_tensors else None
        d = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        u = torch.empty((self.n,), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.cv_pileup_finish(self.h, ctypes.c_void_p(t.data_ptr()) if want_tensors and self.n else None,
                                             ctypes.c_void_p(d.data_ptr()) if self.n else None,
                                             ctypes.c_void_p(u.data_ptr()) if self.n else None, int(bool(subtract)),
                                             self._stream()))
        return t, d, u.bool()

    def extract_candidates(self, threshold=0.125, minCoverage=4, region=None, bed=None):
        """ExtractVariantCandidates.py's selection over the reads added so far (needs evc=True).
        region: (ctgStart, ctgEnd) as the reference compares them with the 0-based position (:181-183);
        bed: list of half-open (begin, end), or None.  -> dict(pos0, late, counts [n,7] in A,C,G,T,I,D,N
        order, reads)"""
        if self._tail:
            self.add_sam(b"", final=True)
        n = ctypes.c_int64(0)
        if bed is not None:
            bb = np.ascontiguousarray([b for b, _ in bed], dtype=np.int64)
            be = np.ascontiguousarray([e for _, e in bed], dtype=np.int64)
            nbed = len(bed)
        else:
            bb = be = np.zeros(1, dtype=np.int64)
            nbed = -1
        _lib.check(self.lib.cv_pileup_extract_candidates(
            self.h, float(threshold), float(minCoverage), int(region is not None), int(region[0]) if region else 0,
            int(region[1]) if region else 0, bb.ctypes.data_as(ctypes.c_void_p), be.ctypes.data_as(ctypes.c_void_p),
            nbed, self._stream(), ctypes.byref(n)))
        k = n.value
        pos0 = np.zeros(k, dtype=np.int64); late = np.zeros(k, dtype=np.int32); c7 = np.zeros((k, 7), dtype=np.int32)
        info = (ctypes.c_int64 * 2)()
        _lib.check(self.lib.cv_pileup_get_extracted(self.h, pos0.ctypes.data_as(ctypes.c_void_p),
                                                    late.ctypes.data_as(ctypes.c_void_p),
                                                    c7.ctypes.data_as(ctypes.c_void_p), info))
        return {"pos0": pos0, "late": late, "counts": c7, "reads": info[0], "last_pos": info[1]}

    def adopt_candidates(self, lo1=None, hi1=None):
        """make the extracted positions (+1, optionally inside [lo1, hi1]) the candidate centres and scatter
        the retained alignments for them (needs retain=True)"""
        n = ctypes.c_int64(0)
        _lib.check(self.lib.cv_pileup_adopt_candidates(self.h, int(lo1 is not None), int(lo1 or 0), int(hi1 or 0),
                                                       self._stream(), ctypes.byref(n)))
        self.n = n.value
        self.centers = np.zeros(self.n, dtype=np.int64)
        _lib.check(self.lib.cv_pileup_get_candidates(self.h, self.centers.ctypes.data_as(ctypes.c_void_p), self.n,
                                                     ctypes.byref(n)))
        return self.centers

    def stats(self):
        ms = (ctypes.c_float * 3)()
        cnt = (ctypes.c_int64 * 3)()
        _lib.check(self.lib.cv_pileup_stats(self.h, ms, cnt))
        return {"scatter_ms": ms[0], "finalize_ms": ms[1], "candidate_ms": ms[2], "columns": cnt[0], "segments": cnt[1],
                "launches": cnt[2]}


def format_rows(ctg, centers, ref_seq, ref_shift, counts):
    """CreateTensor.py:50-52 text rows for host arrays: centers [k] (1-based), counts [k,33,4,4] raw.
    ref_shift = refStart-1 (0 when the whole contig was loaded)."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64 + len(ctg) + WIDTH + WIDTH * 16 * 16)
    cb = ctg.encode()
    rb = ref_seq.encode() if isinstance(ref_seq, str) else ref_seq
    counts = np.ascontiguousarray(counts, dtype=np.float32)
    rows = []
    for k, c in enumerate(centers):
        new_pos = int(c) - ref_shift
        seq = rb[new_pos - (FLANK + 1):new_pos + FLANK]
        n = lib.cv_format_tensor_row(cb, int(c), seq, len(seq), counts[k].ctypes.data_as(ctypes.c_void_p), buf, len(buf))
        if n < 0:
            raise _lib.CvError("cv_format_tensor_row: buffer too small")
        rows.append(buf.raw[:n])
    return rows
