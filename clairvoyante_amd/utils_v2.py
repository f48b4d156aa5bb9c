"""Data plane of the hot path: same functions, arguments and return values as
/root/reference/clairvoyante/utils_v2.py (SetupEnv :14, GetTensor :23-59,
GetTrainingArray :62-186, DecompressArray :189-207), with the per-row tokenising and
the blosc codec done in native code (csrc/cv_hostio.cpp) instead of CPython / python-blosc.
"""
import ctypes
import gc
import gzip
import io
import os
import pickle
import random
import shlex
import subprocess
import sys

import numpy as np

from . import _lib
from . import param

base2num = dict(zip("ACGT", (0, 1, 2, 3)))
_NV = (2 * param.flankingBaseNum + 1) * 4 * param.matrixNum


def SetupEnv():
    """utils_v2.py:14-18 (CXX / TF log level / blosc threads have no meaning here)."""
    os.environ["CXX"] = "g++"
    gc.enable()


class PosBatch(object):
    """The `pos` list of a GetTensor batch ("chrom:coord:seq", utils_v2.py:41), built lazily: at GPU rates only the
    few candidates that become VCF records ever need their string, and the native VCF formatter (cv_format_vcf)
    reads the fields where the parser found them.  A batch is one or more pieces (bytes, meta [rows,6] int64 =
    offset / length of contig, position, sequence inside those bytes), in row order."""

    def __init__(self, buf=None, meta=None, pieces=None):
        self._pieces = list(pieces) if pieces is not None else [(buf, meta)]
        self._starts = np.cumsum([0] + [m.shape[0] for _b, m in self._pieces])

    def __len__(self):
        return int(self._starts[-1])

    def pieces(self):
        """-> [(first row, rows, bytes, meta)]"""
        return [(int(self._starts[k]), m.shape[0], b, m) for k, (b, m) in enumerate(self._pieces)]

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[i] for i in range(*j.indices(len(self)))]
        if j < 0:
            j += len(self)
        k = int(np.searchsorted(self._starts, j, side="right")) - 1
        b, meta = self._pieces[k]
        m = meta[j - int(self._starts[k])]
        return (bytes(b[m[0]:m[0] + m[1]]) + b":" + bytes(b[m[2]:m[2] + m[3]]) + b":" + bytes(b[m[4]:m[4] + m[5]]).upper()).decode("ascii")

    def __iter__(self):
        for j in range(len(self)):
            yield self[j]

    @staticmethod
    def from_columns(chrom, coords, seqs):
        """one contig name, integer coordinates and the reference sequences (bytes) of n candidates"""
        chrom = chrom if isinstance(chrom, bytes) else str(chrom).encode("ascii")
        n = len(coords)
        cs = [b"%d" % int(c) for c in coords]
        clen = np.fromiter((len(c) for c in cs), dtype=np.int64, count=n)
        slen = np.fromiter((len(q) for q in seqs), dtype=np.int64, count=n)
        meta = np.empty((n, 6), dtype=np.int64)
        meta[:, 0] = 0; meta[:, 1] = len(chrom)
        c0 = len(chrom)
        meta[:, 2] = c0 + np.concatenate(([0], np.cumsum(clen)[:-1])) if n else 0
        meta[:, 3] = clen
        s0 = c0 + int(clen.sum())
        meta[:, 4] = s0 + np.concatenate(([0], np.cumsum(slen)[:-1])) if n else 0
        meta[:, 5] = slen
        return PosBatch(chrom + b"".join(cs) + b"".join(seqs), meta)

    @staticmethod
    def from_strings(pos):
        """the same container from "chrom:coord:seq" strings (callers that hold Python strings)"""
        n = len(pos)
        meta = np.empty((n, 6), dtype=np.int64)
        parts, off = [], 0
        for i, p in enumerate(pos):
            f = (p if isinstance(p, bytes) else str(p).encode("ascii")).split(b":")
            if len(f) != 3:
                raise ValueError("position %r is not chrom:coord:seq" % (p,))
            for k in range(3):
                meta[i, 2 * k] = off; meta[i, 2 * k + 1] = len(f[k])
                parts.append(f[k]); off += len(f[k])
        return PosBatch(b"".join(parts), meta)


class _GzipFile(object):
    """A gzip file inflated by the library's own DEFLATE decoder (csrc/cv_inflate.cpp, ~2x the rate of `gzip -dc`, no
    child process, no pipe), for regular files: the file is memory-mapped, every member's header is skipped by hand
    (RFC 1952), the data is decoded block by block into a window buffer (cv_inflate_stream) and each member's CRC-32 and
    length are checked.  read(n) hands out what has been inflated, like the pipe of the gzip process it replaces.
    Anything unexpected -- not a regular gzip file, a block that does not fit, a failed check -- raises _GzipFallback
    BEFORE any byte has been handed out, or CvError after (the stream would otherwise be silently short)."""
    WINDOW = 32768
    WANT = 24 << 20                 # new bytes per decoder call
    CAP = WINDOW + WANT + (40 << 20)      # room for the block that crosses WANT

    def __init__(self, fn):
        import mmap
        self.lib = _lib.load()
        with open(fn, "rb") as fh:
            self.mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        if hasattr(self.mm, "madvise") and hasattr(mmap, "MADV_SEQUENTIAL"):
            self.mm.madvise(mmap.MADV_SEQUENTIAL)
        self.src = np.frombuffer(self.mm, dtype=np.uint8)
        self.n = int(self.src.shape[0])
        self.buf = np.empty(self.CAP, dtype=np.uint8)
        self.have = 0               # bytes of history in front of the fresh output
        self.lo = self.hi = 0       # fresh output not yet handed out: buf[lo:hi]
        self.pos = 0                # byte offset of the current member's DEFLATE data in the file
        self.bitpos = ctypes.c_int64(0)
        self.final = ctypes.c_int32(0)
        self.crc = 0
        self.size = 0
        self.handed = 0
        self.eof = False
        self.fn = fn
        self.fresh_member = True
        self._member_header()

    def _fail(self, what):
        msg = "%s: %s" % (self.fn, what)
        if self.handed == 0:
            raise _GzipFallback(msg)
        raise _lib.CvError("gzip stream broke off after %d bytes: %s" % (self.handed, msg))

    def _member_header(self):
        b, p = self.src, self.pos
        if p + 18 > self.n or b[p] != 0x1f or b[p + 1] != 0x8b or b[p + 2] != 8 or (b[p + 3] & 0xe0):
            self._fail("not a gzip member at byte %d" % p)
        flg = int(b[p + 3])
        p += 10
        if flg & 4:                                           # FEXTRA
            p += 2 + int(b[p]) + 256 * int(b[p + 1])
        for bit in (8, 16):                                   # FNAME, FCOMMENT: zero-terminated
            if flg & bit:
                while p < self.n and b[p] != 0:
                    p += 1
                p += 1
        if flg & 2:                                           # FHCRC
            p += 2
        if p + 8 > self.n:
            self._fail("truncated header")
        self.pos = p
        self.bitpos.value = 0
        self.crc, self.size = 0, 0
        self.fresh_member = True                              # its matches never reach into the member before it

    def _more(self):
        """inflate the next piece into the window buffer (called when everything inflated so far has been handed out);
        False at the end of the file"""
        if self.eof:
            return False
        if self.fresh_member:
            self.have, self.fresh_member = 0, False
        else:                                                 # the last WINDOW bytes of the output stay as history at the front
            tot = self.hi
            keep = min(tot, self.WINDOW)
            if tot > keep:
                self.buf[:keep] = self.buf[tot - keep:tot].copy()
            self.have = keep
        avail = self.n - 8 - self.pos                         # DEFLATE data ends at least 8 bytes (a trailer) before the end of the file
        got = self.lib.cv_inflate_stream(ctypes.c_void_p(self.src.ctypes.data + self.pos), avail, ctypes.byref(self.bitpos),
                                         ctypes.c_void_p(self.buf.ctypes.data), self.have, self.CAP, self.WANT,
                                         ctypes.byref(self.final))
        if got < 0:
            self._fail("malformed DEFLATE data (or a block larger than the window buffer)")
        got = int(got)
        self.lo, self.hi = self.have, self.have + got
        if got:
            self.crc = self.lib.cv_crc32_ieee(self.crc, ctypes.c_void_p(self.buf.ctypes.data + self.lo), got)
            self.size += got
        if self.final.value:
            t = self.pos + ((self.bitpos.value + 7) >> 3)     # trailer: CRC-32, ISIZE (mod 2^32), little endian
            want_crc = int.from_bytes(bytes(self.src[t:t + 4]), "little")
            want_len = int.from_bytes(bytes(self.src[t + 4:t + 8]), "little")
            if want_crc != self.crc or want_len != (self.size & 0xffffffff):
                self._fail("CRC-32 / length of a member do not match its trailer")
            self.pos = t + 8
            while self.pos < self.n and self.src[self.pos] == 0:      # zero padding behind the last member is legal
                self.pos += 1
            if self.pos >= self.n:
                self.eof = True
            else:
                self._member_header()
        return True

    def read(self, n=-1):
        out = []
        need = n if n is not None and n >= 0 else 1 << 62
        while need > 0:
            if self.lo == self.hi:
                if not self._more():
                    break
                continue
            k = min(need, self.hi - self.lo)
            out.append(self.buf[self.lo:self.lo + k].tobytes())
            self.lo += k; need -= k; self.handed += k
        return out[0] if len(out) == 1 else b"".join(out)

    def close(self):
        self.src = None
        try:
            self.mm.close()
        except (BufferError, ValueError):
            pass


class _GzipFallback(Exception):
    pass


def _open_tensor_stream(tensor_fn):
    """-> (child process or None, file object with read(n)): the reference's `gzip -fdc FILE` pipe (utils_v2.py:25), or
    for a regular file the in-process decoder (which hands the file over to that pipe if it meets anything it cannot
    decode or vouch for); CV_GZIP=external forces the child process."""
    if tensor_fn != "PIPE":
        if os.environ.get("CV_GZIP") != "external" and os.path.isfile(tensor_fn):
            return None, _GzipOrPipe(tensor_fn)
        f = subprocess.Popen(shlex.split("gzip -fdc %s" % (tensor_fn)), stdout=subprocess.PIPE, bufsize=8388608)
        return f, f.stdout
    return None, sys.stdin.buffer


class _GzipOrPipe(object):
    """read(n) over a tensor file: the in-process decoder while it is sure of itself; at its first doubt -- a file that is
    not gzip (the reference's `gzip -fdc` also passes plain text and .Z through), a construct it does not take, a member
    whose check fails -- the reference's own `gzip -fdc` child process takes over FROM THE BYTE the caller has reached
    (what was handed out already is read from the pipe and dropped), so the caller sees exactly the reference's stream,
    and an error (exit status 1, raised by close()) where the reference's decompressor reports one."""

    def __init__(self, fn):
        self.fn, self.g, self.proc, self.out = fn, None, None, 0
        try:
            self.g = _GzipFile(fn)
        except (_GzipFallback, OSError, ValueError, IndexError):
            self._to_pipe()

    def _to_pipe(self):
        if self.g is not None:
            self.g.close()
            self.g = None
        self.proc = subprocess.Popen(shlex.split("gzip -fdc %s" % (self.fn)), stdout=subprocess.PIPE, bufsize=8388608)
        skip = self.out
        while skip > 0:
            c = self.proc.stdout.read(min(skip, 1 << 24))
            if not c:
                break
            skip -= len(c)

    def read(self, n=-1):
        if self.g is not None:
            try:
                c = self.g.read(n)
                self.out += len(c)
                return c
            except (_GzipFallback, _lib.CvError, OSError, ValueError, IndexError):
                self._to_pipe()
        c = self.proc.stdout.read(n)
        self.out += len(c)
        return c

    def close(self):
        if self.g is not None:
            self.g.close()
            self.g = None
        if self.proc is not None:
            self.proc.stdout.close()
            rc = self.proc.wait()
            self.proc = None
            if rc == 1:
                raise _lib.CvError("gzip -fdc %s failed (exit status 1): the tensor stream is incomplete" % self.fn)


def _close_tensor_stream(proc, fo, tensor_fn):
    if proc is not None:
        fo.close()
        if proc.wait() == 1:          # gzip: 1 = error (missing / unreadable / corrupt file), 2 = warning.  The reference
            # reads on with whatever arrived (utils_v2.py:25 never looks at the exit status): a truncated call set
            raise _lib.CvError("gzip -fdc %s failed (exit status 1): the tensor stream is incomplete" % tensor_fn)
    elif fo is not sys.stdin.buffer:
        fo.close()


def _close_quietly_unless(done, proc, fo, tensor_fn):
    """the `finally` of the stream generators: a generator that ran to its end closes its stream the loud way (a failed
    `gzip -fdc` raises); one that is dropped early -- the consumer stopped, an error is already on its way up -- still
    closes it (the child process is waited for, the window buffer and the map are released) but adds no second error"""
    if done:
        _close_tensor_stream(proc, fo, tensor_fn)
        return
    try:
        _close_tensor_stream(proc, fo, tensor_fn)
    except Exception:
        pass


def owned_line_blocks(fo, rank, ws, block_lines):
    """Split a text stream into blocks of `block_lines` lines and yield (block index, bytes) for the blocks
    rank `rank` of `ws` owns (block k belongs to rank k % ws): the sharding of callVar under torchrun.  Every rank
    reads (decompresses) the whole stream but only parses its own blocks; a last line without newline gets one."""
    block, have = 0, 0                      # current block index, lines of it seen so far
    parts = []                              # pieces of the current block if it is ours
    while True:
        chunk = fo.read(1 << 24)
        if not chunk:
            break
        nl = np.flatnonzero(np.frombuffer(chunk, dtype=np.uint8) == 10)
        start, used = 0, 0                  # byte offset / newlines of the chunk consumed
        while used < len(nl):
            need = block_lines - have
            if len(nl) - used >= need:      # the block ends inside this chunk
                end = int(nl[used + need - 1]) + 1
                if block % ws == rank:
                    parts.append(chunk[start:end])
                    yield block, b"".join(parts)
                parts = []
                block += 1; have = 0
                start = end; used += need
            else:
                have += len(nl) - used
                used = len(nl)
        if start < len(chunk) and block % ws == rank:
            parts.append(chunk[start:])
    if parts or have:
        tail = b"".join(parts)
        if block % ws == rank and tail:
            yield block, tail if tail.endswith(b"\n") else tail + b"\n"


def GetTensorBlocks(tensor_fn, block_lines, rank, ws):
    """Sharded form of GetTensor (callVar under torchrun): yields (block index, c, X, pos) for every block of
    `block_lines` input lines this rank owns -- one batch per block, rows as GetTensor makes them."""
    if tensor_fn == "PIPE":
        raise ValueError("--tensor_fn PIPE cannot be sharded over ranks: give the tensor file")
    lib = _lib.load()
    proc, fo = _open_tensor_stream(tensor_fn)
    consumed = ctypes.c_int64(); nrows = ctypes.c_int64(); nbad = ctypes.c_int64()
    done = False
    try:
        for block, data in owned_line_blocks(fo, rank, ws, block_lines):
            rows = _pinned.empty((block_lines, _NV), np.float32)
            meta = np.empty((block_lines, 6), dtype=np.int64)
            c, off, bufs = 0, 0, []
            while off < len(data):
                view = data[off:] if off else data
                _lib.check(lib.cv_parse_tensor_text(view, len(view), block_lines - c,
                                                    rows[c:].ctypes.data_as(ctypes.c_void_p),
                                                    meta[c:].ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.byref(consumed), ctypes.byref(nrows), ctypes.byref(nbad)))
                if nbad.value:
                    print("UnpackATensorRecord Failure (%d malformed rows skipped)" % nbad.value, file=sys.stderr)
                if nrows.value:
                    bufs.append((view, meta[c:c + nrows.value].copy()))
                c += nrows.value
                off += consumed.value
                if consumed.value == 0:
                    break
            yield block, c, rows[:c].reshape((c, 2 * param.flankingBaseNum + 1, 4, param.matrixNum)), _join_pos(bufs)
        done = True
    finally:
        _close_quietly_unless(done, proc, fo, tensor_fn)


def _default_readers(nfiles):
    """concurrent file readers of one process: a compressed tensor file arrives at ONE core's inflate rate (~0.15 M
    rows/s), so several files are inflated side by side (each `gzip -dc` is its own process) while the parser threads
    serve whichever batch is complete; bounded by the cores this rank may use and by 8"""
    lws = max(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))), 1)
    return max(1, min(nfiles, _lib.usable_cores() // (2 * lws), 8))


def GetTensorFiles(files, num, rank, ws, readers=None, depth=2, ordered=True):
    """One tensor file per chunk of the genome (the reference's own recipe is one callVarBam / callVar job per chunk,
    README.md:184-202): file k belongs to rank k % ws -- a rank only ever opens (and inflates) its own files.  Yields
    (file index, c, X, pos) batches of <= num rows, at least one per owned file.
    Compressed files arrive at ONE core's inflate rate each, so up to `readers` of the rank's files are inflated and
    parsed concurrently by reader threads (started in list order, `depth` finished batches per reader in flight):
      ordered=True   the consumer sees file after file, batch after batch -- the sequence of reading them one by one
                     (a reader that runs ahead waits with `depth` batches until its file's turn comes);
      ordered=False  batches are handed over as they complete, whichever file they belong to (the batches of ONE file
                     still in order), and (file index, None, None, None) follows the last batch of a file: for a
                     consumer that keeps the per-file results apart and joins them in list order itself (callVar) --
                     all readers stay busy, `readers` x the rate of one file.
    An error in a reader is raised in the consumer."""
    import threading
    from queue import Queue, Full
    owned = [(k, fn) for k, fn in enumerate(files) if k % ws == rank]
    if readers is None:
        compressed = any(is_compressed(fn) for _k, fn in owned)       # (magic bytes only: nothing is mapped here)
        readers = _default_readers(len(owned)) if compressed else 1
    if readers <= 1 or len(owned) <= 1:
        for k, fn in owned:
            for _end, c, X, pos in GetTensor(fn, num, log=False):
                yield k, c, X, pos
            if not ordered:
                yield k, None, None, None
        return
    queues = [Queue(maxsize=depth) for _ in owned] if ordered else [Queue(maxsize=depth * readers)] * len(owned)
    slots = threading.Semaphore(readers)
    stop = threading.Event()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except Full:                                   # the consumer is busy elsewhere: look at `stop` and wait on
                continue
        return False

    def read(i):
        q = queues[i]
        gen = GetTensor(owned[i][1], num, log=False)
        try:
            for _end, c, X, pos in gen:
                if not put(q, (i, c, X, pos)):
                    return
            put(q, (i, None, None, None))
        except BaseException as e:                         # surfaced in the consumer
            put(q, (i, e, None, None))
        finally:
            gen.close()                                    # a reader told to stop still closes its stream (child process, map)
            slots.release()

    def launch():
        for i in range(len(owned)):                        # files start in list order, `readers` at a time
            slots.acquire()
            if stop.is_set():
                slots.release()
                return
            threading.Thread(target=read, args=(i,), daemon=True).start()

    threading.Thread(target=launch, daemon=True).start()
    try:
        if ordered:
            for i, (k, _fn) in enumerate(owned):
                while True:
                    _i, c, X, pos = queues[i].get()
                    if c is None:
                        break
                    if isinstance(c, BaseException):
                        raise c
                    yield k, c, X, pos
        else:
            left = len(owned)
            while left:
                i, c, X, pos = queues[0].get()
                if isinstance(c, BaseException):
                    raise c
                if c is None:
                    left -= 1
                yield owned[i][0], c, X, pos
    finally:
        stop.set()


def _gzip_would_decode(head):
    """first bytes of a file `gzip -fdc` would DEcompress rather than pass through: gzip (1f 8b), compress .Z (1f 9d),
    pack (1f 1e), lzh (1f a0) and a zip local header -- everything else it copies unchanged (gzip -f)"""
    return head[:2] in (b"\x1f\x8b", b"\x1f\x9d", b"\x1f\x1e", b"\x1f\xa0") or head[:4] == b"PK\x03\x04"


def is_compressed(tensor_fn):
    """the file is in a format the reference's `gzip -fdc` pipe (utils_v2.py:25) decompresses"""
    try:
        with open(tensor_fn, "rb") as fh:
            return _gzip_would_decode(fh.read(4))
    except OSError:
        return False


def _map_plain_text(tensor_fn):
    """-> read-only uint8 array over the memory-mapped file when `tensor_fn` is a regular, non-empty file that the
    reference's `gzip -fdc` pipe (utils_v2.py:25) would pass through unchanged, i.e. that does not start with the magic
    of a format gzip decodes (gzip, compress, pack, lzh, zip); None otherwise (PIPE, compressed files, FIFOs: the stream
    path, where _GzipOrPipe hands everything but gzip to the pipe).  CV_TEXT=stream forces the stream path for plain
    files too.  NOTE: the position fields of a batch stay views of the map until its VCF records are formatted -- a file
    that is truncated or rewritten while callVar reads it ends the process with SIGBUS instead of an exception (the
    stream path copies; use CV_TEXT=stream for inputs another process is still writing)."""
    import mmap
    import stat
    if tensor_fn == "PIPE" or os.environ.get("CV_TEXT") == "stream":
        return None
    try:
        st = os.stat(tensor_fn)
        if not stat.S_ISREG(st.st_mode) or st.st_size == 0:
            return None
        with open(tensor_fn, "rb") as fh:
            if _gzip_would_decode(fh.read(4)):
                return None
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    except OSError:
        return None
    if hasattr(mm, "madvise") and hasattr(mmap, "MADV_SEQUENTIAL"):
        mm.madvise(mmap.MADV_SEQUENTIAL)
    return np.frombuffer(mm, dtype=np.uint8)


def _get_tensor_mapped(data, num, log):
    """GetTensor over a memory-mapped plain-text file: the parser threads read the lines where the page cache holds them
    (no pipe, no copies; cv_parse_tensor_text sizes its own window), the position fields of a batch stay views of the map."""
    lib = _lib.load()
    n = int(data.shape[0])
    base = data.ctypes.data
    shape = (2 * param.flankingBaseNum + 1, 4, param.matrixNum)
    consumed = ctypes.c_int64(); nrows = ctypes.c_int64(); nbad = ctypes.c_int64()
    total, off = 0, 0
    tail = None                      # a last line without newline: parsed from a copy that has one
    if data[n - 1] != 10:
        last_nl = n - 1
        while last_nl >= 0 and data[last_nl] != 10:
            last_nl -= 1
        tail = bytes(data[last_nl + 1:]) + b"\n"
        n = last_nl + 1
    rows = _pinned.empty((num, _NV), np.float32)
    meta = np.empty((num, 6), dtype=np.int64)
    c, bufs = 0, []

    def parse(ptr, length, view):
        nonlocal c
        _lib.check(lib.cv_parse_tensor_text(ctypes.c_void_p(ptr), length, num - c,
                                            rows[c:].ctypes.data_as(ctypes.c_void_p), meta[c:].ctypes.data_as(ctypes.c_void_p),
                                            ctypes.byref(consumed), ctypes.byref(nrows), ctypes.byref(nbad)))
        if nbad.value:
            print("UnpackATensorRecord Failure (%d malformed rows skipped)" % nbad.value, file=sys.stderr)
        if nrows.value:
            bufs.append((view(consumed.value), meta[c:c + nrows.value].copy()))
        c += nrows.value
        return consumed.value

    while off < n or tail is not None:
        if off < n:
            o = off
            used = parse(base + off, n - off, lambda k: data[o:o + k])
            off += used
            if used == 0:
                off = n
        else:
            keep = np.frombuffer(tail, dtype=np.uint8)
            parse(keep.ctypes.data, len(tail), lambda k: keep[:k])
            tail = None
        if c == num:
            total += c
            if log:
                print("Processed %d tensors" % total, file=sys.stderr)
            yield 0, c, rows.reshape((num,) + shape), _join_pos(bufs)
            rows = _pinned.empty((num, _NV), np.float32)
            meta = np.empty((num, 6), dtype=np.int64)
            c, bufs = 0, []
    total += c
    if log:
        print("Processed %d tensors" % total, file=sys.stderr)
    yield 1, c, rows[:c].reshape((c,) + shape), _join_pos(bufs)


def GetTensor(tensor_fn, num, log=True):
    """Generator over batches of `num` candidates: yields (endFlag, c, X, pos) exactly like
    utils_v2.py:23-59 -- X [c,33,4,4] fp32 with matrices 1..3 minus matrix 0, rows whose
    centre base is not ACGT dropped, a final (possibly empty) batch with endFlag 1."""
    mapped = _map_plain_text(tensor_fn)
    if mapped is not None:
        for item in _get_tensor_mapped(mapped, num, log):
            yield item
        return
    lib = _lib.load()
    proc, fo = _open_tensor_stream(tensor_fn)
    total = 0
    pending = b""
    rows = _pinned.empty((num, _NV), np.float32)      # page-locked when a GPU is present: the consumer copies it to HBM
    meta = np.empty((num, 6), dtype=np.int64)
    c = 0
    bufs = []          # (bytes, meta rows) pieces of the batch being filled
    consumed = ctypes.c_int64(); nrows = ctypes.c_int64(); nbad = ctypes.c_int64()
    eof = False
    done = False
    try:
        while True:
            chunk = fo.read(1 << 24) if not eof else b""
            if not chunk:
                eof = True
                if pending and not pending.endswith(b"\n"):
                    pending += b"\n"
            data = pending + chunk if pending else chunk
            arr = np.frombuffer(data, dtype=np.uint8)          # addresses into the bytes object: no slice copies
            off = 0
            while off < len(data):
                _lib.check(lib.cv_parse_tensor_text(ctypes.c_void_p(arr.ctypes.data + off), len(data) - off, num - c,
                                                    rows[c:].ctypes.data_as(ctypes.c_void_p),
                                                    meta[c:].ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.byref(consumed), ctypes.byref(nrows), ctypes.byref(nbad)))
                if nbad.value:
                    print("UnpackATensorRecord Failure (%d malformed rows skipped)" % nbad.value, file=sys.stderr)
                if nrows.value:
                    bufs.append((arr[off:off + consumed.value], meta[c:c + nrows.value].copy()))
                c += nrows.value
                off += consumed.value
                if c == num:
                    total += c
                    if log:
                        print("Processed %d tensors" % total, file=sys.stderr)
                    yield 0, c, rows.reshape((num, 2 * param.flankingBaseNum + 1, 4, param.matrixNum)), _join_pos(bufs)
                    rows = _pinned.empty((num, _NV), np.float32)      # page-locked when a GPU is present: the consumer copies it to HBM
                    meta = np.empty((num, 6), dtype=np.int64)
                    c = 0
                    bufs = []
                elif consumed.value == 0:
                    break
            pending = data[off:]
            if eof:
                break
        done = True
    finally:
        _close_quietly_unless(done, proc, fo, tensor_fn)
    total += c
    if log:
        print("Processed %d tensors" % total, file=sys.stderr)
    yield 1, c, rows[:c].reshape((c, 2 * param.flankingBaseNum + 1, 4, param.matrixNum)), _join_pos(bufs)


def _join_pos(bufs):
    return PosBatch(pieces=bufs)


# ---- blosc container (python-blosc pack_array / unpack_array equivalents) ---------------
def blosc_decompress(chunk):
    lib = _lib.load()
    n = lib.cv_blosc_nbytes(chunk, len(chunk))
    if n < 0:
        raise _lib.CvError("blosc: truncated chunk")
    out = ctypes.create_string_buffer(max(int(n), 1))
    _lib.check(lib.cv_blosc_decompress(chunk, len(chunk), out, n))
    return out.raw[:n]


def blosc_compress(data, typesize):
    lib = _lib.load()
    out = ctypes.create_string_buffer(len(data) + 64)
    clen = ctypes.c_int64()
    _lib.check(lib.cv_blosc_compress_lz4(data, len(data), int(typesize), out, len(out), ctypes.byref(clen)))
    return out.raw[:clen.value]


def pack_array(arr):
    """blosc.pack_array: compress(pickle.dumps(array, HIGHEST_PROTOCOL), typesize=itemsize)"""
    return blosc_compress(pickle.dumps(arr, pickle.HIGHEST_PROTOCOL), arr.itemsize)


def unpack_array(chunk):
    """blosc.unpack_array; accepts blocks pickled by Python 2 (latin1 / bytes payloads)."""
    if isinstance(chunk, str):
        chunk = chunk.encode("latin1")
    raw = blosc_decompress(bytes(chunk))
    try:
        return pickle.loads(raw)
    except (UnicodeDecodeError, ValueError):
        return pickle.loads(raw, encoding="latin1")


def unpack_arrays(chunks):
    """blosc.unpack_array of several blocks: the chunks are decompressed concurrently by the native host threads
    (cv_blosc_decompress_many), then un-pickled"""
    lib = _lib.load()
    n = len(chunks)
    if n == 0:
        return []
    raw = [c.encode("latin1") if isinstance(c, str) else bytes(c) for c in chunks]
    sizes = [lib.cv_blosc_nbytes(r, len(r)) for r in raw]
    if min(sizes) < 0:
        raise _lib.CvError("blosc: truncated chunk")
    outs = [bytearray(max(int(sz), 1)) for sz in sizes]
    src = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(r), ctypes.c_void_p).value for r in raw])
    dst = (ctypes.c_void_p * n)(*[ctypes.addressof((ctypes.c_char * len(o)).from_buffer(o)) for o in outs])
    clen = (ctypes.c_int64 * n)(*[len(r) for r in raw])
    cap = (ctypes.c_int64 * n)(*[len(o) for o in outs])
    status = (ctypes.c_int32 * n)()
    _lib.check(lib.cv_blosc_decompress_many(src, clen, dst, cap, n, status))
    arrays = []
    for o, sz in zip(outs, sizes):
        view = memoryview(o)[:sz]
        try:
            arrays.append(pickle.loads(view))
        except (UnicodeDecodeError, ValueError):
            arrays.append(pickle.loads(view, encoding="latin1"))
    return arrays


class LazyBlocks(object):
    """The compressed-block list of a .bin file WITHOUT loading it: (offset, length) of every block inside the
    memory-mapped file.  Items are zero-copy memoryviews of the page cache, so the N ranks of a node share ONE copy
    of the data set in RAM (un-pickling gives every rank its own heap copy of all blocks) and a rank only ever
    touches the pages of the blocks its slices of the batches live in."""

    def __init__(self, mm, index):
        self._mm, self._index = mm, index
        self._view = memoryview(mm)

    def __len__(self):
        return len(self._index)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        off, n = self._index[i]
        return self._view[off:off + n]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def _scan_block_list(mm, pos):
    """Walks ONE pickle that holds a list of byte strings (what tensor2Bin.py:24-28 dumps, protocols 2..5) starting at
    byte `pos` of the mapped file, reading the opcode headers only -> ([(offset, length)], position after STOP),
    or None when the stream holds anything else (the caller then un-pickles)."""
    import struct
    n = len(mm)
    index = []

    def block(hdr, ln):
        # a block that would run past the end of the file (truncated / corrupt .bin): no index, the caller un-pickles
        # and reports the damage in pickle's own words
        if pos + hdr + ln > n:
            raise IndexError("block past the end of the file")
        index.append((pos + hdr, ln))
        return pos + hdr + ln
    try:
        while pos < n:
            op = mm[pos]
            if op == 0x80: pos += 2                                    # PROTO
            elif op == 0x95: pos += 9                                  # FRAME
            elif op in (0x5d, 0x28, 0x94, 0x65, 0x61): pos += 1        # EMPTY_LIST, MARK, MEMOIZE, APPENDS, APPEND
            elif op == 0x71: pos += 2                                  # BINPUT
            elif op == 0x72: pos += 5                                  # LONG_BINPUT
            elif op in (0x43, 0x55):                                   # SHORT_BINBYTES, SHORT_BINSTRING
                pos = block(2, mm[pos + 1])
            elif op in (0x42, 0x54):                                   # BINBYTES, BINSTRING
                pos = block(5, struct.unpack_from("<I", mm, pos + 1)[0])
            elif op == 0x8e:                                           # BINBYTES8
                pos = block(9, struct.unpack_from("<Q", mm, pos + 1)[0])
            elif op == 0x2e:                                           # STOP
                return index, pos + 1
            else:
                return None
    except (IndexError, struct.error):
        return None
    return None


def LoadBin(bin_fn, lazy=False):
    """The four back-to-back pickles of tensor2Bin.py:24-28 (also files written by Python 2) -> (total, XC, YC, posC).
    lazy: the three block lists as LazyBlocks over the memory-mapped file (the training loop under data parallelism:
    no rank holds a private copy of the data set); falls back to un-pickling when the file is not the plain
    list-of-byte-strings layout."""
    if lazy:
        import mmap
        fh = open(bin_fn, "rb")
        try:
            try:
                total = pickle.load(fh)
            except (UnicodeDecodeError, ValueError):
                fh.seek(0); total = pickle.load(fh, encoding="bytes")
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            pos = fh.tell()
            lists = []
            for _ in range(3):
                got = _scan_block_list(mm, pos)
                if got is None:
                    lists = None
                    break
                lists.append(LazyBlocks(mm, got[0])); pos = got[1]
            if lists is not None and all(len(l) == len(lists[0]) for l in lists):
                return total, lists[0], lists[1], lists[2]
        finally:
            fh.close()
    with open(bin_fn, "rb") as fh:
        try:
            objs = [pickle.load(fh) for _ in range(4)]
        except (UnicodeDecodeError, ValueError):
            fh.seek(0)
            objs = [pickle.load(fh, encoding="bytes") for _ in range(4)]
    return objs[0], objs[1], objs[2], objs[3]


def _label(row):
    """16-vector label of one truth row `ctg pos ref alt gt1 gt2` (utils_v2.py:90-119):
    base A,C,G,T | HET,HOM | REF,SNP,INS,DEL | length 0,1,2,3,4,>4"""
    ref, alt, g1, g2 = row[2], row[3], row[4], row[5]
    v = [0.0] * 16
    snp_like = len(ref) == 1 and len(alt) == 1
    if g1 == "0" and g2 == "1":
        if snp_like:
            v[base2num[ref[0]]] = 0.5
            v[base2num[alt[0]]] = 0.5
        else:
            v[base2num[ref[0]]] = 0.5
        v[4] = 1.0
    elif g1 == "1" and g2 == "1":
        if snp_like:
            v[base2num[alt[0]]] = 1
        v[5] = 1.0
    if len(ref) > 1 and len(alt) == 1:
        v[9] = 1.0
    elif len(alt) > 1 and len(ref) == 1:
        v[8] = 1.0
    else:
        v[7] = 1.0
    d = abs(len(ref) - len(alt))
    v[15 if d > 4 else 10 + d] = 1.0
    return v


class _Intervals(object):
    """Point-stabbing over the BED intervals of one contig (the reference uses
    intervaltree.IntervalTree.addi(begin, end) / search(pos), half-open [begin, end))."""

    def __init__(self):
        self.iv = []
        self._sorted = None

    def addi(self, b, e):
        self.iv.append((b, e))
        self._sorted = None

    def hit(self, p):
        if self._sorted is None:
            iv = sorted(self.iv)
            self._b = np.array([x[0] for x in iv], dtype=np.int64)
            # running maximum of the ends lets one bisect answer "any interval covers p"
            self._emax = np.maximum.accumulate(np.array([x[1] for x in iv], dtype=np.int64)) if iv else np.array([], dtype=np.int64)
            self._sorted = True
        k = int(np.searchsorted(self._b, p, side="right"))
        return k > 0 and self._emax[k - 1] > p


def _gz_lines(fn):
    f = subprocess.Popen(shlex.split("gzip -fdc %s" % (fn)), stdout=subprocess.PIPE, bufsize=8388608)
    for row in io.TextIOWrapper(f.stdout, encoding="ascii", errors="replace"):
        yield row
    f.stdout.close()
    f.wait()


def GetTrainingArray(tensor_fn, var_fn, bed_fn, shuffle=True):
    """utils_v2.py:62-186 -> (total, XArrayCompressed, YArrayCompressed, posArrayCompressed):
    blocks of param.bloscBlockSize items; X fp32 [k,33,4,4] (matrix-0-subtracted), Y float64
    [k,16], pos string array; a trailing (possibly empty) block is always appended."""
    tree = {}
    if bed_fn is not None:
        for row in _gz_lines(bed_fn):
            row = row.split()
            if not row:
                continue
            t = tree.setdefault(row[0], _Intervals())
            begin = int(row[1]); end = int(row[2]) - 1
            if end == begin:
                end += 1
            t.addi(begin, end)
    Y = {}
    if var_fn is not None:
        for row in _gz_lines(var_fn):
            row = row.split()
            if not row:
                continue
            ctg = row[0]; pos = int(row[1])
            if bed_fn is not None and not tree[ctg].hit(pos):
                continue
            Y[ctg + ":" + str(pos)] = _label(row)
    X = {}
    total = 0
    for end, c, xb, posb in GetTensor(tensor_fn, 4096, log=False):
        for j in range(c):
            chrom, coord, seq = posb[j].split(":")
            if bed_fn is not None:
                if chrom not in tree or not tree[chrom].hit(int(coord)):
                    continue
            key = chrom + ":" + coord
            X[key] = np.copy(xb[j])
            if key not in Y:
                v = [0.0] * 16
                v[5] = 1.0; v[6] = 1.0; v[10] = 1.0          # HOM, REF, length 0
                v[base2num[seq[param.flankingBaseNum]]] = 1.0
                Y[key] = v
            total += 1
            if total % 100000 == 0:
                print("Processed %d tensors" % total, file=sys.stderr)
    allPos = sorted(X.keys())
    if shuffle:
        random.shuffle(allPos)
    XC, YC, PC = [], [], []
    bs = param.bloscBlockSize
    for s in range(0, len(allPos), bs):
        keys = allPos[s:s + bs]
        if len(keys) < bs:
            break
        XC.append(pack_array(np.array([X[k] for k in keys])))
        YC.append(pack_array(np.array([Y[k] for k in keys])))
        PC.append(pack_array(np.array(keys)))
    keys = allPos[len(allPos) // bs * bs:]
    XC.append(pack_array(np.array([X[k] for k in keys])))
    YC.append(pack_array(np.array([Y[k] for k in keys])))
    PC.append(pack_array(np.array(keys)))
    return len(allPos), XC, YC, PC


_block_layout = {}          # id(block list) -> (dtype, item shape) learnt from its first block


class _PinnedPool(object):
    """Page-locked host buffers for the arrays DecompressArray hands out: the training loop copies every batch to the
    GPU right away, and a pinned source makes that copy a plain DMA.  A buffer goes back to the free list when the
    last numpy view of it has been garbage-collected (weakref.finalize on the root array), so handing arrays out
    is as safe as np.empty.  Without a GPU (or for very small / very large requests) np.empty is used."""
    MIN_BYTES, MAX_BYTES, KEEP = 1 << 18, 1 << 28, 6

    def __init__(self):
        import threading
        self.free = {}
        self.enabled = None
        self.lock = threading.RLock()     # (re-entrant: a finalizer may run inside _give when its allocation triggers a collection)
                                           # GetTensorFiles takes buffers from up to 8 reader threads; finalizers give them back

    def _give(self, cls, t):
        with self.lock:
            lst = self.free.setdefault(cls, [])
            if len(lst) < self.KEEP:
                lst.append(t)

    def empty(self, shape, dtype):
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if self.enabled is None:
            try:
                import torch
                self.enabled = bool(torch.cuda.is_available())
            except Exception:
                self.enabled = False
        if not self.enabled or nbytes < self.MIN_BYTES or nbytes > self.MAX_BYTES:
            return np.empty(shape, dtype=dtype)
        import torch
        import weakref
        cls = (nbytes + (1 << 20) - 1) >> 20 << 20
        with self.lock:
            lst = self.free.get(cls)
            t = lst.pop() if lst else None
        if t is None:
            t = torch.empty(cls, dtype=torch.uint8, pin_memory=True)
        root = t.numpy()
        weakref.finalize(root, self._give, cls, t)
        return root[:nbytes].view(dtype).reshape(shape)


_pinned = _PinnedPool()


def _unpack_into_one(blocks, key):
    """the blocks of one DecompressArray call decompressed straight into one array (no per-block un-pickling, no
    concatenation); None when the layout is not the plain one (the caller then takes the generic path)"""
    lib = _lib.load()
    lay = _block_layout.get(key) if key[2] is not None else None
    if lay is None:
        if len(_block_layout) > 64:
            _block_layout.clear()
        a0 = unpack_array(blocks[0]) if len(blocks) else None
        # numeric arrays only (X fp32, Y f8): their item size is the same in every block, whereas numpy sizes a
        # string array ('<U8' vs '<U10' position keys) per block, so a narrower last block could pass the length check
        # and be reinterpreted with the first block's item size
        if not isinstance(a0, np.ndarray) or a0.ndim < 1 or not a0.flags.c_contiguous or a0.dtype.kind not in "fiub" \
                or len(a0) != param.bloscBlockSize:
            return None
        lay = (a0.dtype, a0.shape[1:])
        if key[2] is not None:
            _block_layout[key] = lay
    dtype, ishape = lay
    bs = param.bloscBlockSize
    item = int(np.prod(ishape, dtype=np.int64)) * dtype.itemsize
    n = len(blocks)
    # addresses of the compressed blocks where they lie (bytes objects, or views of the mapped file: no copies)
    raw = [c.encode("latin1") if isinstance(c, str) else c for c in blocks]
    hold = [r if isinstance(r, bytes) else np.frombuffer(r, dtype=np.uint8) for r in raw]
    out = _pinned.empty((n * bs,) + tuple(ishape), dtype)
    src = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(h), ctypes.c_void_p).value if isinstance(h, bytes)
                                  else h.ctypes.data for h in hold])
    clen = (ctypes.c_int64 * n)(*[len(r) for r in raw])
    lens = (ctypes.c_int64 * n)()
    status = (ctypes.c_int32 * n)()
    if lib.cv_blosc_unpack_blocks(src, clen, n, out.ctypes.data_as(ctypes.c_void_p), bs * item, lens, status) != 0:
        return None
    if lens[n - 1] % item:
        return None
    return out[:(n - 1) * bs + lens[n - 1] // item]


def DecompressArray(array, start, num, maximum):
    """utils_v2.py:189-207 -> (items [start, start+num) clipped at maximum, count, endFlag)."""
    endFlag = 0
    if start + num >= maximum:
        num = maximum - start
        endFlag = 1
    bs = param.bloscBlockSize
    leftEnd = start % bs
    first = int(start / bs)
    last = int((start + num - 1) / bs)
    # layout cache key: the list object plus a fingerprint of its first block (ids are reused after a list dies)
    key = (id(array), len(array), bytes(array[0][:48]) if len(array) and not isinstance(array[0], str) else None)
    out = _unpack_into_one(array[first:last + 1], key)
    if out is None:
        parts = unpack_arrays(array[first:last + 1])
        out = np.concatenate(parts[:]) if len(parts) > 1 else parts[0]
    if leftEnd != 0 or num % bs != 0:
        out = out[leftEnd:(leftEnd + num)]
    return out, num, endFlag
