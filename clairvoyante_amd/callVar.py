"""Variant-calling driver: same command line, VCF output and function surface as
/root/reference/clairvoyante/callVar.py (Run :21-47, Output :50-153, PrintVCFHeader
:156-178, Test :180-216, main :219-262).

    python -m clairvoyante_amd.callVar --chkpnt_fn MODEL --tensor_fn TENSORS.gz --call_fn OUT.vcf

Pipeline (one process per GPU): a reader thread parses text tensors into pinned batches
(native parser), the GPU runs the network and the per-candidate arg-max / quality / depth
reductions (cv_call_postproc), and the host formats VCF lines only for the candidates
that produce one.  `Output()` keeps the reference's signature (host arrays in, text out)
and is the formatter both paths share.
"""
import argparse
import ctypes
import logging
import os
import sys
import time
from math import log
from queue import Queue
from threading import Thread

import numpy as np

if __package__ in (None, ""):      # run as `python <dir>/callVar.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param

logging.basicConfig(format='%(message)s', level=logging.INFO)
num2base = dict(zip((0, 1, 2, 3), "ACGT"))
base2num = dict(zip("ACGT", (0, 1, 2, 3)))
v2Zygosity2Name = dict(zip((0, 1), ('HET', 'HOM')))
v2Type2Name = dict(zip((0, 1, 2, 3), ('REF', 'SNP', 'INS', 'DEL')))
v2Length2Name = dict(zip((0, 1, 2, 3, 4, 5), ('0', '1', '2', '3', '4', '4+')))
maxVarLength = 5
inferIndelLengthMinimumAF = 0.125
_F = param.flankingBaseNum


def _top2_products(t, z, l):
    """fp32 products of the best and second-best probabilities of the three softmax heads
    (np.sort(x)[::-1][0/1], callVar.py:69-72); evaluated left to right in fp32 like NumPy."""
    st = -np.sort(-t, axis=1); sz = -np.sort(-z, axis=1); sl = -np.sort(-l, axis=1)
    p1 = (st[:, 0] * sz[:, 0]) * sl[:, 0]
    p2 = (st[:, 1] * sz[:, 1]) * sl[:, 1]
    return p1.astype(np.float32), p2.astype(np.float32)


def _qual(p1, p2):
    """callVar.py:72 with reference-era promotion: fp32 products, then float64."""
    return int(-4.343 * log((float(p2) + 1e-300) / (float(p1) + 1e-300)))


def _depth(X):
    """dp of callVar.py:86-87 for every candidate (values are integers: exact in fp32)"""
    return ((X[:, _F, :, 0].sum(1, dtype=np.float32) + X[:, _F + 1, :, 1].sum(1, dtype=np.float32))
            + X[:, _F + 1, :, 2].sum(1, dtype=np.float32)) + X[:, _F, :, 3].sum(1, dtype=np.float32)


def _format_record(args, x, pos, varType, varZygosity, varLength, base1, base2, qual, dp):
    """One VCF line (or None when dp == 0) from per-candidate decisions; x is the candidate's
    [33,4,4] tensor.  Allele / length inference of callVar.py:88-153."""
    if dp == 0:
        return None
    chromosome, coordination, refSeq = pos.split(":")
    coordination = int(coordination)
    info = []
    inferred = 0
    refBase = refSeq[_F]
    if varType == 1 or varType == 0:
        b1 = num2base[base1]; b2 = num2base[base2]
        altBase = refBase if varType == 0 else (b1 if b1 != refBase else b2)
        af = x[_F, base2num[altBase], 3] / dp
    elif varType == 2:
        if varLength == 0:
            varLength = 1
        af = x[_F + 1, :, 1].sum(dtype=np.float32) / dp
        ins = ""
        if varLength != maxVarLength:
            for k in range(_F + 1, _F + varLength + 1):
                ins += num2base[int(np.argmax(x[k, :, 1]))]
        else:
            for k in range(_F + 1, 2 * _F + 1):
                if k < (_F + maxVarLength) or x[k, :, 1].sum(dtype=np.float32) >= inferIndelLengthMinimumAF * x[k, :, 0].sum(dtype=np.float32):
                    inferred += 1
                    ins += num2base[int(np.argmax(x[k, :, 1]))]
                else:
                    break
        if inferred >= _F:
            altBase = "<INS>"
            info.append("SVTYPE=INS")
        else:
            altBase = refBase + ins
    else:
        if varLength == 0:
            varLength = 1
        af = x[_F + 1, :, 2].sum(dtype=np.float32) / dp
        if varLength == maxVarLength:
            for k in range(_F + 1, 2 * _F + 1):
                if k < (_F + maxVarLength) or x[k, :, 2].sum(dtype=np.float32) >= inferIndelLengthMinimumAF * x[k, :, 0].sum(dtype=np.float32):
                    inferred += 1
                else:
                    break
        if inferred >= _F:
            altBase = "<DEL>"
            info.append("SVTYPE=DEL")
        elif varLength != maxVarLength:
            refBase = refSeq[_F:_F + varLength + 1]
            altBase = refSeq[_F]
        else:
            refBase = refSeq[_F:_F + inferred + 1]
            altBase = refSeq[_F]
    if 0 < inferred < _F:
        info.append("LENGUESS=%d" % inferred)
    infoStr = ";".join(info) if info else "."
    if varType == 0:
        gt = "0/0"
    else:
        gt = "0/1" if varZygosity == 0 else "1/1"
    filt = "."
    if args.qual is not None:
        filt = "PASS" if qual >= args.qual else "LowQual"
    return "%s\t%d\t.\t%s\t%s\t%d\t%s\t%s\tGT:GQ:DP:AF\t%s:%d:%d:%.4f" % (
        chromosome, coordination, refBase, altBase, qual, filt, infoStr, gt, qual, dp, af)


def Output(args, call_fh, num, XBatch, posBatch, base, z, t, l):
    """callVar.py:50-153: writes the VCF records of one batch given host arrays."""
    if num != len(base):
        sys.exit("Inconsistent shape between input tensor and output predictions %d/%d" % (num, len(base)))
    if num == 0:
        return
    X = np.asarray(XBatch, dtype=np.float32)
    base = np.asarray(base); z = np.asarray(z); t = np.asarray(t); l = np.asarray(l)
    varType = np.argmax(t, axis=1)
    keep = np.arange(num) if args.showRef else np.nonzero(varType != 0)[0]
    if keep.size == 0:
        return
    varZyg = np.argmax(z, axis=1)
    varLen = np.argmax(l, axis=1)
    p1, p2 = _top2_products(t[keep], z[keep], l[keep])
    # argsort()[::-1]: descending, the higher index first among equal values (callVar.py:81)
    order = np.argsort(base[keep], axis=1, kind="stable")[:, ::-1]
    dp = _depth(X[keep])
    lines = []
    for i, j in enumerate(keep):
        rec = _format_record(args, X[j], posBatch[j], int(varType[j]), int(varZyg[j]), int(varLen[j]),
                             int(order[i, 0]), int(order[i, 1]), _qual(p1[i], p2[i]), dp[i])
        if rec is not None:
            lines.append(rec)
    if lines:
        call_fh.write("\n".join(lines) + "\n")


def PrintVCFHeader(args, call_fh):
    """callVar.py:156-178"""
    hdr = ['##fileformat=VCFv4.1',
           '##FILTER=<ID=PASS,Description="All filters passed">',
           '##FILTER=<ID=LowQual,Description="Confidence in this variant being real is below calling threshold.">',
           '##ALT=<ID=DEL,Description="Deletion">',
           '##ALT=<ID=INS,Description="Insertion of novel sequence">',
           '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">',
           '##INFO=<ID=LENGUESS,Number=.,Type=Integer,Description="Best guess of the indel length">',
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
           '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">',
           '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">',
           '##FORMAT=<ID=AF,Number=1,Type=Float,Description="Estimated allele frequency in the range (0,1)">']
    if args.ref_fn is not None:
        with open(args.ref_fn + ".fai") as fai_fp:
            for line in fai_fp:
                fields = line.strip().split("\t")
                hdr.append("##contig=<ID=%s,length=%d>" % (fields[0], int(fields[1])))
    hdr.append('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s' % (args.sampleName))
    call_fh.write("\n".join(hdr) + "\n")


_tls = __import__("threading").local()


def _out_buffer(cap):
    """a grow-only output buffer per calling thread (a fresh ctypes buffer is zero-filled on every call)"""
    buf = getattr(_tls, "buf", None)
    if buf is None or len(buf) < cap:
        buf = _tls.buf = (ctypes.c_char * max(cap, 1 << 20))()
    return buf


def format_records(args, num, X, pos, call, qual, xrow=None):
    """The VCF text (bytes) of one batch through the native formatter (cv_format_vcf, csrc/cv_hostio.cpp):
    `call` [num,8] int32 / `qual` [num,4] fp32 host arrays from cv_call_postproc, X the tensors ([rows,33,4,4]
    fp32, C-contiguous; candidate i = row xrow[i], or row i), pos a utils_v2.PosBatch or a sequence of
    "chrom:coord:seq" strings.  Same records as `_format_record` line for line (tests/test_host_golden.py)."""
    from . import _lib
    from .utils_v2 import PosBatch
    if num == 0:
        return b""
    lib = _lib.load()
    if not isinstance(pos, PosBatch):
        pos = PosBatch.from_strings([pos[i] for i in range(num)])
    call = np.ascontiguousarray(call, dtype=np.int32); qual = np.ascontiguousarray(qual, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    if xrow is not None:
        xrow = np.ascontiguousarray(xrow, dtype=np.int64)
    out = []
    nlen = ctypes.c_int64(); nrec = ctypes.c_int64()
    row_bytes = X.strides[0] if X.ndim > 1 else 0
    for start, rows, buf, meta in pos.pieces():
        if rows == 0:
            continue
        stop = min(start + rows, num)
        if stop <= start:
            break
        n = stop - start
        kept = n if args.showRef else int(np.count_nonzero(call[start:stop, 0]))
        if kept == 0:
            continue
        meta = np.ascontiguousarray(meta, dtype=np.int64)
        cap = kept * (int(meta[:n, 1].max()) + 160)
        if isinstance(buf, np.ndarray):              # a view of the memory-mapped input: its address, no copy
            hold = buf if buf.flags.c_contiguous else np.ascontiguousarray(buf)
            cbuf = ctypes.c_void_p(hold.ctypes.data)
        else:
            cbuf = buf if isinstance(buf, bytes) else bytes(buf)
        xptr = X.ctypes.data + (0 if xrow is not None else start * row_bytes)
        xr = ctypes.c_void_p(xrow.ctypes.data + start * 8) if xrow is not None else None
        for _try in range(2):
            dst = _out_buffer(cap)
            rc = lib.cv_format_vcf(ctypes.c_void_p(call.ctypes.data + start * 32), ctypes.c_void_p(qual.ctypes.data + start * 16),
                                   n, ctypes.c_void_p(xptr), xr, cbuf, ctypes.c_void_p(meta.ctypes.data), None,
                                   1 if args.showRef else 0, 0 if args.qual is None else 1,
                                   0 if args.qual is None else int(args.qual), dst, cap, ctypes.byref(nlen), ctypes.byref(nrec))
            if rc != 2:
                break
            cap = nlen.value
        _lib.check(rc)
        out.append(ctypes.string_at(dst, nlen.value))
    return b"".join(out)


def OutputFromDevice(args, call_fh, num, XBatch, posBatch, call, qual, xrow=None):
    """Formatter of the GPU path: `call` [n,8] int32 and `qual` [n,4] fp32 come from cv_call_postproc (arg-maxes,
    two best bases, the fp32 top-2 products, dp); the records are written by the native formatter."""
    text = format_records(args, num, XBatch, posBatch, call, qual, xrow)
    if text:
        call_fh.write(text.decode("ascii"))


def predict_and_reduce(m, xd):
    """network + per-candidate reductions on the device: x [n,33,4,4] -> (call [n,8] int32, qual [n,4] fp32)"""
    import torch
    from . import _lib
    num = xd.shape[0]
    out = m.predict_device(xd)
    call = torch.empty((num, 8), dtype=torch.int32, device=m.device)
    qual = torch.empty((num, 4), dtype=torch.float32, device=m.device)
    _lib.check(m._lib.cv_call_postproc(m._h, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(out.data_ptr()), num,
                                       ctypes.c_void_p(call.data_ptr()), ctypes.c_void_p(qual.data_ptr()), m._stream()))
    return call, qual


class ResultFetcher(object):
    """Brings the per-candidate decisions of a batch to the host WITHOUT queueing behind the next batch's kernels:
    the copies run on a side stream that waits for the event recorded after the batch's own kernels, into page-locked
    buffers, so the host formats batch k while the GPU computes batch k+1 (callVar.py:197-204 overlaps Output(k) with
    predict(k+1) the same way)."""

    def __init__(self, m):
        import torch
        self.m = m
        self.side = torch.cuda.Stream(device=m.device)

    def mark(self):
        """call right after enqueueing a batch's kernels on the current stream"""
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.m.device))
        return ev

    def fetch(self, ev, call_d, qual_d, x_dev=None, show_ref=False):
        """-> (call, qual) host arrays; with x_dev also (rows [k,33,4,4] of the candidates that give a record, xrow [n]
        = row of candidate i in `rows`) -- only those rows cross PCIe"""
        import torch
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            call_h = torch.empty(call_d.shape, dtype=call_d.dtype, pin_memory=True)
            qual_h = torch.empty(qual_d.shape, dtype=qual_d.dtype, pin_memory=True)
            call_h.copy_(call_d, non_blocking=True); qual_h.copy_(qual_d, non_blocking=True)
            self.side.synchronize()
            call = call_h.numpy(); qual = qual_h.numpy()
            if x_dev is None:
                return call, qual
            n = call.shape[0]
            keep = np.arange(n) if show_ref else np.flatnonzero(call[:, 0] != 0)
            xrow = np.zeros(n, dtype=np.int64)
            xrow[keep] = np.arange(len(keep))
            if len(keep) == 0:
                return call, qual, np.zeros((0, 33, 4, 4), np.float32), xrow
            idx = torch.from_numpy(keep).to(self.m.device, non_blocking=True)
            rows_h = torch.empty((len(keep),) + tuple(x_dev.shape[1:]), dtype=torch.float32, pin_memory=True)
            rows_h.copy_(x_dev.index_select(0, idx), non_blocking=True)
            self.side.synchronize()
            return call, qual, rows_h.numpy(), xrow


def CallFromDevice(args, m, call_fh, X_dev, pos, batch=65536):
    """VCF records for tensors that are already in HBM (callVarBam's fused path): X_dev [n,33,4,4] with matrices 1..3
    minus matrix 0; pos: utils_v2.PosBatch (or a function i -> "chrom:coord:seq") for the n rows.  Batch k's decisions
    and the rows that produce a record (non-REF calls, or all with --showRef) come to the host on a side stream and
    are formatted by the host threads while the GPU runs batch k+1."""
    import torch
    from .utils_v2 import PosBatch
    n = X_dev.shape[0]
    if callable(pos):
        pos = PosBatch.from_strings([pos(i) for i in range(n)])
    pieces = pos.pieces()
    if len(pieces) != 1:
        pos = PosBatch.from_strings(list(pos))
        pieces = pos.pieces()
    _start, _rows, buf, meta = pieces[0]
    with torch.cuda.device(m.device):
        fetcher = ResultFetcher(m)
        pending = None

        def finish(p):
            s, xd, call_d, qual_d, ev = p
            call, qual, rows, xrow = fetcher.fetch(ev, call_d, qual_d, xd, bool(args.showRef))
            OutputFromDevice(args, call_fh, call.shape[0], rows, PosBatch(buf, meta[s:s + call.shape[0]]), call, qual, xrow)

        for s in range(0, n, batch):
            xd = X_dev[s:s + batch].contiguous()
            call_d, qual_d = predict_and_reduce(m, xd)
            nxt = (s, xd, call_d, qual_d, fetcher.mark())
            if pending is not None:
                finish(pending)
            pending = nxt
        if pending is not None:
            finish(pending)


def Run(args):
    """callVar.py:21-47"""
    logging.info("Loading model ...")
    from . import utils_v2 as utils
    utils.SetupEnv()
    if args.v2:
        sys.exit("Clairvoyante v2 topologies are not part of this build (v3 / v3 slim only)")
    if args.slim:
        from . import clairvoyante_v3_slim as cv
    else:
        from . import clairvoyante_v3 as cv
    if args.threads is None:
        if args.tensor_fn == "PIPE":
            param.NUM_THREADS = 4
    else:
        param.NUM_THREADS = args.threads
    from . import parallel
    rank, ws, _local = parallel.init_from_env(os.environ.get("CV_DIST_BACKEND"))   # torchrun: one process per GPU
    m = cv.Clairvoyante()
    m.init()
    m.restoreParameters(os.path.abspath(args.chkpnt_fn))
    if ws > 1:
        TestSharded(args, m, utils, rank, ws)
    else:
        Test(args, m, utils)


def tensor_files(tensor_fn):
    """--tensor_fn as a list: "a.gz,b.gz,c.gz" names one tensor file per chunk of the genome, called in list order (one
    process) or file k by rank k % N (torchrun).  A name that exists as given is ONE file even if it holds a comma; PIPE
    stays PIPE.  (A list whose every item holds a comma of its own cannot be expressed: rename the files.)"""
    if tensor_fn == "PIPE" or "," not in tensor_fn or os.path.exists(tensor_fn):
        return [tensor_fn]
    return [f for f in tensor_fn.split(",") if f]


# input lines per block of the multi-rank split (block k -> rank k % N); CV_SHARD_BLOCK_LINES overrides (tests)
SHARD_BLOCK_LINES = int(os.environ.get("CV_SHARD_BLOCK_LINES", "16384"))


def merge_fragments(call_fn, ws, out_fh):
    """Rank 0: append the per-rank record fragments to the VCF in input order.  Rank r wrote CALL.rank<r> (its
    records, block after block) and CALL.rank<r>.idx (one line per block it owns: "<block> <bytes>"); block k
    belongs to rank k % ws, so walking k = 0, 1, 2, ... and taking the next <bytes> of that rank's fragment
    reproduces the single-process file -- the reference's own multi-process recipe is per-chunk VCFs joined by
    `vcfcat` (README.md:184-202)."""
    frags = [open("%s.rank%d" % (call_fn, r), "rb") for r in range(ws)]
    idx = []
    for r in range(ws):
        with open("%s.rank%d.idx" % (call_fn, r)) as f:
            idx.append([tuple(int(v) for v in line.split()) for line in f if line.strip()])
    nblocks = sum(len(i) for i in idx)
    for k in range(nblocks):
        r = k % ws
        if k // ws >= len(idx[r]) or idx[r][k // ws][0] != k:
            raise RuntimeError("fragment index of rank %d does not hold block %d" % (r, k))
        nbytes = idx[r][k // ws][1]
        if nbytes:
            data = frags[r].read(nbytes)
            if len(data) != nbytes:
                raise RuntimeError("fragment of rank %d is short at block %d" % (r, k))
            out_fh.write(data.decode("ascii"))
    for f in frags:
        f.close()
    for r in range(ws):
        os.remove("%s.rank%d" % (call_fn, r)); os.remove("%s.rank%d.idx" % (call_fn, r))


def TestSharded(args, m, utils, rank, ws):
    """Test() under torchrun (BASELINE configs[2]: candidates sharded over the GPUs of a node): candidates are
    independent (v3.py:54-138 has no cross-candidate state), so the ranks split the INPUT LINES block-cyclically,
    every rank calls its blocks with its own replica of the weights -- no collective on the data path -- and
    rank 0 joins the per-rank record fragments in input order: the VCF is byte-identical to the single-process
    one."""
    import torch
    import torch.distributed as dist
    # the fragments are files next to --call_fn that rank 0 reads back: one node, or a file system all ranks share
    if int(os.environ.get("LOCAL_WORLD_SIZE", str(ws))) != ws and rank == 0:
        logging.warning("callVar under %d ranks on several nodes: --call_fn must lie on a file system every rank shares" % ws)
    logging.info("Calling variants (rank %d of %d) ..." % (rank, ws))
    predictStart = time.time()
    frag_fn = "%s.rank%d" % (args.call_fn, rank)
    frag = open(frag_fn, "wb")
    index = []
    fetcher = ResultFetcher(m)
    q_in = Queue(maxsize=4)

    # "--tensor_fn a.gz,b.gz,...": one file per chunk, file k -> rank k % ws (each rank inflates only its own files);
    # a single file: its LINES are split block-cyclically (every rank inflates the whole stream -- the job is then
    # capped by one core's gzip rate whatever the number of GPUs)
    files = tensor_files(args.tensor_fn)
    if len(files) == 1 and ws > 1 and rank == 0 and utils.is_compressed(files[0]):
        logging.warning("callVar: ONE compressed tensor file under %d ranks -- every rank inflates the whole stream to find its "
                        "line blocks, so the job runs at one core's inflate rate (~0.35 M rows/s) whatever the number of GPUs. "
                        "Give --tensor_fn a comma-separated list (one file per chunk of the genome; file k goes to rank k mod N) "
                        "or uncompressed text to scale." % ws)

    def reader():
        try:
            if len(files) > 1:       # this rank's files, compressed ones several at a time, batches as they complete
                for item in utils.GetTensorFiles(files, max(param.predictBatchSize, 16384), rank, ws, ordered=False):
                    q_in.put(item)
            else:                    # one batch per owned block of lines
                for block, num, X, pos in utils.GetTensorBlocks(files[0], SHARD_BLOCK_LINES, rank, ws):
                    q_in.put((block, num, X, pos))
                    q_in.put((block, None, None, None))
        except BaseException as e:
            q_in.put(e)
        q_in.put(None)

    # the fragment holds this rank's blocks / files in ascending order (merge_fragments walks k = 0, 1, 2, ...): text that
    # is ready before its turn waits in memory (_OrderedWriter); one index entry per block that was seen
    seen = set()

    def write(k, text):
        frag.write(text)
        if index and index[-1][0] == k:
            index[-1] = (k, index[-1][1] + len(text))
        else:
            index.append((k, len(text)))

    out = _OrderedWriter(write, first=rank, step=ws)

    rt = Thread(target=reader, daemon=True)
    rt.start()
    pending = None
    failure = None
    try:
        with torch.cuda.device(m.device):
            while True:
                item = q_in.get()
                if isinstance(item, BaseException):
                    raise item
                nxt = None
                if item is not None and item[1] is not None and item[1] > 0:
                    block, num, X, pos = item
                    xd = torch.from_numpy(X).to(m.device, non_blocking=True)
                    call, qual = predict_and_reduce(m, xd)
                    nxt = (block, num, X, pos, call, qual, fetcher.mark())
                if pending is not None:
                    pblock, pnum, pX, ppos, pcall, pqual, pev = pending
                    hcall, hqual = fetcher.fetch(pev, pcall, pqual)
                    out.add(pblock, format_records(args, pnum, pX, ppos, hcall, hqual))
                pending = nxt
                if item is not None and item[1] is None:      # block / file complete (its last batch was formatted above)
                    seen.add(item[0])
                    out.end(item[0])
                if item is None:
                    break
        # every block this rank owns gets an index entry, also one without records (merge_fragments counts them)
        have = dict(index)
        index = [(k, have.get(k, 0)) for k in sorted(seen)]
        frag.close()
        with open(frag_fn + ".idx", "w") as f:
            f.write("".join("%d %d\n" % e for e in index))
    except BaseException as e:          # the other ranks must not wait at the barrier for a rank that died
        failure = e
    # every rank learns whether all fragments are complete (MAX of a flag) before anyone merges or leaves
    flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32,
                        device=m.device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) != 0:
        for fn in (frag_fn, frag_fn + ".idx"):
            try:
                os.remove(fn)
            except OSError:
                pass
        if failure is not None:
            raise failure
        sys.exit("callVar: another rank failed; no VCF written")
    if rank == 0:
        try:
            with open(args.call_fn, "w") as call_fh:
                PrintVCFHeader(args, call_fh)
                merge_fragments(args.call_fn, ws, call_fh)
        finally:                                    # also when the merge fails: no fragment is left behind
            for r in range(ws):
                for fn in ("%s.rank%d" % (args.call_fn, r), "%s.rank%d.idx" % (args.call_fn, r)):
                    try:
                        os.remove(fn)
                    except OSError:
                        pass
        logging.info("Total time elapsed: %.2f s" % (time.time() - predictStart))
    dist.barrier()


class _OrderedWriter(object):
    """VCF text of several input files, produced in any order, written in LIST order: the text of the file whose turn
    it is goes straight to the output, that of files further down the list waits in memory (records only -- a few per
    cent of the input rows) until the files in front of them are complete."""

    def __init__(self, write, first=0, step=1):
        self.write, self.next, self.step = write, first, step
        self.held, self.done = {}, set()

    def add(self, k, text):
        if not text:
            return
        if k == self.next:
            self.write(k, text)
        else:
            self.held.setdefault(k, []).append(text)

    def end(self, k):
        self.done.add(k)
        while self.next in self.done:
            self.done.discard(self.next)
            self.next += self.step
            for text in self.held.pop(self.next, []):
                self.write(self.next, text)


def Test(args, m, utils):
    """callVar.py:180-216 re-cut for the GPU: reader thread(s) (inflate, parse) || GPU (predict + per-candidate
    reductions) || writer (format); the records of a file stay in the order of its rows, files in list order, so the VCF
    is the reference's record for record."""
    import torch
    from . import _lib
    call_fh = open(args.call_fn, "w")
    PrintVCFHeader(args, call_fh)
    logging.info("Calling variants ...")
    predictStart = time.time()
    files = tensor_files(args.tensor_fn)
    # plain text is parsed where the page cache holds it at millions of rows/s: batches of 65 536 (the pass size of
    # the network); a compressed stream arrives at one core's inflate rate: smaller batches keep the stages overlapped
    mapped = utils._map_plain_text(files[0]) is not None if hasattr(utils, "_map_plain_text") else False
    batch = getattr(args, "batch_size", None) or (65536 if mapped else max(param.predictBatchSize, 16384))
    q_in = Queue(maxsize=2 if mapped else 4)

    def reader():
        try:
            if len(files) > 1:                   # a list of files: compressed ones are inflated several at a time and their
                for item in utils.GetTensorFiles(files, batch, 0, 1, ordered=False):     # batches taken as they complete
                    q_in.put(item)
            else:
                for _end, c, X, pos in utils.GetTensor(files[0], batch):
                    q_in.put((0, c, X, pos))
                q_in.put((0, None, None, None))
        except BaseException as e:   # surfaced on the consumer side (the reference loses it)
            q_in.put(e)
        q_in.put(None)

    rt = Thread(target=reader, daemon=True)
    rt.start()
    out = _OrderedWriter(lambda _k, text: call_fh.write(text.decode("ascii")))
    pending = None                   # (file, num, X, pos, call_dev, qual_dev, event)
    with torch.cuda.device(m.device):
        fetcher = ResultFetcher(m)
        while True:
            item = q_in.get()
            if isinstance(item, BaseException):
                raise item
            nxt = None
            if item is not None and item[1] is not None and item[1] > 0:
                k, num, X, pos = item
                xd = torch.from_numpy(X).to(m.device, non_blocking=True)
                call, qual = predict_and_reduce(m, xd)
                nxt = (k, num, X, pos, call, qual, fetcher.mark())
            if pending is not None:      # format batch j while the GPU works on batch j+1
                pk, pnum, pX, ppos, pcall, pqual, pev = pending
                hcall, hqual = fetcher.fetch(pev, pcall, pqual)
                out.add(pk, format_records(args, pnum, pX, ppos, hcall, hqual))
            pending = nxt
            if item is not None and item[1] is None:      # end of file k (its last batch was formatted just above)
                out.end(item[0])
            if item is None:
                break
    call_fh.close()
    logging.info("Total time elapsed: %.2f s" % (time.time() - predictStart))


_CLI = (   # flag, type, default, help  -- the reference's options and defaults (callVar.py:219-254)
    ("--tensor_fn", str, "PIPE", "Tensor input, use PIPE for standard input"),
    ("--chkpnt_fn", str, None, "Input a checkpoint for testing or continue training"),
    ("--call_fn", str, None, "Output variant predictions"),
    ("--qual", int, None, "If set, variant with equal or higher quality will be marked PASS, or LowQual otherwise, optional"),
    ("--sampleName", str, "SAMPLE", "Define the sample name to be shown in the VCF file"),
    ("--ref_fn", str, None, "Reference fasta file input, optional, print contig tags in the VCF header if set"),
    ("--threads", int, None, "Number of threads, optional"),
)
_SWITCHES = (("--showRef", False, "Show reference calls, optional"), ("--v3", True, "Use Clairvoyante version 3"),
             ("--v2", False, "Use Clairvoyante version 2"),
             ("--slim", False, "Train using the slim version of Clairvoyante, optional"))


def main():
    parser = argparse.ArgumentParser(
        description="Call variants using a trained Clairvoyante model and tensors of candididate variants")
    for flag, typ, default, text in _CLI:
        parser.add_argument(flag, type=typ, default=default, help=text)
    for flag, default, text in _SWITCHES:
        parser.add_argument(flag, type=param.str2bool, nargs='?', const=True, default=default, help=text)
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
