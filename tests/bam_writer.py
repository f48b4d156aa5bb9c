"""Test infrastructure: write a BAM file (+ .bai) from SAM text, per the SAM/BAM specification (SAMv1 sections 4.1
BGZF, 4.2 BAM records, 5.2 BAI with the standard binning scheme and 16 kb linear index), so that the native reader
(csrc/cv_bam.cpp) can be checked against the same alignments it would get as text from `samtools view`.
No htslib / samtools exists in the build image: reader and writer are two independent restatements of the format."""
import re
import struct
import zlib

_OPS = "MIDNSHP=X"
_NT = "=ACMGRSVTWYHKDBN"
_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
            + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def encode_record(fields, tid_of):
    qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = fields[:11]
    flag, pos, mapq, pnext, tlen = int(flag), int(pos) - 1, int(mapq), int(pnext) - 1, int(tlen)
    tid = tid_of.get(rname, -1)
    ntid = tid if rnext == "=" else tid_of.get(rnext, -1)
    ops = [(int(n), _OPS.index(o)) for n, o in _CIG.findall(cigar)] if cigar != "*" else []
    span = sum(n for n, o in ops if o in (0, 2, 3, 7, 8)) or 1
    l_seq = 0 if seq == "*" else len(seq)
    name = qname.encode() + b"\0"
    aux = b""
    inline = ops
    if len(ops) > 65535:                 # SAMv1 4.2.2: placeholder <l_seq>S<span>N inline, the real CIGAR in CG:B,I
        inline = [(l_seq, 4), (span, 3)]
        aux = b"NMi" + struct.pack("<i", 7) + b"XZZhello\0" + b"CGBI" + struct.pack("<i", len(ops)) + \
            b"".join(struct.pack("<I", (n << 4) | o) for n, o in ops) + b"XBBs" + struct.pack("<ihh", 2, -1, 5)
    out = struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, reg2bin(pos, pos + span), len(inline), flag, l_seq, ntid, pnext, tlen)
    out += name + b"".join(struct.pack("<I", (n << 4) | o) for n, o in inline)
    sq = bytearray((l_seq + 1) // 2)
    for i in range(l_seq):
        c = seq[i].upper()
        v = _NT.index(c) if c in _NT else 15
        sq[i >> 1] |= v << (4 if i % 2 == 0 else 0)
    out += bytes(sq)
    out += (b"\xff" * l_seq) if qual == "*" or len(qual) != l_seq else bytes(ord(c) - 33 for c in qual)
    out += aux
    return struct.pack("<i", len(out)) + out, tid, pos, pos + span


def write_bam(path, sam_lines, refs, block_payload=60000, index=True, header_text=None):
    """sam_lines: records (header lines ignored) sorted by (contig order, POS); refs: [(name, length)]."""
    tid_of = {n: i for i, (n, _l) in enumerate(refs)}
    text = (header_text if header_text is not None else
            "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)).encode()
    stream = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for n, l in refs:
        nb = n.encode() + b"\0"
        stream += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    recs = []          # (offset in the inflated stream, length, tid, beg, end)
    for line in sam_lines:
        if not line or line.startswith("@"):
            continue
        blob, tid, beg, end = encode_record(line.rstrip("\n").split("\t"), tid_of)
        recs.append((len(stream), len(blob), tid, beg, end))
        stream += blob
    # BGZF blocks of `block_payload` inflated bytes (records may straddle blocks, as htslib writes them)
    blocks = []        # (file offset, inflated offset, inflated length)
    out = bytearray()
    for off in range(0, len(stream), block_payload):
        chunk = bytes(stream[off:off + block_payload])
        blocks.append((len(out), off, len(chunk)))
        out += _bgzf_block(chunk)
    out += _EOF
    with open(path, "wb") as fh:
        fh.write(out)
    if not index:
        return

    def voff(pos):
        k = min(pos // block_payload, len(blocks) - 1)
        if pos == len(stream) and pos % block_payload == 0 and pos > 0:
            return (len(out) - len(_EOF)) << 16
        return (blocks[k][0] << 16) | (pos - blocks[k][1])
    bins = [dict() for _ in refs]
    lin = [dict() for _ in refs]
    for off, ln, tid, beg, end in recs:
        if tid < 0:
            continue
        v0, v1 = voff(off), voff(off + ln)
        bins[tid].setdefault(reg2bin(beg, end), []).append((v0, v1))
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            if w not in lin[tid] or v0 < lin[tid][w]:
                lin[tid][w] = v0
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for tid in range(len(refs)):
        bai += struct.pack("<i", len(bins[tid]))
        for b, chunks in sorted(bins[tid].items()):
            bai += struct.pack("<Ii", b, len(chunks))
            for c in chunks:
                bai += struct.pack("<QQ", *c)
        n_intv = (max(lin[tid]) + 1) if lin[tid] else 0
        bai += struct.pack("<i", n_intv)
        prev = 0
        for w in range(n_intv):
            prev = lin[tid].get(w, prev)          # empty windows inherit the previous offset (htslib fills them so)
            bai += struct.pack("<Q", prev)
    with open(path + ".bai", "wb") as fh:
        fh.write(bai)
