"""Host only: the block DEFLATE decoder of the native BAM reader (csrc/cv_inflate.cpp) against zlib -- every
compression level and strategy (stored, fixed and dynamic codes, long and short distances, several DEFLATE blocks per
stream, empty and 64 KiB outputs), and malformed input: truncated, bit-flipped and random streams must be rejected or
at worst decode to something else -- never write outside the output buffer."""
import ctypes
import gzip
import os
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module")
def lib():
    from clairvoyante_amd import _lib
    L = _lib.load()
    L.cv_inflate_raw.restype = ctypes.c_int64
    L.cv_inflate_raw.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    return L


def inflate(lib, comp, size, guard=64):
    """-> (return code, output bytes, guard intact)"""
    src = np.frombuffer(comp + b"\xa5" * 8, dtype=np.uint8).copy()          # 8 readable bytes behind the stream
    dst = np.full(size + guard, 0x5A, dtype=np.uint8)
    rc = lib.cv_inflate_raw(src.ctypes.data_as(ctypes.c_void_p), len(comp), dst.ctypes.data_as(ctypes.c_void_p), size)
    return rc, dst[:size].tobytes(), bool((dst[size:] == 0x5A).all())


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if flush_every is None:
        return co.compress(data) + co.flush()
    out = b""
    for s in range(0, len(data), flush_every):
        out += co.compress(data[s:s + flush_every]) + co.flush(zlib.Z_FULL_FLUSH if (s // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + co.flush()


def corpora():
    rng = np.random.RandomState(5)
    bam_like = b"".join(rng.bytes(4) + b"\x00" * 3 + bytes([rng.randint(60)]) + b"readname%05d\x00" % i
                        + rng.choice(np.frombuffer(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88", np.uint8), 75).tobytes()
                        + b"\xff" * 150 for i in range(260))
    return {
        "empty": b"", "one": b"x", "two": b"ab", "zeros": b"\0" * 65536, "run": b"abcabcabc" * 5000,
        "random": rng.bytes(65536), "random_small": rng.bytes(300),
        "text": (b"the quick brown fox jumps over the lazy dog. " * 1500)[:65536],
        "skewed": rng.choice(np.frombuffer(b"ACGTN\n", np.uint8), 65536, p=[.3, .2, .2, .28, .01, .01]).tobytes(),
        "bam_like": bam_like[:65536],
        "far_matches": (rng.bytes(30000) + rng.bytes(2000) * 3 + rng.bytes(100))[:65536] + b"",
        "all_bytes": bytes(range(256)) * 200,
    }


@pytest.mark.parametrize("name", sorted(corpora()))
def test_decoder_equals_zlib(lib, name):
    data = corpora()[name]
    streams = [raw_deflate(data, lvl) for lvl in range(0, 10)]
    streams += [raw_deflate(data, 6, st) for st in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED)]
    streams += [raw_deflate(data, 5, flush_every=f) for f in (1000, 7777)] if len(data) > 2000 else []
    for comp in streams:
        assert zlib.decompress(comp, -15) == data
        rc, out, intact = inflate(lib, comp, len(data))
        assert intact
        assert rc == len(data) and out == data


def test_wrong_sizes_and_truncation_are_rejected(lib):
    data = corpora()["text"]
    comp = raw_deflate(data, 6)
    for size in (len(data) - 1, len(data) + 1, 0, 10):
        rc, _out, intact = inflate(lib, comp, size)
        assert rc == -1 and intact
    for cut in (1, 2, 5, len(comp) // 2, len(comp) - 1):
        rc, _out, intact = inflate(lib, comp[:cut], len(data))
        assert intact and rc == -1


def test_garbage_never_writes_outside_the_buffer(lib):
    rng = np.random.RandomState(9)
    data = corpora()["bam_like"]
    comp = bytearray(raw_deflate(data, 6))
    ok = 0
    for trial in range(3000):
        c = bytearray(comp)
        for _ in range(rng.randint(1, 4)):
            c[rng.randint(len(c))] ^= 1 << rng.randint(8)
        rc, out, intact = inflate(lib, bytes(c), len(data))
        assert intact
        ok += rc == len(data) and out == data
    assert ok < 3000                                        # most flips break the stream or change the bytes
    for trial in range(2000):
        c = rng.bytes(rng.randint(1, 400))
        rc, _out, intact = inflate(lib, c, rng.randint(0, 70000))
        assert intact


def test_crc32_equals_zlib(lib):
    rng = np.random.RandomState(3)
    for n in (0, 1, 7, 15, 16, 17, 31, 33, 1000, 65536):
        d = rng.bytes(n)
        a = np.frombuffer(d + b"\0", dtype=np.uint8).copy()
        assert lib.cv_crc32_ieee(0, a.ctypes.data_as(ctypes.c_void_p), n) == (zlib.crc32(d) & 0xffffffff)
    d = rng.bytes(5000)                                      # running value over pieces
    a = np.frombuffer(d, dtype=np.uint8).copy()
    c = lib.cv_crc32_ieee(0, a.ctypes.data_as(ctypes.c_void_p), 1234)
    c = lib.cv_crc32_ieee(c, ctypes.c_void_p(a.ctypes.data + 1234), 5000 - 1234)
    assert c == (zlib.crc32(d) & 0xffffffff)


def test_decoder_equals_zlib_on_generated_inputs(lib):
    """property test: byte strings built from random literals, repeats at random distances and runs, compressed with
    a random level / strategy / flush pattern"""
    from hypothesis import given, settings, strategies as st

    piece = st.one_of(
        st.binary(min_size=0, max_size=300),
        st.tuples(st.integers(1, 40000), st.integers(3, 600)),          # (distance, length) copy from what is there
        st.tuples(st.integers(0, 255), st.integers(1, 3000)).map(lambda t: bytes([t[0]]) * t[1]),
    )

    @settings(max_examples=150, deadline=None)
    @given(st.lists(piece, min_size=0, max_size=40), st.integers(0, 9),
           st.sampled_from([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]),
           st.sampled_from([None, 97, 4000]))
    def check(pieces, level, strategy, flush):
        buf = bytearray()
        for p in pieces:
            if isinstance(p, tuple):
                d, n = p
                if len(buf) == 0:
                    continue
                d = min(d, len(buf))
                for _ in range(n):
                    buf.append(buf[-d])
            else:
                buf += p
            if len(buf) > 65536:
                break
        data = bytes(buf[:65536])
        comp = raw_deflate(data, level, strategy, flush_every=flush if len(data) > 0 else None)
        rc, out, intact = inflate(lib, comp, len(data))
        assert intact and rc == len(data) and out == data
    check()


# ---- the same decoder streaming over gzip files too large to inflate at once (utils_v2._GzipFile) --------------------

def _read_all(g, piece):
    out = []
    while True:
        c = g.read(piece)
        if not c:
            return b"".join(out)
        out.append(c)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_gzip_file_reader_equals_gzip(tmp_path, level, monkeypatch):
    """utils_v2._GzipFile (memory-mapped file, RFC 1952 header by hand, cv_inflate_stream block by block into a window buffer,
    CRC-32 + length of every member checked) hands out the bytes `gzip -dc` does: small and empty files, stored blocks
    (incompressible data), a file whose output crosses several decoder calls, several members, zero padding at the end"""
    from clairvoyante_amd import utils_v2
    monkeypatch.setattr(utils_v2._GzipFile, "WANT", 1 << 20)          # many decoder calls on modest data
    monkeypatch.setattr(utils_v2._GzipFile, "CAP", 32768 + (1 << 20) + (8 << 20))
    rng = np.random.RandomState(level)
    row = b"chr1 12345 ACGTACGTACGTACGTACGTACGTACGTACGTA " + b" ".join(b"%d.0" % v for v in rng.randint(0, 60, 528)) + b"\n"
    texts = [b"", b"x", b"hello\nworld\n" * 1000, rng.bytes(700000),
             b"".join(row[:40] + b" ".join(b"%d.0" % v for v in rng.randint(0, 60, 528)) + b"\n" for _ in range(3000))]
    for i, t in enumerate(texts):
        fn = str(tmp_path / ("t%d.gz" % i))
        with gzip.open(fn, "wb", compresslevel=level) as f:
            f.write(t)
        g = utils_v2._GzipFile(fn)
        assert _read_all(g, 333333) == t
        g.close()
    multi = str(tmp_path / "multi.gz")
    with open(multi, "wb") as f:
        for t in texts[2:]:
            f.write(gzip.compress(t, compresslevel=level))
        f.write(b"\0" * 11)
    g = utils_v2._GzipFile(multi)
    assert _read_all(g, 1 << 22) == b"".join(texts[2:])
    g.close()


def test_gzip_file_reader_refuses_what_it_cannot_vouch_for(tmp_path, monkeypatch):
    """damage found before the first byte is handed out sends the file to the external gzip (_GzipFallback); damage found
    later raises CvError -- a call set must not end early in silence; GetTensor gives the rows of the reference's pipe
    through either decoder"""
    from clairvoyante_amd import _lib, utils_v2
    rng = np.random.RandomState(3)
    text = b"".join(b"line %d " % i + rng.bytes(40).hex().encode() + b"\n" for i in range(60000))
    good = gzip.compress(text, compresslevel=6)
    bad_crc = bytearray(good); bad_crc[-6] ^= 0xff
    fn = str(tmp_path / "bad_crc.gz"); open(fn, "wb").write(bytes(bad_crc))
    with pytest.raises(utils_v2._GzipFallback):
        utils_v2._GzipFile(fn).read(10)
    monkeypatch.setattr(utils_v2._GzipFile, "WANT", 1 << 18)
    g = utils_v2._GzipFile(fn)                       # with small pieces the first bytes leave before the trailer is seen
    assert g.read(1000) == text[:1000]
    with pytest.raises(_lib.CvError, match="broke off"):
        _read_all(g, 1 << 20)
    cut = str(tmp_path / "cut.gz"); open(cut, "wb").write(good[:len(good) // 2])
    g = utils_v2._GzipFile(cut)
    with pytest.raises((_lib.CvError, utils_v2._GzipFallback)):
        _read_all(g, 1 << 20)
    notgz = str(tmp_path / "plain.txt.gz"); open(notgz, "wb").write(text[:5000])
    with pytest.raises(utils_v2._GzipFallback):
        utils_v2._GzipFile(notgz)
    # the reader GetTensor uses: the decoder's doubts end in the reference's pipe, at the byte the caller had reached
    f = utils_v2._GzipOrPipe(fn)                                   # bad CRC: gzip itself writes the data, then reports the error
    assert _read_all(f, 1 << 18) == text
    with pytest.raises(_lib.CvError, match="exit status 1"):
        f.close()
    f = utils_v2._GzipOrPipe(notgz)                                # plain text under a .gz name: `gzip -fdc` passes it through
    assert _read_all(f, 999) == text[:5000]
    f.close()
    whole = str(tmp_path / "good.gz"); open(whole, "wb").write(good)
    monkeypatch.setattr(utils_v2._GzipFile, "CAP", 32768 + (1 << 18) + 4096)      # a window too small for a block: hand-over mid-stream
    f = utils_v2._GzipOrPipe(whole)
    assert _read_all(f, 77777) == text
    f.close()
    monkeypatch.undo()
    # GetTensor: same batches through the in-process decoder and through the gzip child process
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    src = os.path.join(G, "gettensor_b.txt.gz")

    def rows():
        return [(e, c, np.array(x), list(p)) for e, c, x, p in utils_v2.GetTensor(src, 43, log=False)]
    a = rows()
    monkeypatch.setenv("CV_GZIP", "external")
    b = rows()
    assert len(a) == len(b) and all(u[0] == v[0] and u[1] == v[1] and np.array_equal(u[2], v[2]) and u[3] == v[3] for u, v in zip(a, b))
