"""Host logic of the hot path against golden vectors produced by the REFERENCE ITSELF
(tests/golden/make_golden_ref.py: /root/reference/clairvoyante/{callVar,utils_v2}.py imported
through 2to3 in the build container).  CPU only; calls the native host code through the C ABI."""
import io
import os
import pickle
import random
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _args(showRef, qual, ref_fn, sample):
    return types.SimpleNamespace(v2=False, v3=True, showRef=showRef, qual=qual, ref_fn=ref_fn, sampleName=sample)


@pytest.mark.parametrize("tag,showRef,qual,ref,sample", [
    ("a", False, None, None, "SAMPLE"), ("b", True, 20, os.path.join(G, "mini.fa"), "HG001"),
    ("c", False, 150, None, "SAMPLE")])
def test_output_matches_reference_vcf(tag, showRef, qual, ref, sample):
    from clairvoyante_amd import callVar
    d = np.load(os.path.join(G, "output_cases.npz"))
    X = d["X"].astype(np.float32); pos = [str(s) for s in d["pos"]]
    args = _args(showRef, qual, ref, sample)
    fh = io.StringIO()
    callVar.PrintVCFHeader(args, fh)
    n = X.shape[0]
    for s in range(0, n, 100):
        e = min(n, s + 100)
        callVar.Output(args, fh, e - s, X[s:e], pos[s:e], d["base"][s:e], d["z"][s:e], d["t"][s:e], d["l"][s:e])
    want = open(os.path.join(G, "output_%s.vcf" % tag)).read()
    got = fh.getvalue()
    assert got.splitlines() == want.splitlines()


def test_output_rejects_inconsistent_batch():
    from clairvoyante_amd import callVar
    d = np.load(os.path.join(G, "output_cases.npz"))
    with pytest.raises(SystemExit):
        callVar.Output(_args(False, None, None, "S"), io.StringIO(), 5, d["X"][:5].astype(np.float32),
                       list(d["pos"][:5]), d["base"][:4], d["z"][:4], d["t"][:4], d["l"][:4])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gettensor_matches_reference(tag):
    from clairvoyante_amd import utils_v2
    d = np.load(os.path.join(G, "gettensor_%s.npz" % tag))
    ends, nums, Xs, poss = [], [], [], []
    for end, c, x, pos in utils_v2.GetTensor(os.path.join(G, "gettensor_%s.txt.gz" % tag), int(d["num"]), log=False):
        assert x.shape == (c, 33, 4, 4) and x.dtype == np.float32 and len(pos) == c
        ends.append(end); nums.append(c); Xs.append(np.array(x)); poss += list(pos)
    assert ends == list(d["ends"]) and nums == list(d["nums"])
    assert np.array_equal(np.concatenate(Xs), d["X"])
    assert poss == [str(s) for s in d["pos"]]


def test_gettensor_odd_batch_sizes():
    """batch boundaries anywhere in the stream give the same rows"""
    from clairvoyante_amd import utils_v2
    d = np.load(os.path.join(G, "gettensor_a.npz"))
    for num in (1, 7, 43, 44, 1000):
        Xs, poss, ends = [], [], []
        for end, c, x, pos in utils_v2.GetTensor(os.path.join(G, "gettensor_a.txt.gz"), num, log=False):
            Xs.append(np.array(x)); poss += list(pos); ends.append(end)
        assert ends[-1] == 1 and sum(ends) == 1
        assert np.array_equal(np.concatenate(Xs), d["X"]) and poss == [str(s) for s in d["pos"]]


def test_only_what_gzip_passes_through_is_memory_mapped(tmp_path, monkeypatch):
    """`gzip -fdc` (utils_v2.py:25) also decompresses compress (.Z), pack, lzh and single-member zip input: such files go to
    the stream path (where the pipe takes them), never to the text parser as raw bytes; CV_TEXT=stream sends plain text
    there too; a list counts as compressed when ANY of its files is; a stream generator that is dropped early still closes
    its stream (the `gzip` child is waited for)"""
    import subprocess
    from clairvoyante_amd import utils_v2
    for name, head in (("a.Z", b"\x1f\x9d\x90"), ("a.pack", b"\x1f\x1e\x00"), ("a.lzh", b"\x1f\xa0\x00"), ("a.zip", b"PK\x03\x04\x14")):
        fn = str(tmp_path / name)
        open(fn, "wb").write(head + b"\x00" * 64)
        assert utils_v2._map_plain_text(fn) is None and utils_v2.is_compressed(fn)
    plain = str(tmp_path / "plain.txt")
    import gzip
    open(plain, "wb").write(gzip.open(os.path.join(G, "gettensor_a.txt.gz"), "rb").read())
    assert utils_v2._map_plain_text(plain) is not None and not utils_v2.is_compressed(plain)
    want = [np.array(x) for _e, _c, x, _p in utils_v2.GetTensor(plain, 1000, log=False)]
    monkeypatch.setenv("CV_TEXT", "stream")
    assert utils_v2._map_plain_text(plain) is None
    got = [np.array(x) for _e, _c, x, _p in utils_v2.GetTensor(plain, 1000, log=False)]
    assert all(np.array_equal(a, b) for a, b in zip(want, got)) and len(want) == len(got)
    monkeypatch.delenv("CV_TEXT")
    # dropped after the first batch, through the reference's pipe: the child process does not outlive the generator
    monkeypatch.setenv("CV_GZIP", "external")
    started = []
    real = subprocess.Popen

    def spy(*a, **k):
        p = real(*a, **k)
        started.append(p)
        return p
    monkeypatch.setattr(utils_v2.subprocess, "Popen", spy)
    gen = utils_v2.GetTensor(os.path.join(G, "gettensor_a.txt.gz"), 3, log=False)
    next(gen)
    gen.close()
    assert len(started) == 1 and started[0].returncode is not None


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("strip_last_newline", [False, True])
def test_gettensor_plain_text_through_the_memory_map(tag, strip_last_newline, tmp_path):
    """the same rows in an UNCOMPRESSED file (the reference's `gzip -fdc` passes plain text through, utils_v2.py:25):
    read through the memory map instead of a pipe -- the reference's batches, whatever the batch size, also when the
    last line has no newline, and with 1 or several parser threads (the large copy crosses the 1 MiB threshold above
    which cv_parse_tensor_text splits its input)"""
    import gzip
    from clairvoyante_amd import _lib, utils_v2
    d = np.load(os.path.join(G, "gettensor_%s.npz" % tag))
    text = gzip.open(os.path.join(G, "gettensor_%s.txt.gz" % tag), "rb").read()
    if strip_last_newline:
        text = text.rstrip(b"\n")
    fn = str(tmp_path / "plain.txt")
    open(fn, "wb").write(text)
    assert utils_v2._map_plain_text(fn) is not None and utils_v2._map_plain_text(os.path.join(G, "gettensor_a.txt.gz")) is None
    for num in (int(d["num"]), 1, 43, 1000):
        ends, nums, Xs, poss = [], [], [], []
        for end, c, x, pos in utils_v2.GetTensor(fn, num, log=False):
            assert x.shape == (c, 33, 4, 4) and x.dtype == np.float32 and len(pos) == c
            ends.append(end); nums.append(c); Xs.append(np.array(x)); poss += list(pos)
        if num == int(d["num"]):
            assert ends == list(d["ends"]) and nums == list(d["nums"])
        assert ends[-1] == 1 and sum(ends) == 1
        assert np.array_equal(np.concatenate(Xs), d["X"]) and poss == [str(s) for s in d["pos"]]
    # many copies: > 1 MiB, parsed by several threads (slices cut at arbitrary bytes, moved to line starts)
    reps = (3 << 20) // len(text) + 2
    body = text if text.endswith(b"\n") else text + b"\n"
    big = str(tmp_path / "big.txt")
    open(big, "wb").write(body * reps if not strip_last_newline else (body * reps).rstrip(b"\n"))
    lib = _lib.load()
    try:
        for threads in (1, 3, 8):
            lib.cv_set_host_threads(threads)
            for num in (997, 100000):
                got = [np.array(x) for _e, _c, x, _p in utils_v2.GetTensor(big, num, log=False)]
                assert all(len(g) == num for g in got[:-1])
                got = np.concatenate(got)
                assert got.shape[0] == reps * d["X"].shape[0]
                assert np.array_equal(got.reshape((reps,) + d["X"].shape), np.broadcast_to(d["X"], (reps,) + d["X"].shape))
        last = None
        for _e, c, _x, pos in utils_v2.GetTensor(big, 100000, log=False):
            if c:
                last = pos[c - 1]
        assert last == str(d["pos"][-1])
    finally:
        lib.cv_set_host_threads(min(_lib.usable_cores(), 16))


def test_list_of_compressed_files_read_ahead_gives_the_sequential_batches(tmp_path):
    """GetTensorFiles with several reader threads (a list of .gz files, inflated side by side) yields exactly what reading
    the files one after the other yields -- order of files, batch boundaries, rows, positions --, for every rank's share;
    an unreadable file raises where it is consumed, after the batches of the files in front of it"""
    import gzip
    import shutil
    from clairvoyante_amd import utils_v2
    srcs = [os.path.join(G, "gettensor_a.txt.gz"), os.path.join(G, "gettensor_b.txt.gz")]
    files = []
    for i in range(7):
        fn = str(tmp_path / ("f%d.gz" % i))
        if i == 3:
            gzip.open(fn, "wb").close()                       # an empty chunk
        else:
            shutil.copy(srcs[i % 2], fn)
        files.append(fn)

    def flat(it):
        return [(k, c, np.array(X), list(pos)) for k, c, X, pos in it]
    for rank, ws in ((0, 1), (1, 2), (2, 3)):
        for num in (16, 1000):
            want = flat(utils_v2.GetTensorFiles(files, num, rank, ws, readers=1))
            got = flat(utils_v2.GetTensorFiles(files, num, rank, ws, readers=4, depth=1))
            assert [k for k, _c, _x, _p in want] == [k for k, _c, _x, _p in got] and len(want) >= len(files[rank::ws])
            for a, b in zip(want, got):
                assert a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3] == b[3]
    # unordered hand-over: per file the same batches in the same order, an end marker behind each file's last batch
    for readers in (1, 4):
        per_file, ended = {}, []
        for k, c, X, pos in utils_v2.GetTensorFiles(files, 16, 0, 1, readers=readers, ordered=False):
            if c is None:
                ended.append(k)
            else:
                assert k not in ended
                per_file.setdefault(k, []).append((k, c, np.array(X), list(pos)))
        assert sorted(ended) == list(range(len(files)))
        want = flat(utils_v2.GetTensorFiles(files, 16, 0, 1, readers=1))
        got = [b for k in range(len(files)) for b in per_file.get(k, [])]
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3] == b[3]
    assert utils_v2._default_readers(5) >= 1
    # a consumer that stops early leaves no reader stuck (the generator's finally tells them)
    it = utils_v2.GetTensorFiles(files, 16, 0, 1, readers=3, depth=1)
    next(it); it.close()
    broken = files[:2] + [str(tmp_path / "missing.gz")] + files[2:3]
    it = utils_v2.GetTensorFiles(broken, 1000, 0, 1, readers=4)
    seen = []
    with pytest.raises(Exception):
        for k, c, _x, _p in it:
            seen.append(k)
    assert set(seen) <= {0, 1, 2} and {0, 1} <= set(seen)


def test_training_array_matches_reference():
    from clairvoyante_amd import utils_v2
    d = np.load(os.path.join(G, "trainarray.npz"))
    random.seed(1234)       # same seed as the generator: the reference shuffles with `random`
    total, XC, YC, PC = utils_v2.GetTrainingArray(os.path.join(G, "trainarray_tensor.txt.gz"),
                                                 os.path.join(G, "trainarray_var.txt.gz"),
                                                 os.path.join(G, "trainarray.bed.gz"))
    assert total == int(d["total"]) and len(XC) == len(YC) == len(PC) == int(d["nblocks"])
    X, n, e = utils_v2.DecompressArray(XC, 0, total, total)
    Y, _, _ = utils_v2.DecompressArray(YC, 0, total, total)
    P, _, _ = utils_v2.DecompressArray(PC, 0, total, total)
    assert (n, e) == (total, 1)
    assert X.dtype == np.float32 and np.array_equal(X, d["X"])
    assert Y.dtype == np.float64 and np.array_equal(Y, d["Y"])
    assert [str(s) for s in P] == [str(s) for s in d["pos"]]


@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("fn", ["mini.bin", "mini_py2proto.bin"])
def test_bin_file_blocks_written_by_c_blosc(fn, lazy):
    """`.bin` = 4 pickles (tensor2Bin.py:24-28); blocks here were packed by the real c-blosc.  lazy: the block lists
    as offsets into the memory-mapped file (what train.py loads: no private copy per rank) -- same blocks"""
    from clairvoyante_amd import utils_v2
    d = np.load(os.path.join(G, "trainarray.npz"))
    total, XC, YC, PC = utils_v2.LoadBin(os.path.join(G, fn), lazy=lazy)
    assert total == int(d["total"])
    if lazy and fn == "mini_py2proto.bin":
        assert isinstance(XC, list)          # a protocol-0 text pickle: not the plain layout, un-pickled as before
    elif lazy:
        assert isinstance(XC, utils_v2.LazyBlocks) and isinstance(YC, utils_v2.LazyBlocks)
        _t, XE, YE, PE = utils_v2.LoadBin(os.path.join(G, fn))
        enc = lambda b: b.encode("latin1") if isinstance(b, str) else bytes(b)
        assert [bytes(b) for b in XC] == [enc(b) for b in XE] and [bytes(b) for b in PC] == [enc(b) for b in PE]
    X, _, _ = utils_v2.DecompressArray(XC, 0, total, total)
    Y, _, _ = utils_v2.DecompressArray(YC, 0, total, total)
    assert np.array_equal(X, d["X"]) and np.array_equal(Y, d["Y"])
    dc = np.load(os.path.join(G, "decompress.npz"))
    k = 0
    while "m%d" % k in dc:
        st, nm, mx, nn, ef = [int(v) for v in dc["m%d" % k]]
        a, n2, e2 = utils_v2.DecompressArray(XC, st, nm, mx)
        assert (n2, e2) == (nn, ef) and np.array_equal(np.asarray(a, dtype=np.float32), dc["x%d" % k])
        k += 1
    assert k >= 10


def test_lazy_scan_of_a_truncated_bin_gives_up_instead_of_raising(tmp_path):
    """a .bin cut inside a block header / inside a block: the opcode scan returns None (no index of blocks that are not
    all there), so LoadBin(lazy=True) takes the un-pickling path, whose error names the damage"""
    import mmap
    import pickle
    from clairvoyante_amd import utils_v2
    raw = open(os.path.join(G, "mini.bin"), "rb").read()
    with open(os.path.join(G, "mini.bin"), "rb") as fh:
        pickle.load(fh); start = fh.tell()
    whole = utils_v2._scan_block_list(raw, start)
    assert whole is not None and len(whole[0]) >= 1
    first_off, first_len = whole[0][0]
    for cut in (first_off - 2, first_off + first_len // 2, whole[1] - 1, start + 1):
        fn = str(tmp_path / ("cut%d.bin" % cut))
        open(fn, "wb").write(raw[:cut])
        with open(fn, "rb") as fh:
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            assert utils_v2._scan_block_list(mm, start) is None
            mm.close()
        with pytest.raises((EOFError, pickle.UnpicklingError)):
            utils_v2.LoadBin(fn, lazy=True)


def test_blosc_roundtrip_and_edge_sizes():
    from clairvoyante_amd import utils_v2
    rng = np.random.RandomState(3)
    for n, ts in ((0, 4), (1, 4), (63, 4), (64, 4), (1000, 8), (4096, 4), (100003, 4), (5000, 1), (7777, 44)):
        raw = (rng.randint(0, 4, n).astype(np.uint8) * rng.randint(0, 2, n).astype(np.uint8)).tobytes()
        c = utils_v2.blosc_compress(raw, ts)
        assert utils_v2.blosc_decompress(c) == raw
        noise = rng.bytes(n)
        assert utils_v2.blosc_decompress(utils_v2.blosc_compress(noise, ts)) == noise
    a = rng.standard_normal((500, 33, 4, 4)).astype(np.float32).round()
    assert np.array_equal(utils_v2.unpack_array(utils_v2.pack_array(a)), a)
    assert len(utils_v2.pack_array(a)) < a.nbytes // 2


def test_tensor2bin_writes_the_reference_layout(tmp_path):
    """tensor2Bin -> .bin -> LoadBin -> DecompressArray reproduces the reference's arrays"""
    from clairvoyante_amd import tensor2Bin, utils_v2
    d = np.load(os.path.join(G, "trainarray.npz"))
    out = str(tmp_path / "t.bin")
    args = types.SimpleNamespace(tensor_fn=os.path.join(G, "trainarray_tensor.txt.gz"),
                                 var_fn=os.path.join(G, "trainarray_var.txt.gz"),
                                 bed_fn=os.path.join(G, "trainarray.bed.gz"), bin_fn=out)
    random.seed(1234)
    tensor2Bin.Convert(args, utils_v2)
    with open(out, "rb") as fh:
        assert pickle.load(fh) == int(d["total"])          # first pickle is the plain int total
    total, XC, YC, PC = utils_v2.LoadBin(out)
    X, _, _ = utils_v2.DecompressArray(XC, 0, total, total)
    Y, _, _ = utils_v2.DecompressArray(YC, 0, total, total)
    assert np.array_equal(X, d["X"]) and np.array_equal(Y, d["Y"])


def test_text_parser_thread_count_does_not_change_the_rows():
    """cv_parse_tensor_text with 1 and with 6 host threads: same rows, same positions, same counters, also when
    slices drop rows (non-ACGT centre base, malformed rows, blank lines) and when max_rows cuts the text"""
    import ctypes
    from clairvoyante_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(3)
    n = 3000
    vals = rng.randint(0, 60, size=(n, 528)).astype(np.float32)
    lines = []
    for i in range(n):
        seq = "".join("ACGT"[k] for k in rng.randint(0, 4, 33))
        if i % 97 == 5:
            seq = seq[:16] + "N" + seq[17:]                  # dropped: centre base
        if i % 211 == 7:
            seq = seq.lower()                                # kept: upper-cased before the test
        row = "chr%d %d %s " % (i % 3, 1000 + i, seq) + " ".join("%0.1f" % v for v in vals[i])
        if i % 301 == 9:
            row = row.rsplit(" ", 5)[0]                      # malformed: too few values
        lines.append(row)
        if i % 500 == 3:
            lines.append("")
    text = ("\n".join(lines) + "\n").encode()
    assert len(text) > (2 << 20)

    def run(threads, max_rows):
        lib.cv_set_host_threads(threads)
        x = np.full((max_rows, 528), -7, dtype=np.float32); meta = np.zeros((max_rows, 6), dtype=np.int64)
        c = ctypes.c_int64(); r = ctypes.c_int64(); b = ctypes.c_int64()
        _lib.check(lib.cv_parse_tensor_text(text, len(text), max_rows, x.ctypes.data_as(ctypes.c_void_p),
                                            meta.ctypes.data_as(ctypes.c_void_p), ctypes.byref(c), ctypes.byref(r), ctypes.byref(b)))
        return c.value, r.value, b.value, x[:r.value].copy(), meta[:r.value].copy()
    try:
        for max_rows in (n + 100, 1777):
            one = run(1, max_rows)
            six = run(6, max_rows)
            # with a row limit the threaded parser stops after max_rows LINES (fewer rows if some were dropped);
            # what it returns must be the single-threaded rows of exactly the bytes it consumed
            assert six[1] > 1000 and six[0] <= one[0]
            k = six[1]
            assert np.array_equal(one[3][:k], six[3]) and np.array_equal(one[4][:k], six[4])
            if max_rows > n:
                assert one[:3] == six[:3] and one[2] >= 9
    finally:
        lib.cv_set_host_threads(min(_lib.usable_cores(), 16))


def test_blosc_decoder_survives_corrupt_chunks():
    """bit flips and truncations of a chunk written by this build and of one written by the real c-blosc (lz4hc):
    the decoder may reject or return garbage, but never writes outside the destination"""
    import ctypes
    from clairvoyante_amd import _lib, utils_v2
    lib = _lib.load()
    rng = np.random.RandomState(1)
    X = rng.randint(-30, 60, size=(500, 33, 4, 4)).astype(np.float32)
    X[rng.rand(*X.shape) < 0.7] = 0
    _, XC, _, _ = utils_v2.LoadBin(os.path.join(G, "mini.bin"))
    real = XC[0].encode("latin1") if isinstance(XC[0], str) else bytes(XC[0])
    for base in (bytearray(utils_v2.pack_array(X)), bytearray(real)):
        n = lib.cv_blosc_nbytes(bytes(base), len(base))
        out = ctypes.create_string_buffer(n + 64)
        assert lib.cv_blosc_decompress(bytes(base), len(base), out, n) == 0
        for trial in range(400):
            c = bytearray(base)
            for _ in range(rng.randint(1, 6)):
                c[rng.randint(16 if trial % 3 else 0, len(c))] = rng.randint(0, 256)
            if trial % 7 == 0:
                c = c[:rng.randint(1, len(c))]
            lib.cv_blosc_decompress(bytes(c), len(c), out, n)
            assert out.raw[n:n + 64] == b"\0" * 64


def test_decompressarray_fast_path_equals_unpickling():
    """blocks decompressed straight into one array (cv_blosc_unpack_blocks) == un-pickling each block: float32
    tensors, float64 labels, ranges that start / end inside blocks, the short last block, the empty trailing block"""
    from clairvoyante_amd import utils_v2
    rng = np.random.RandomState(2)
    for total in (3000, 2750):
        X = rng.randint(-9, 60, size=(total, 33, 4, 4)).astype(np.float32)
        Y = rng.rand(total, 16)
        XC = [utils_v2.pack_array(X[s:s + 500]) for s in range(0, total + 1, 500)]      # total 3000: last block is empty
        YC = [utils_v2.pack_array(Y[s:s + 500]) for s in range(0, total + 1, 500)]
        for st, nm in ((0, 1000), (250, 1000), (499, 2), (1000, 5000), (total - 3, 10), (0, total), (2500, 500)):
            for arr, blocks in ((X, XC), (Y, YC)):
                got, k, flag = utils_v2.DecompressArray(blocks, st, nm, total)
                want = arr[st:min(st + nm, total)]
                assert k == len(want) and flag == int(st + nm >= total)
                assert got.dtype == arr.dtype and np.array_equal(got, want)
                slow = np.concatenate([utils_v2.unpack_array(b) for b in blocks[st // 500:(st + k - 1) // 500 + 1]])
                assert np.array_equal(slow[st % 500:st % 500 + k], got)


def test_string_blocks_of_varying_width_take_the_generic_path():
    """position keys are '<U..' arrays whose item size differs per block; the one-array fast path
    must not reinterpret a narrower last block with the first block's item size"""
    import numpy as np
    from clairvoyante_amd import param, utils_v2
    bs = param.bloscBlockSize
    wide = np.array(["chr10:%08d" % i for i in range(bs)])                    # '<U14': 56 bytes per item
    narrow = np.array(["c:%d" % (i % 10) for i in range(14 * 10)])            # '<U3': 140 * 12 bytes = 30 * 56
    assert wide.dtype.itemsize == 56 and (narrow.nbytes % 56) == 0
    blocks = [utils_v2.pack_array(wide), utils_v2.pack_array(narrow)]
    total = bs + len(narrow)
    out, num, end = utils_v2.DecompressArray(blocks, 0, total, total)
    assert num == total and end == 1
    assert list(out) == list(wide) + list(narrow)


def _decisions(callVar, X, base, z, t, l):
    """what cv_call_postproc computes on the device, restated with callVar's own NumPy helpers (the checker path)"""
    n = X.shape[0]
    call = np.zeros((n, 8), np.int32); qual = np.zeros((n, 4), np.float32)
    call[:, 0] = np.argmax(t, 1); call[:, 1] = np.argmax(z, 1); call[:, 2] = np.argmax(l, 1)
    order = np.argsort(base, axis=1, kind="stable")[:, ::-1]
    call[:, 3] = order[:, 0]; call[:, 4] = order[:, 1]
    qual[:, 0], qual[:, 1] = callVar._top2_products(t, z, l)
    qual[:, 2] = callVar._depth(X)
    return call, qual


@pytest.mark.parametrize("tag,showRef,qual,ref,sample", [
    ("a", False, None, None, "SAMPLE"), ("b", True, 20, os.path.join(G, "mini.fa"), "HG001"),
    ("c", False, 150, None, "SAMPLE")])
@pytest.mark.parametrize("threads", [1, 5])
def test_native_formatter_matches_reference_vcf(tag, showRef, qual, ref, sample, threads):
    """cv_format_vcf (the product's formatter) against the VCF text the reference's own Output() wrote"""
    from clairvoyante_amd import callVar, _lib
    from clairvoyante_amd.utils_v2 import PosBatch
    d = np.load(os.path.join(G, "output_cases.npz"))
    X = d["X"].astype(np.float32); pos = [str(s) for s in d["pos"]]
    args = _args(showRef, qual, ref, sample)
    call, q = _decisions(callVar, X, d["base"], d["z"], d["t"], d["l"])
    want = [ln for ln in open(os.path.join(G, "output_%s.vcf" % tag)).read().splitlines() if not ln.startswith("#")]
    lib = _lib.load()
    lib.cv_set_host_threads(threads)
    try:
        n = X.shape[0]
        got = callVar.format_records(args, n, X, pos, call, q).decode().splitlines()
        assert got == want
        # positions as the parser delivers them (pieces of a byte buffer), tensors through a row map
        half = n // 2
        pb = PosBatch(pieces=[PosBatch.from_strings(pos[:half]).pieces()[0][2:], PosBatch.from_strings(pos[half:]).pieces()[0][2:]])
        perm = np.random.RandomState(3).permutation(n)
        inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
        got2 = callVar.format_records(args, n, X[perm], pb, call, q, xrow=inv).decode().splitlines()
        assert got2 == want
    finally:
        lib.cv_set_host_threads(min(_lib.usable_cores(), 16))


def test_native_formatter_equals_python_formatter_on_random_decisions():
    """every branch of callVar.py:88-153 (types, lengths, length guesses up to <INS>/<DEL>, depth 0, quality filter,
    lower-case reference) on 20 000 random candidates: cv_format_vcf == _format_record, line for line"""
    from clairvoyante_amd import callVar, _lib
    rng = np.random.RandomState(11)
    n = 20000
    X = rng.randint(-3, 40, size=(n, 33, 4, 4)).astype(np.float32)
    long_ins = rng.rand(n) < 0.15          # long runs that pass the 0.125 rule position after position
    X[long_ins, 17:, :, 1] += 60; X[long_ins, 17:, :, 2] += 60
    short = rng.rand(n) < 0.3
    cut = rng.randint(21, 33, size=n)
    for i in np.flatnonzero(short):
        X[i, cut[i]:, :, 1] = 0; X[i, cut[i]:, :, 2] = 0; X[i, cut[i]:, :, 0] = 50
    zero = rng.rand(n) < 0.02
    X[zero] = 0
    call = np.zeros((n, 8), np.int32)
    call[:, 0] = rng.randint(0, 4, n); call[:, 1] = rng.randint(0, 2, n); call[:, 2] = rng.randint(0, 6, n)
    call[:, 3] = rng.randint(0, 4, n); call[:, 4] = (call[:, 3] + rng.randint(1, 4, n)) % 4
    q = np.zeros((n, 4), np.float32)
    q[:, 0] = rng.rand(n).astype(np.float32); q[:, 1] = (q[:, 0] * rng.rand(n)).astype(np.float32)
    q[rng.rand(n) < 0.01, 1] = 0
    q[:, 2] = callVar._depth(X)
    seqs = ["".join(rng.choice(list("ACGTacgt"), 33)) for _ in range(n)]
    pos = ["chr%d:%d:%s" % (rng.randint(1, 23), rng.randint(1, 250000000), s) for s in seqs]
    for showRef, qual in ((False, None), (True, 30)):
        args = _args(showRef, qual, None, "S")
        want = []
        for i in range(n):
            if call[i, 0] == 0 and not showRef:
                continue
            p = pos[i].split(":")
            rec = callVar._format_record(args, X[i], "%s:%s:%s" % (p[0], p[1], p[2].upper()), int(call[i, 0]), int(call[i, 1]),
                                         int(call[i, 2]), int(call[i, 3]), int(call[i, 4]), callVar._qual(q[i, 0], q[i, 1]), q[i, 2])
            if rec is not None:
                want.append(rec)
        got = callVar.format_records(args, n, X, pos, call, q).decode().splitlines()
        assert len(got) == len(want) and got == want


def test_native_formatter_rejects_malformed_positions():
    from clairvoyante_amd import callVar, _lib
    X = np.ones((1, 33, 4, 4), np.float32)
    call = np.array([[1, 0, 0, 1, 2, 0, 0, 0]], np.int32); q = np.array([[0.9, 0.1, 8, 0]], np.float32)
    with pytest.raises(_lib.CvError):
        callVar.format_records(_args(False, None, None, "S"), 1, X, ["chr1:12x:" + "A" * 33], call, q)
    with pytest.raises(_lib.CvError):
        callVar.format_records(_args(False, None, None, "S"), 1, X, ["chr1:12:ACGT"], call, q)
