"""The host data plane (csrc/cv_hostio.cpp, cv_inflate.cpp: text-tensor parser, blosc / LZ4 decoder, raw-DEFLATE decoder in
its whole-block and streaming forms, VCF formatter, the host thread pool) under AddressSanitizer + UBSan and under
ThreadSanitizer on the CPU -- tests/native/fuzz_host.cpp feeds valid inputs (checked against what they were made from)
and byte-mutated / truncated copies of them.  SURVEY.md 5 lists no sanitizer run in the reference; GPU sanitizers are not
available on this pool, so the device code is covered by the parity tests and this covers the native host code.
Found when first run (round 6): a block start offset of a corrupt blosc chunk was used unchecked (read before the chunk),
cv_format_vcf sized a copy by an out-of-range decision word, cv_format_tensor_row could step past its buffer after a value
that prints longer than the 16 bytes budgeted."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")


def _build(target):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    r = subprocess.run(["make", "-C", HERE, "-s", target], capture_output=True, text=True)
    if r.returncode != 0 and ("libasan" in r.stderr or "libtsan" in r.stderr or "libubsan" in r.stderr):
        pytest.skip("sanitizer runtime not installed: %s" % r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-2000:]
    return os.path.join(HERE, target)


@pytest.mark.parametrize("seed", [1, 2])
def test_host_data_plane_under_address_and_undefined_behaviour_sanitizers(seed):
    exe = _build("fuzz_host")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "120", str(seed)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_host_thread_pool_under_thread_sanitizer():
    exe = _build("fuzz_host_tsan")
    r = subprocess.run([exe, "48", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_bam_reader_under_address_sanitizer_over_mutated_files(tmp_path):
    """csrc/cv_bam.cpp (BGZF blocks, BAM records, the .bai linear index) over a valid file from tests/bam_writer.py and 120
    damaged copies: bytes of the compressed file changed (mostly caught by the block CRC), bytes of the INFLATED stream
    changed before recompressing (record lengths, name / CIGAR / sequence counts, reference ids: the CRC then matches),
    truncations, and damaged copies of the index."""
    import gzip  # noqa: F401  (zlib-backed: the writer below uses it)
    import random
    import struct
    import zlib
    import numpy as np
    from test_bam_native import sam_records
    import bam_writer
    exe = _build("fuzz_bam")
    recs = sam_records("noisy")
    refs = [("ctgA", 4000), ("other", 10)]
    good = str(tmp_path / "good.bam")
    bam_writer.write_bam(good, recs, refs, block_payload=4093)
    raw = open(good, "rb").read(); bai = open(good + ".bai", "rb").read()
    rng = random.Random(5); files = [good]

    def blocks(data):          # (offset, size) of the BGZF blocks
        out, off = [], 0
        while off + 18 <= len(data):
            bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
            out.append((off, bsize)); off += bsize
        return out
    bl = blocks(raw)
    for k in range(120):
        p = str(tmp_path / ("m%03d.bam" % k)); d = bytearray(raw); idx = bytearray(bai)
        kind = k % 4
        if kind == 0:          # compressed bytes
            for _ in range(rng.randint(1, 4)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:        # inflated bytes, recompressed with a matching CRC
            off, size = bl[rng.randrange(len(bl) - 1)]
            payload = bytearray(zlib.decompress(bytes(d[off + 18:off + size - 8]), -15))
            for _ in range(rng.randint(1, 3)):
                at = rng.randrange(len(payload))
                payload[at:at + 4] = struct.pack("<I", rng.choice([0, 1, 0x7fffffff, 0xffffffff, rng.randrange(1 << 32), rng.randrange(70000)]))[:len(payload) - at]
            blk = bam_writer._bgzf_block(bytes(payload))
            d[off:off + size] = blk
        elif kind == 2:        # truncated
            d = d[:rng.randrange(1, len(d))]
        else:                  # the index
            for _ in range(rng.randint(1, 6)):
                at = rng.randrange(len(idx))
                idx[at:at + 4] = struct.pack("<I", rng.choice([0, 0xffffffff, rng.randrange(1 << 32), rng.randrange(1 << 20)]))[:len(idx) - at]
            if rng.random() < 0.3:
                idx = idx[:rng.randrange(1, len(idx))]
        open(p, "wb").write(bytes(d)); open(p + ".bai", "wb").write(bytes(idx))
        files.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert int(r.stdout.split(" files, ")[1].split(" opened")[0]) >= 2      # (the good file with both thread counts at least)
