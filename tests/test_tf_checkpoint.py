"""TensorFlow-V2 checkpoint codec: round trips and structural checks (no TensorFlow-written
file exists in this image, so the format follows the published bundle layout: parity unpinned)."""
import os
import struct

import numpy as np
import pytest

from clairvoyante_amd import tf_checkpoint as ck


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert ck.crc32c(b"\x00" * 32) == 0x8a9136aa
    assert ck.crc32c(b"\xff" * 32) == 0x62a8ab43
    assert ck.crc32c(bytes(range(32))) == 0x46dd794e
    assert ck.crc32c(b"123456789") == 0xe3069283
    assert ck.mask_crc(0) == 0xa282ead8


def test_bundle_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    t = {"conv1/kernel": rng.standard_normal((1, 4, 4, 16)).astype(np.float32),
         "fc4/kernel": rng.standard_normal((4608, 336)).astype(np.float32),
         "fc4/kernel/Adam": np.zeros((4608, 336), np.float32),
         "beta1_power": np.array(0.9 ** 5, dtype=np.float32)}
    for i in range(60):
        t["pad/var_%03d" % i] = rng.standard_normal((3, i + 1)).astype(np.float32)
    prefix = str(tmp_path / "model-000001")
    ck.write_bundle(prefix, t)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(a.nbytes for a in t.values())
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == ck.MAGIC
    items = ck.read_table(prefix + ".index")
    keys = [k for k, _ in items]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) == len(t) + 1
    back = ck.read_bundle(prefix)
    assert set(back) == set(t)
    for k in t:
        assert back[k].shape == t[k].shape and np.array_equal(back[k], t[k])


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    ck.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32)})
    p = prefix + ".data-00000-of-00001"
    b = bytearray(open(p, "rb").read()); b[17] ^= 1
    open(p, "wb").write(bytes(b))
    with pytest.raises(ValueError):
        ck.read_bundle(prefix)
    with pytest.raises(ValueError):
        open(prefix + ".index", "wb").write(b"not a table" * 10)
        ck.read_table(prefix + ".index")
