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


@pytest.mark.parametrize("t", [0, 1, 7, 829, 985, 1500, 20000, 60000])
def test_adam_step_count_survives_the_fp32_accumulators(t):
    """beta1_power = 0.9^(t+1) underflows in fp32 near t = 985; the step count is read from
    beta2_power, as stored by TF (fp32, multiplied once per step)"""
    from clairvoyante_amd import tf_checkpoint
    b1 = np.float32(0.9); b2 = np.float32(0.999)
    p1 = np.float32(0.9); p2 = np.float32(0.999)
    for _ in range(t):
        p1 = np.float32(p1 * b1); p2 = np.float32(p2 * b2)        # TF's own running products
    assert tf_checkpoint._adam_steps(p1, p2) == t
    # and from the closed form the codec writes itself
    assert tf_checkpoint._adam_steps(np.float32(tf_checkpoint._B1 ** (t + 1)), np.float32(tf_checkpoint._B2 ** (t + 1))) == t
    # an older bundle without beta2_power still works while beta1_power is a normal number
    if t < 600:
        assert tf_checkpoint._adam_steps(p1, None) == t


def test_adam_step_count_after_both_accumulators_underflowed():
    from clairvoyante_amd import tf_checkpoint
    assert tf_checkpoint._adam_steps(np.float32(0.0), np.float32(0.0)) >= 90000
