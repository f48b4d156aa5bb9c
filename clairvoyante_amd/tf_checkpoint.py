"""TensorFlow "V2" checkpoint (tensor bundle) reader / writer without TensorFlow.

The reference saves and restores with `tf.train.Saver()` over ALL global variables
(/root/reference/clairvoyante/clairvoyante_v3.py:243-251): files
`<prefix>.index`, `<prefix>.data-00000-of-00001`, `<prefix>.meta` (README.md:304), epoch
files `<prefix>-%06d` (train.py:127-129).  Variables: the 18 trainable tensors
(jupyter_nb/visualization.ipynb:103-120) in TF layouts, their Adam slots `<name>/Adam`
and `<name>/Adam_1`, and the scalars `beta1_power`, `beta2_power`.

TensorFlow itself is a third-party dependency of the reference (tensorflow==1.12.0,
requirements.txt:1) and is not available here; this module follows the published on-disk
format of its tensor bundle:
  .index  = a LevelDB-style sorted string table: data blocks of prefix-compressed
            (shared, non_shared, value_len, key_delta, value) entries with a restart array,
            each followed by a 5-byte trailer (compression type 0 + masked CRC32C), a
            meta-index block, an index block of (last_key -> BlockHandle) and a 48-byte
            footer ending in the magic 0xdb4775248b80fb57.  Key "" holds BundleHeaderProto
            {num_shards, endianness, version}; every other key is a variable name mapped to
            BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
  .data-00000-of-00001 = the raw little-endian tensor bytes at those offsets.
No TensorFlow-written checkpoint exists in this image, so the codec is validated by
round trips and by structural checks (tests/test_tf_checkpoint.py): "parity unpinned".
"""
import os
import struct

import numpy as np

from . import _lib

MAGIC = 0xdb4775248b80fb57
DT_FLOAT = 1
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


# ---- crc32c (Castagnoli), masked as in leveldb / TF ---------------------------------------
def crc32c(data):
    import ctypes
    lib = _lib.load()
    if isinstance(data, np.ndarray):
        data = np.ascontiguousarray(data)
        return int(lib.cv_crc32c(0, data.ctypes.data_as(ctypes.c_void_p), data.nbytes))
    return int(lib.cv_crc32c(0, bytes(data), len(data)))


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


# ---- protobuf wire helpers ------------------------------------------------------------------
def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """yield (field_number, wire_type, value) of one protobuf message"""
    pos = 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _msg(fn, payload):
    return _varint((fn << 3) | 2) + _varint(len(payload)) + payload


def _int(fn, v):
    return _varint((fn << 3) | 0) + _varint(v)


def _entry_proto(dtype, shape, offset, size, crc):
    shp = b"".join(_msg(2, _int(1, d)) for d in shape)
    out = _int(1, dtype) + _msg(2, shp)
    if offset:
        out += _int(4, offset)
    out += _int(5, size)
    out += _varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for fn, wt, v in _fields(buf):
        if fn == 1: e["dtype"] = v
        elif fn == 2:
            for f2, w2, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, w3, v3 in _fields(v2):
                        if f3 == 1: size = v3
                    e["shape"].append(size)
        elif fn == 3: e["shard_id"] = v
        elif fn == 4: e["offset"] = v
        elif fn == 5: e["size"] = v
        elif fn == 6: e["crc32c"] = struct.unpack("<I", v)[0]
        elif fn == 7: raise ValueError("sliced (partitioned) variables are not supported")
    return e


# ---- sorted string table ------------------------------------------------------------------
def _block(entries, restart_interval=16):
    out = bytearray()
    restarts = []
    last = b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block):
    t = b"\x00"                                   # kNoCompression
    return block + t + struct.pack("<I", mask_crc(crc32c(block + t)))


def _handle(offset, size):
    return _varint(offset) + _varint(size)


def write_table(path, items, block_size=4096):
    """items: list of (key bytes, value bytes) sorted by key"""
    out = bytearray()
    index = []
    cur = []
    cur_bytes = 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        b = _block(cur)
        index.append((cur[-1][0], _handle(len(out), len(b))))
        out.extend(_with_trailer(b))
        cur = []
        cur_bytes = 0

    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = _block([])
    meta_h = _handle(len(out), len(meta))
    out.extend(_with_trailer(meta))
    idx = _block(index, restart_interval=1)
    idx_h = _handle(len(out), len(idx))
    out.extend(_with_trailer(idx))
    footer = meta_h + idx_h
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


def _read_block(buf, offset, size):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
    if mask_crc(crc32c(raw + bytes([ctype]))) != stored:
        raise ValueError("checkpoint index: block checksum mismatch")
    if ctype != 0:
        raise ValueError("checkpoint index: compressed table blocks (type %d) are not supported" % ctype)
    nrest = struct.unpack("<I", raw[-4:])[0]
    end = len(raw) - 4 - 4 * nrest
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _read_varint(raw, pos)
        non_shared, pos = _read_varint(raw, pos)
        vlen, pos = _read_varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]; pos += non_shared
        yield key, raw[pos:pos + vlen]; pos += vlen


def read_table(path):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise ValueError("%s is not a TensorFlow V2 checkpoint index (bad magic)" % path)
    footer = buf[-48:]
    pos = 0
    _, pos = _read_varint(footer, pos); _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos); isz, pos = _read_varint(footer, pos)
    items = []
    for _, handle in _read_block(buf, ioff, isz):
        off, p = _read_varint(handle, 0); sz, p = _read_varint(handle, p)
        items.extend(_read_block(buf, off, sz))
    return items


# ---- bundle ---------------------------------------------------------------------------------
def write_bundle(prefix, tensors):
    """tensors: dict name -> np.ndarray (float32).  Writes prefix.index / .data-00000-of-00001."""
    names = sorted(tensors)
    data_path = prefix + ".data-00000-of-00001"
    items = [(b"", _int(1, 1) + _int(2, 0) + _msg(3, _int(1, 1)))]     # num_shards 1, little endian, producer 1
    offset = 0
    with open(data_path, "wb") as f:
        for n in names:
            a = np.asarray(tensors[n], dtype=np.float32)      # (ascontiguousarray would make scalars 1-d)
            raw = a.tobytes()
            f.write(raw)
            items.append((n.encode(), _entry_proto(DT_FLOAT, a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", items)


def read_bundle(prefix, check_crc=True):
    items = read_table(prefix + ".index")
    out = {}
    shards = {}
    nshards = 1
    for k, v in items:
        if k == b"":
            for fn, wt, val in _fields(v):
                if fn == 1: nshards = val
                if fn == 2 and val != 0: raise ValueError("big-endian checkpoints are not supported")
            continue
        e = _parse_entry(v)
        fn = "%s.data-%05d-of-%05d" % (prefix, e["shard_id"], nshards)
        if fn not in shards:
            shards[fn] = np.memmap(fn, dtype=np.uint8, mode="r") if os.path.getsize(fn) else np.zeros(0, np.uint8)
        raw = np.asarray(shards[fn][e["offset"]:e["offset"] + e["size"]])
        if e["dtype"] not in _DTYPES:
            continue
        if check_crc and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("checkpoint tensor %s: checksum mismatch" % k.decode())
        out[k.decode()] = raw.view(_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ---- model <-> checkpoint -------------------------------------------------------------------
def _flat_to_host(model, which):
    import ctypes
    import torch
    b = torch.empty(model.numParameters, dtype=torch.float32, device=model.device)
    _lib.check(model._lib.cv_flat_copy(model._h, which, ctypes.c_void_p(b.data_ptr()), 0, model._stream()))
    return b.cpu().numpy()


def _host_to_flat(model, which, arr):
    import ctypes
    import torch
    b = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(model.device)
    _lib.check(model._lib.cv_flat_copy(model._h, which, ctypes.c_void_p(b.data_ptr()), 1, model._stream()))
    torch.cuda.synchronize(model.device)


def save_model(model, fn):
    from .model import PARAM_NAMES
    shapes = model.paramShapes()
    flat = [_flat_to_host(model, w) for w in (0, 2, 3)]
    tensors = {}
    off = 0
    for n in PARAM_NAMES:
        sz = int(np.prod(shapes[n]))
        tensors[n] = flat[0][off:off + sz].reshape(shapes[n])
        tensors[n + "/Adam"] = flat[1][off:off + sz].reshape(shapes[n])
        tensors[n + "/Adam_1"] = flat[2][off:off + sz].reshape(shapes[n])
        off += sz
    # TF keeps beta^(t+1) after t steps (initial value beta, multiplied once per step)
    # (fp32 accumulators of the fp32 constants: the powers are taken of float32(0.9) / float32(0.999))
    tensors["beta1_power"] = np.array(_B1 ** (model._adam_t + 1), dtype=np.float32)
    tensors["beta2_power"] = np.array(_B2 ** (model._adam_t + 1), dtype=np.float32)
    d = os.path.dirname(os.path.abspath(fn))
    os.makedirs(d, exist_ok=True)
    write_bundle(fn, tensors)
    # minimal MetaGraphDef {meta_info_def {tensorflow_version}}: existence is what callers test
    # (callVarBam.py:65); the graph itself is rebuilt by the model class, never read from here
    with open(fn + ".meta", "wb") as f:
        f.write(_msg(1, _msg(5, b"clairvoyante_amd (no graph: weights only)")))
    base = os.path.basename(fn)
    with open(os.path.join(d, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


_B1 = float(np.float32(0.9))       # the constants as TF holds them (fp32)
_B2 = float(np.float32(0.999))


def _adam_steps(beta1_power, beta2_power):
    """Optimizer step count t from the two fp32 accumulators TF keeps (beta^(t+1) after t steps).
    beta2_power = 0.999^(t+1) is the one to read: it stays a normal fp32 number until t ~ 87 000, whereas
    beta1_power = 0.9^(t+1) is denormal from t ~ 830 and exactly 0 from t ~ 985 (a resumed run would restart its
    bias correction with lr_t ~ 0.316 lr).  Once beta2_power has underflowed too, both corrections are 1 in TF as
    well, so any large t reproduces it."""
    b1 = float(beta1_power) if beta1_power is not None else None
    b2 = float(beta2_power) if beta2_power is not None else None
    if b2 is not None and 0.0 < b2 < 1.0:
        return max(0, int(round(np.log(b2) / np.log(_B2))) - 1)
    if b2 is not None and b2 == 0.0:
        return 100000
    if b1 is not None and 1e-30 < b1 < 1.0:
        return max(0, int(round(np.log(b1) / np.log(_B1))) - 1)
    if b1 is not None and 0.0 <= b1 <= 1e-30:
        return 100000 if b1 == 0.0 else max(0, int(round(np.log(b1) / np.log(_B1))) - 1)
    return 0


def restore_model(model, fn):
    from .model import PARAM_NAMES
    if not os.path.exists(fn + ".index"):
        raise IOError("checkpoint %s.index not found" % fn)
    t = read_bundle(fn)
    shapes = model.paramShapes()
    missing = [n for n in PARAM_NAMES if n not in t]
    if missing:
        raise ValueError("checkpoint %s lacks variables %s" % (fn, missing))
    for n in PARAM_NAMES:
        if tuple(t[n].shape) != tuple(shapes[n]):
            raise ValueError("checkpoint %s: %s has shape %s, model expects %s" % (fn, n, t[n].shape, shapes[n]))
    w = np.concatenate([t[n].ravel() for n in PARAM_NAMES])
    _host_to_flat(model, 0, w)
    if all((n + "/Adam") in t and (n + "/Adam_1") in t for n in PARAM_NAMES):
        _host_to_flat(model, 2, np.concatenate([t[n + "/Adam"].ravel() for n in PARAM_NAMES]))
        _host_to_flat(model, 3, np.concatenate([t[n + "/Adam_1"].ravel() for n in PARAM_NAMES]))
        model._adam_t = _adam_steps(t.get("beta1_power"), t.get("beta2_power"))
    else:
        _host_to_flat(model, 2, np.zeros_like(w)); _host_to_flat(model, 3, np.zeros_like(w))
        model._adam_t = 0
