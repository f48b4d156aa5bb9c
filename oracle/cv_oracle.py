"""ctypes front-end of the CPU ORACLE (oracle/cv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under clairvoyante_amd/ imports this.
Parity status: *parity unpinned by the reference* (see cv_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcv_oracle.so")

H, W, CIN, KW, NOUT = 33, 4, 4, 4, 16


class Arch(ctypes.Structure):
    _fields_ = [("kh", ctypes.c_int * 3), ("cout", ctypes.c_int * 3), ("pool", ctypes.c_int * 3),
                ("fc4", ctypes.c_int), ("fc5", ctypes.c_int)]


def _arch(kh, cout, pool, fc4, fc5):
    a = Arch()
    a.kh[:] = kh
    a.cout[:] = cout
    a.pool[:] = pool
    a.fc4, a.fc5 = fc4, fc5
    return a


# clairvoyante_v3.py:9-12 / clairvoyante_v3_slim.py:9-11
ARCH = {
    "full": _arch((1, 2, 3), (16, 32, 48), (5, 4, 3), 336, 168),
    "slim": _arch((1, 3, 5), (8, 16, 32), (1, 1, 1), 36, 18),
}

# TF variable names, in the order of the C parameter array
PARAM_NAMES = [
    "conv1/kernel", "conv1/bias", "conv2/kernel", "conv2/bias", "conv3/kernel", "conv3/bias",
    "fc4/kernel", "fc4/bias", "fc5/kernel", "fc5/bias",
    "YBaseChangeSigmoid/kernel", "YBaseChangeSigmoid/bias",
    "YZygosityFC/kernel", "YZygosityFC/bias",
    "YVarTypeFC/kernel", "YVarTypeFC/bias",
    "YIndelLengthFC/kernel", "YIndelLengthFC/bias",
]


def build(force=False):
    """Compile the oracle with oracle/Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cv_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        flags = open("/proc/cpuinfo").read()
        if " fma" not in flags or " avx2" not in flags:
            raise RuntimeError("cv_oracle is built with -mavx2 -mfma; this CPU lacks them")
        _lib = ctypes.CDLL(build())
        _lib.cvo_expf.restype = ctypes.c_float
        _lib.cvo_expf.argtypes = [ctypes.c_float]
        _lib.cvo_selu_scalar.restype = ctypes.c_float
        _lib.cvo_selu_scalar.argtypes = [ctypes.c_float]
        _lib.cvo_sigmoid_scalar.restype = ctypes.c_float
        _lib.cvo_sigmoid_scalar.argtypes = [ctypes.c_float]
        _lib.cvo_flat_size.restype = ctypes.c_int
        _lib.cvo_record_size.restype = ctypes.c_int64
        _lib.cvo_loss_grad.restype = ctypes.c_double
    return _lib


def param_shapes(arch_name):
    a = ARCH[arch_name]
    cin = [CIN, a.cout[0], a.cout[1]]
    h = H
    for l in range(3):
        h -= a.pool[l] - 1
    flat = h * W * a.cout[2]
    shapes = []
    for l in range(3):
        shapes.append((a.kh[l], KW, cin[l], a.cout[l]))
        shapes.append((a.cout[l],))
    shapes += [(flat, a.fc4), (a.fc4,), (a.fc4, a.fc5), (a.fc5,),
               (a.fc4, 4), (4,), (a.fc5, 2), (2,), (a.fc5, 4), (4,), (a.fc5, 6), (6,)]
    return dict(zip(PARAM_NAMES, shapes))


def init_params(arch_name, seed=0, bias_scale=0.0, head_scale=1.0):
    """Reference initialisers: tf.contrib variance_scaling_initializer(factor=2.0,
    FAN_IN, uniform=False) = truncated normal, stddev sqrt(1.3*2/fan_in), cut at 2
    sigma (clairvoyante_v3.py:57,72,87,107,117); heads use tf.layers.dense's default
    glorot_uniform (v3.py:125-135); biases zero (bias_scale>0 perturbs them so that
    fixtures exercise the bias path)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(arch_name).items():
        if name.endswith("bias"):
            out[name] = (bias_scale * rng.standard_normal(shp)).astype(np.float32)
            continue
        fan_in = int(np.prod(shp[:-1]))
        fan_out = shp[-1]
        if name.startswith("Y"):
            lim = np.sqrt(6.0 / (fan_in + fan_out)) * head_scale
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
        else:
            std = np.sqrt(1.3 * 2.0 / fan_in)
            v = rng.standard_normal(shp)
            bad = np.abs(v) > 2.0
            while bad.any():
                v[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(v) > 2.0
            out[name] = (v * std).astype(np.float32)
    return out


def _parr(params):
    arrs = [np.ascontiguousarray(params[n], dtype=np.float32) for n in PARAM_NAMES]
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return arrs, ptrs


def predict(arch_name, params, x, nthreads=0):
    """[n,33,4,4] fp32 -> [n,16] fp32 (base4 | zygosity2 | type4 | length6)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, H, W, CIN)
    n = x.shape[0]
    out = np.empty((n, NOUT), dtype=np.float32)
    arrs, ptrs = _parr(params)
    lib().cvo_predict(ctypes.byref(ARCH[arch_name]), ptrs, ctypes.c_void_p(x.ctypes.data),
                      ctypes.c_int64(n), ctypes.c_void_p(out.ctypes.data), ctypes.c_int(nthreads))
    return out


_REC_FIELDS = ["pre1", "act1", "pool1", "pre2", "act2", "pool2", "pre3", "act3", "pool3",
               "fc4pre", "fc4", "d4", "fc5pre", "fc5", "hpre0", "hpre1", "hpre2", "hpre3", "out"]


def forward_all(arch_name, params, x, mask4=None, rate4=0.0):
    """All intermediates as a dict of [n, ...] arrays (conv maps are [n,h,4,c])."""
    a = ARCH[arch_name]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, H, W, CIN)
    n = x.shape[0]
    off = (ctypes.c_int64 * 20)()
    lib().cvo_record_offsets(ctypes.byref(a), off)
    off = list(off)
    total = off[19]
    recs = np.empty((n, total), dtype=np.float32)
    arrs, ptrs = _parr(params)
    mptr = None
    if mask4 is not None:
        mask4 = np.ascontiguousarray(mask4, dtype=np.float32)
        mptr = ctypes.c_void_p(mask4.ctypes.data)
    lib().cvo_forward_all(ctypes.byref(a), ptrs, ctypes.c_void_p(x.ctypes.data), ctypes.c_int64(n),
                          ctypes.c_void_p(recs.ctypes.data), mptr, ctypes.c_float(rate4))
    res = {}
    for i, f in enumerate(_REC_FIELDS):
        res[f] = recs[:, off[i]:off[i + 1]]
    hc = [H, H - (a.pool[0] - 1), H - (a.pool[0] - 1) - (a.pool[1] - 1)]
    for l in range(3):
        c = a.cout[l]
        for f in ("pre", "act"):
            res["%s%d" % (f, l + 1)] = res["%s%d" % (f, l + 1)].reshape(n, hc[l], W, c)
        res["pool%d" % (l + 1)] = res["pool%d" % (l + 1)].reshape(n, hc[l] - (a.pool[l] - 1), W, c)
    return res


def loss_grad(arch_name, params, x, y, lam=0.0, mask4=None, rate4=0.0, want_grads=True):
    """Returns (loss, losses[5], grads dict|None); sums over the batch (v3.py:140-151)."""
    a = ARCH[arch_name]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, H, W, CIN)
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1, NOUT)
    n = x.shape[0]
    arrs, ptrs = _parr(params)
    losses = (ctypes.c_double * 5)()
    grads, gptrs = None, None
    if want_grads:
        grads = [np.zeros_like(a_) for a_ in arrs]
        gptrs = (ctypes.c_void_p * len(grads))(*[g.ctypes.data for g in grads])
    mptr = None
    if mask4 is not None:
        mask4 = np.ascontiguousarray(mask4, dtype=np.float32)
        mptr = ctypes.c_void_p(mask4.ctypes.data)
    loss = lib().cvo_loss_grad(ctypes.byref(a), ptrs, ctypes.c_void_p(x.ctypes.data),
                               ctypes.c_void_p(y.ctypes.data), ctypes.c_int64(n), ctypes.c_float(lam),
                               mptr, ctypes.c_float(rate4), losses, gptrs)
    gd = dict(zip(PARAM_NAMES, grads)) if want_grads else None
    return float(loss), list(losses), gd


def adam_step(w, m, v, g, lr, t):
    """In-place TF1-Adam update of contiguous fp32 arrays."""
    for a_ in (w, m, v, g):
        assert a_.dtype == np.float32 and a_.flags["C_CONTIGUOUS"]
    lib().cvo_adam_step(ctypes.c_void_p(w.ctypes.data), ctypes.c_void_p(m.ctypes.data),
                        ctypes.c_void_p(v.ctypes.data), ctypes.c_void_p(g.ctypes.data),
                        ctypes.c_int64(w.size), ctypes.c_float(lr), ctypes.c_int(t))


def selu_sweep(lo=0x80000000, hi=0xff800000):
    """(monotonicity violations, checksum) of the canonical SELU over the negative floats with bit patterns
    [lo, hi] -- see cvo_selu_sweep; ~4 s on 8 cores for the whole axis."""
    chk = ctypes.c_uint64()
    f = lib().cvo_selu_sweep
    f.restype = ctypes.c_uint64
    v = f(ctypes.c_uint32(lo), ctypes.c_uint32(hi), ctypes.byref(chk))
    return int(v), int(chk.value)


def expf(x):
    return float(lib().cvo_expf(ctypes.c_float(x)))
