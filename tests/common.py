"""Shared helpers of the parity tests (seeded weights + inputs, comparison metrics)."""
import numpy as np

HEADS = ((0, 4), (4, 6), (6, 10), (10, 16))
DEFAULT_VARIANT = 2031     # cv_create's default kernel selection (include/clairvoyante_amd.h, option "variant")


def bench_params(oracle, arch, seed=1):
    """The seeded weight set of the bench line and the parity tests (clairvoyante_amd/synth.py: needs nothing from
    oracle/; the first argument is kept for the call sites that pass the oracle module)."""
    from clairvoyante_amd import synth
    return synth.bench_params(arch, seed=seed)


def inputs(n, seed=5, stress=0):
    from clairvoyante_amd import synth
    x = synth.make_candidates(n, seed=seed).numpy()
    if stress:
        x = np.concatenate([x, synth.make_stress(stress, seed=seed).numpy() * np.float32(0.25)])
    return np.ascontiguousarray(x, dtype=np.float32)


def argmax_match(a, b):
    """per-head fraction of identical argmax (np.argmax semantics) between two [n,16] arrays"""
    return [float(np.mean(np.argmax(a[:, lo:hi], 1) == np.argmax(b[:, lo:hi], 1))) for lo, hi in HEADS]


def bitwise_frac(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.uint32)
    return float(np.mean(a == b))
