"""Shared helpers of the parity tests (seeded weights + inputs, comparison metrics)."""
import numpy as np

HEADS = ((0, 4), (4, 6), (6, 10), (10, 16))
DEFAULT_VARIANT = 1519     # cv_create's default kernel selection (include/clairvoyante_amd.h, option "variant")


def bench_params(oracle, arch, seed=1):
    """Reference-initialiser weights with conv1 scaled by 1/32 so that count-valued inputs give
    O(1) logits (un-scaled He-initialised weights saturate every softmax), and
    non-zero biases so the bias path is exercised."""
    P = oracle.init_params(arch, seed=seed, bias_scale=0.05)
    P["conv1/kernel"] = (P["conv1/kernel"] * np.float32(1.0 / 32.0)).astype(np.float32)
    return P


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
