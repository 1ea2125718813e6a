"""Shared helpers for golden fixtures: deterministic synthetic inputs + compact storage.

``synth`` makes pseudo-random arrays from pure integer arithmetic (splitmix64 on the
flat index), so generator (oracle/make_golden.py, build container) and tests (any
machine) rebuild bit-identical inputs without storing them.  Large expected outputs
are stored as a strided sample + fp64 norm/sum (``pack``); ``expect`` re-applies the
same sampling to a candidate tensor so callers compare like with like.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_LIMIT = 40000          # arrays up to this many elements are stored whole
SAMPLE_TARGET = 20000       # otherwise about this many strided samples


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_uniform(shape, seed: int) -> np.ndarray:
    """float64 uniform [0,1) array, value = hash(seed, flat index)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
    bits = _splitmix64(idx) >> np.uint64(11)
    return (bits.astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def synth(shape, seed: int, lo=None, hi=None) -> np.ndarray:
    """fp32 array: uniform [lo,hi) if bounds given, else unit-variance bell-shaped
    (sum of four uniforms, centred and scaled)."""
    if lo is not None:
        return (synth_uniform(shape, seed) * (hi - lo) + lo).astype(np.float32)
    acc = sum(synth_uniform(shape, seed * 4 + i) for i in range(4))
    return ((acc - 2.0) * np.sqrt(3.0)).astype(np.float32)


def pack(store: dict, name: str, arr) -> None:
    """Add ``arr`` to ``store`` whole (small) or as sample+norm+sum (large)."""
    a = np.asarray(arr)
    if a.size <= FULL_LIMIT:
        store[name] = a
        return
    stride = max(1, a.size // SAMPLE_TARGET)
    flat = a.reshape(-1)
    store[name + "@stride"] = np.int64(stride)
    store[name + "@shape"] = np.asarray(a.shape, dtype=np.int64)
    store[name + "@sample"] = flat[::stride].copy()
    store[name + "@norm"] = np.float64(np.linalg.norm(flat.astype(np.float64)))
    store[name + "@sum"] = np.float64(flat.astype(np.float64).sum())


def expect(fix, name: str, candidate):
    """Return (expected, got) arrays to compare for fixture entry ``name``; for
    sampled entries ``got`` is the candidate under the same strided sampling and the
    shape is verified.  Also returns (norm_expected, norm_got) or (None, None)."""
    cand = np.asarray(candidate)
    if name in fix:
        exp = fix[name]
        assert tuple(exp.shape) == tuple(cand.shape), (name, exp.shape, cand.shape)
        return exp, cand, None, None
    stride = int(fix[name + "@stride"])
    shape = tuple(int(v) for v in fix[name + "@shape"])
    assert shape == tuple(cand.shape), (name, shape, cand.shape)
    flat = cand.reshape(-1)
    return (fix[name + "@sample"], flat[::stride],
            float(fix[name + "@norm"]), float(np.linalg.norm(flat.astype(np.float64))))


def rel_l2(exp, got) -> float:
    e = np.asarray(exp, dtype=np.float64)
    g = np.asarray(got, dtype=np.float64)
    den = np.linalg.norm(e)
    return float(np.linalg.norm(e - g) / (den if den > 0 else 1.0))


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def list_cases(prefixes):
    names = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))
    return [n for n in names if n.startswith(tuple(prefixes))]


def synth_params(shapes, seed: int) -> dict:
    """Deterministic parameter set with the reference's init scales (pit.py:18-19,35):
    lmda ~ U[0,1), weights ~ bell(0, sqrt(2/fan_in)), biases ~ U(+-1/sqrt(fan_in))."""
    out = {}
    fan_in = 1
    for i, (name, shape) in enumerate(shapes):
        s = seed * 1000 + i
        if name.endswith("lmda"):
            out[name] = synth(shape, s, 0.0, 1.0)
        elif name.endswith("weight"):
            fan_in = shape[1]
            out[name] = (synth(shape, s) * np.float32(np.sqrt(2.0 / fan_in))).astype(np.float32)
        else:
            b = 1.0 / np.sqrt(fan_in)
            out[name] = synth(shape, s, -b, b)
    return out
