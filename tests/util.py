"""Shared helpers for the tests (fixtures loading, flat layouts, tolerances)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity bar of BASELINE.json north_star: class probabilities within 1e-5 relative (fp32).
PROB_RTOL = 1e-5
# logits are O(0.1..10); an absolute bar that implies the probability bar
LOGIT_ATOL = 5e-6


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out, sds = {}, {}
    for k in z.files:
        if "/" in k:
            a, b = k.split("/", 1)
            sds.setdefault(a, {})[b] = z[k]
        else:
            out[k] = z[k]
    out.update(sds)
    return out


def flat(sd, keys):
    return np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k in keys])


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


def softmax_np(x):
    x = np.asarray(x, np.float64)
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def make_curves(seed, n, T, D, min_len=3):
    g = np.random.default_rng(seed)
    lengths = g.integers(min_len, T + 1, size=n).astype(np.int32)
    lengths[g.integers(0, n)] = T
    x = g.standard_normal((T, n, D)).astype(np.float32)
    for b in range(n):
        x[lengths[b]:, b] = 0
    return x, lengths
