"""CPU oracle for the SuperNNova RNN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product (supernnova_amd/) never
does.  See oracle/snn_oracle.c for the restated algorithm and the reference
file:line each function follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OracleCfg(ctypes.Structure):
    _fields_ = [
        ("input_size", ctypes.c_int32),
        ("hidden_size", ctypes.c_int32),
        ("num_layers", ctypes.c_int32),
        ("bidirectional", ctypes.c_int32),
        ("nb_classes", ctypes.c_int32),
        ("pool", ctypes.c_int32),
        ("cell", ctypes.c_int32),
    ]


def build():
    """Compile the C restatement (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_libs = {}
_fp32_name = "libsnn_oracle.so"


def use_library(name):
    """Route the fp32 entry points through another build of the same source (tests/test_host_sanitizers.py:
    libsnn_oracle_asan.so, built by `make -C oracle asan`)."""
    global _fp32_name
    _fp32_name = name


def _lib(f64=False):
    name = "libsnn_oracle_f64.so" if f64 else _fp32_name
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        cp = ctypes.POINTER(OracleCfg)
        lib.snn_oracle_param_count.restype = ctypes.c_int64
        lib.snn_oracle_param_count.argtypes = [cp]
        lib.snn_oracle_feat.restype = ctypes.c_int64
        lib.snn_oracle_feat.argtypes = [cp]
        lib.snn_oracle_forward.restype = ctypes.c_int
        lib.snn_oracle_forward.argtypes = [cp, fp, ctypes.c_int, ctypes.c_int, ip, fp,
                                           fp, fp, fp, ctypes.c_int, fp, fp, fp, fp]
        lib.snn_oracle_bbb_sample.restype = None
        lib.snn_oracle_bbb_sample.argtypes = [ctypes.c_int64, fp, fp, fp, fp, fp]
        lib.snn_oracle_kl.restype = ctypes.c_float
        lib.snn_oracle_kl.argtypes = [cp, fp, fp, fp] + [ctypes.c_double] * 5
        _libs[name] = lib
    return _libs[name]


def _f(a):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


CELLS = {"lstm": 0, "gru": 1, "rnn": 2}


def make_cfg(input_size, hidden_size, num_layers=2, bidirectional=True, nb_classes=2, pool="mean", cell="lstm"):
    return OracleCfg(int(input_size), int(hidden_size), int(num_layers), int(bool(bidirectional)),
                     int(nb_classes), 0 if pool == "mean" else 1, CELLS[cell])


def param_count(cfg):
    return int(_lib().snn_oracle_param_count(ctypes.byref(cfg)))


def feat(cfg):
    return int(_lib().snn_oracle_feat(ctypes.byref(cfg)))


def forward(cfg, x, lengths, w, mask_in=None, mask_whh=None, mask_mid=None, mask_out=None, f64=False):
    """x (T,B,D) padded time-major; lengths (B,); w canonical flat (P,).

    Returns dict(logits (B,C), probs (B,C), pooled (B,F)).
    """
    x = _c32(x)
    T, B, D = x.shape
    assert D == cfg.input_size
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    assert lengths.shape == (B,)
    w = _c32(w)
    assert w.shape == (param_count(cfg),), (w.shape, param_count(cfg))
    C, F = cfg.nb_classes, feat(cfg)
    dirs = 2 if cfg.bidirectional else 1
    mask_in, mask_whh, mask_mid, mask_out = map(_c32, (mask_in, mask_whh, mask_mid, mask_out))
    mid_T = 1
    if mask_in is not None:
        assert mask_in.shape == (B, D)
    if mask_whh is not None:
        assert mask_whh.shape == (cfg.num_layers, 4 * cfg.hidden_size)
    if mask_mid is not None:
        if mask_mid.ndim == 3:
            mask_mid = mask_mid[:, None]
        mid_T = mask_mid.shape[1]
        assert mid_T in (1, T)
        assert mask_mid.shape == (cfg.num_layers - 1, mid_T, B, dirs * cfg.hidden_size)
        mask_mid = np.ascontiguousarray(mask_mid)
    if mask_out is not None:
        assert mask_out.shape == (B, F)
    logits = np.zeros((B, C), np.float32)
    probs = np.zeros((B, C), np.float32)
    pooled = np.zeros((B, F), np.float32)
    rc = _lib(f64).snn_oracle_forward(
        ctypes.byref(cfg), _f(x), T, B, lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _f(w),
        _f(mask_in), _f(mask_whh), _f(mask_mid), mid_T, _f(mask_out), _f(logits), _f(probs), _f(pooled))
    if rc != 0:
        raise RuntimeError(f"snn_oracle_forward failed: {rc}")
    return {"logits": logits, "probs": probs, "pooled": pooled}


def bbb_sample(mu, rho, eps=None):
    """w = mu + eps*(softplus(rho)+1e-5) (eps None: w = mu). Returns (w, sigma)."""
    mu, rho, eps = _c32(mu), _c32(rho), _c32(eps)
    n = mu.size
    w = np.zeros(n, np.float32)
    sigma = np.zeros(n, np.float32)
    _lib().snn_oracle_bbb_sample(n, _f(mu), _f(rho), _f(eps), _f(w), _f(sigma))
    return w, sigma


def kl(cfg, w, mu, sigma, pi=0.75, log_sigma1=-1.0, log_sigma2=-7.0,
       log_sigma1_output=-1.0, log_sigma2_output=-7.0):
    """Sum of compute_KL over the 4*L*dirs LSTM tensors + head weight."""
    import math
    w, mu, sigma = _c32(w), _c32(mu), _c32(sigma)
    n = param_count(cfg) - cfg.nb_classes
    assert w.size >= n and mu.size >= n and sigma.size >= n
    return float(_lib().snn_oracle_kl(ctypes.byref(cfg), _f(w), _f(mu), _f(sigma), pi,
                                      math.exp(log_sigma1), math.exp(log_sigma2),
                                      math.exp(log_sigma1_output), math.exp(log_sigma2_output)))
