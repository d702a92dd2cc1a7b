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


def sampled_gate(mean_b, std_b, mean_a, std_a, S, group, mean_ref2=None):
    """The sampled-mode gate of SURVEY 8d on per-curve MC means / stds of class probabilities, (N, C) each:
    `b` = the run under test, `a` = the reference run (S draws per curve each); curves come in mini-batches of
    `group` that share one weight draw / W_hh row mask, so only one curve per mini-batch enters the KS tests.
      frac_ok      share of curves with |mean_b - mean_a| <= 4*sqrt(2)*std_a/sqrt(S) (+1e-5)      gate: >= 0.999
      bias         pooled mean of (mean_b - mean_a)                                                  gate: |.| < 1e-3
      std_ratio    pooled predictive std of b / of a (last class)                                    gate: within +-10 %
      ks_t         KS p-value of z = (mean_b - mean_a) / (sqrt(2) std_a / sqrt(S)) against Student t(S-1)   gate: > 0.01
      ks_ref       two-sample KS p-value of z (even mini-batches) against the reference's own z of a second,
                   independent reference run `mean_ref2` vs `a` (odd mini-batches): no normality assumption    gate: > 0.01
    """
    from scipy import stats
    mean_b, std_b, mean_a, std_a = (np.asarray(v, np.float64) for v in (mean_b, std_b, mean_a, std_a))
    se = np.sqrt(2.0) * std_a / np.sqrt(S)
    out = {"frac_ok": float((np.abs(mean_b - mean_a) <= 4 * se + 1e-5).mean()),
           "bias": float((mean_b - mean_a).mean()),
           "std_ratio": float(std_b[:, -1].mean() / std_a[:, -1].mean())}
    first = np.arange(0, mean_a.shape[0], group)          # one curve per mini-batch: independent draws
    ok = se[first, -1] > 1e-9
    z = (mean_b[first, -1] - mean_a[first, -1])[ok] / se[first, -1][ok]
    out["ks_t"] = float(stats.kstest(z, "t", args=(S - 1,)).pvalue)
    out["n_ks"] = int(z.size)
    if mean_ref2 is not None:
        z2 = (np.asarray(mean_ref2, np.float64)[first, -1] - mean_a[first, -1])[ok] / se[first, -1][ok]
        out["ks_ref"] = float(stats.ks_2samp(z[0::2], z2[1::2]).pvalue)
    return out


def assert_sampled_gate(g, need_ks_t=True):
    assert g["frac_ok"] >= 0.999, g
    assert abs(g["bias"]) < 1e-3, g
    assert 0.9 < g["std_ratio"] < 1.1, g
    if need_ks_t:
        assert g["ks_t"] > 0.01, g
    if "ks_ref" in g:
        assert g["ks_ref"] > 0.01, g


# ---- stand-in for an open database.h5 (h5py.File / utils.h5reader.File interface used by load_HDF5) -------------
class FakeDataset:
    def __init__(self, arr, attrs=None):
        self.arr, self.attrs = arr, dict(attrs or {})
        self.shape = arr.shape

    def __getitem__(self, idx):          # a read from disk: the caller gets fresh arrays (the reference clips them in place)
        out = self.arr[idx]
        if isinstance(out, np.ndarray) and out.dtype == object:
            out = out.copy()
            for i in np.ndindex(out.shape):
                if isinstance(out[i], np.ndarray):
                    out[i] = out[i].copy()
            return out
        return out.copy() if isinstance(out, np.ndarray) else out

    def __len__(self):
        return len(self.arr)


class FakeH5(dict):
    def keys(self):                      # h5py lists a group's members in name order
        return sorted(dict.keys(self))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def synthetic_database(n=60, seed=3, n_filters=4):
    """The datasets save_to_HDF5 (utils/data_utils.py:486-745) writes, for n synthetic light curves (4 filters,
    no host columns dropped): returns (dict name -> (array, attrs), feature names)."""
    g = np.random.default_rng(seed)
    flt = "griz"[:n_filters]
    combos = sorted("".join(flt[i] for i in range(n_filters) if (m >> i) & 1) for m in range(1, 2 ** n_filters))
    feats = ([f"FLUXCAL_{f}" for f in flt] + [f"FLUXCALERR_{f}" for f in flt] +
             ["delta_time", "HOSTGAL_PHOTOZ", "HOSTGAL_PHOTOZ_ERR", "HOSTGAL_SPECZ", "HOSTGAL_SPECZ_ERR"] + combos)
    nf = len(feats)
    data = np.empty(n, dtype=object)
    for i in range(n):
        L = int(g.integers(1, 25))
        X = np.zeros((L, nf), np.float32)
        X[:, :n_filters] = g.lognormal(3.0, 1.0, (L, n_filters)) - 30.0       # some values below the training minimum
        X[:, n_filters:2 * n_filters] = g.lognormal(1.0, 0.5, (L, n_filters))
        dt = g.uniform(0.4, 6.0, L)
        dt[0] = 0.0
        X[:, 2 * n_filters] = dt
        X[:, 2 * n_filters + 1:2 * n_filters + 5] = g.uniform(0.01, 1.0, 4)
        X[np.arange(L), 2 * n_filters + 5 + g.integers(0, len(combos), L)] = 1.0
        data[i] = X.ravel()
    db = {
        "data": (data, {"n_features": np.int64(nf)}),
        "features": (np.array([f.encode() for f in feats], dtype=object), {}),
        "SNID": (np.array([str(7000 + 3 * i).encode() for i in range(n)], dtype=object), {}),
        "dataset_photometry_2classes": (g.choice([0, 0, 0, 1, 2, 2], n).astype(np.int8), {}),
        "dataset_saltfit_2classes": (g.choice([0, 1, 2, -1], n).astype(np.int8), {}),
        "target_2classes": (g.integers(0, 2, n).astype(np.int8), {}),
        "dataset_photometry_7classes": (g.choice([0, 1, 2], n).astype(np.int8), {}),
        "target_7classes": (g.integers(0, 7, n).astype(np.int8), {}),
        "data_types_training": (np.array([b"Ia", b"nonIa"], dtype="S100"), {}),
        "PEAKMJD": (g.uniform(56000, 57000, n).astype(np.float32), {}),
        "PEAKMJDNORM": (g.uniform(-5, 60, n).astype(np.float32), {}),
        "SNTYPE": (g.choice([101, 120, 121], n).astype(np.int32), {}),
        "mB": (g.normal(20, 1, n).astype(np.float32), {}),
        "c": (g.normal(0, 0.1, n).astype(np.float32), {}),
        "x1": (g.normal(0, 1, n).astype(np.float32), {}),
        "SIM_REDSHIFT_CMB": (g.uniform(0.05, 1.2, n).astype(np.float32), {}),
        "PEAKMJD-2_unique_nights": (g.integers(0, 9, n).astype(np.uint8), {}),
    }
    return db, feats


def loader_settings(feats, norm="global", redshift="none", nb_classes=2):
    """The attributes load_HDF5 / fill_data_list read from ExperimentSettings (experiment_settings.py:105-246)."""
    from types import SimpleNamespace
    flt = [f.split("_")[1] for f in feats if f.startswith("FLUXCAL_")]
    st = SimpleNamespace(processed_dir="/nonexistent", source_data="photometry", nb_classes=nb_classes, data_fraction=0.8,
                         norm=norm, redshift=redshift, sntype_var="SNTYPE", list_filters=flt)
    st.training_features_to_normalize = [f"FLUXCAL_{f}" for f in flt] + [f"FLUXCALERR_{f}" for f in flt] + ["delta_time"]
    st.all_features = list(feats)
    combos = [f for f in feats if set(f) <= set(flt)]
    non_z = [f for f in feats if "FLUX" in f or f in combos or "delta_time" in f]
    z = [f for f in feats if "HOSTGAL_PHOTOZ" in f] if redshift == "zpho" else \
        [f for f in feats if "HOSTGAL_SPECZ" in f] if redshift == "zspe" else []
    st.training_features = non_z + z
    st.idx_features = [i for i, f in enumerate(feats) if f in st.training_features]
    st.idx_flux = [i for i, f in enumerate(st.training_features) if "FLUXCAL_" in f]
    st.idx_fluxerr = [i for i, f in enumerate(st.training_features) if "FLUXCALERR_" in f]
    st.idx_delta_time = [i for i, f in enumerate(st.training_features) if "delta_time" in f]
    st.idx_features_to_normalize = [i for i, f in enumerate(feats) if f in st.training_features_to_normalize]
    st.d_feat_to_idx = {f: i for i, f in enumerate(feats)}
    nF = len(flt)
    st.arr_norm = np.array([[-10.0, 3.2, 0.9]] * nF + [[0.0, 1.0, 0.5]] * nF + [[0.0, 0.9, 0.8]], np.float64)
    return st


def fake_h5(db):
    return FakeH5({k: FakeDataset(a, at) for k, (a, at) in db.items()})
