"""On-the-fly (broker) classification: DataFrame of photometry -> (ids, probabilities (B, S, C)).

Mirror of the reference's validation/validate_onthefly.py (`get_settings` :16-62, `format_data`
:65-120, `classify_lcs` :123-236) and of the pieces it pulls in:
  utils/data_utils.compute_delta_time            :364-390
  data/make_dataset.pivot_dataframe_single_from_df :800-938   (8-hour grouping, lowest-error pick, pivot)
  utils/training_utils.normalize_arr             :23-86       (clip + log-standardise, "global" norm)
  conf.get_norm_from_model                       :722-738
The featurisation is vectorised numpy over all light curves at once (no per-row Python loop, no pandas
pivot); the classification is ONE launch for all curves x all MC samples (predict_mc) instead of
num_inference_samples sequential forwards.
"""
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pandas as pd
import torch

from ..utils import training_utils as tu
from .validate_rnn import predict_mc

MIN_DT = 0.33   # observations within ~8 hours share one row (make_dataset.py:823)


def get_settings(model_file):
    """Settings of a trained model from the cli_args.json / data_norm.json next to it."""
    model_dir = Path(model_file).parent
    with open(model_dir / "cli_args.json") as f:
        cli = json.load(f)
    cli.setdefault("sntype_var", "SNTYPE")
    cli.setdefault("additional_train_var", None)
    cli["no_dump"] = True
    cli["models_dir"] = model_dir.as_posix()
    st = SimpleNamespace(**cli)
    with open(model_dir / "data_norm.json") as f:
        norm = json.load(f)
    st.arr_norm = np.array([[norm[k]["min"], norm[k]["mean"], norm[k]["std"]]
                            for k in st.training_features_to_normalize])
    return st


def _host_columns(df, st):
    """Missing host-galaxy columns are zero-filled when the model does not use them (validate_onthefly.py:76-95)."""
    host_model = [k for k in st.all_features if "HOST" in k]
    missing = [k for k in host_model if k not in df.keys()]
    ok = ((len(missing) > 0 and st.redshift == "none")
          or (st.redshift == "zspe" and "HOSTGAL_SPECZ" not in missing)
          or (st.redshift == "zpho" and "HOSTGAL_PHOTOZ" not in missing))
    if ok:
        for k in missing:
            df[k] = np.zeros(len(df))
    elif len(missing) > 0:
        raise ValueError(f"Missing features needed for classification {missing}")
    return df


def featurize(df, st):
    """DataFrame(SNID, MJD, FLUXCAL, FLUXCALERR, FLT[, HOSTGAL_*]) -> (ids, offsets, X_all).

    X_all is (n_rows, len(all_features)) float32 with one row per 8-hour group, light curves in sorted SNID
    order and contiguous; offsets[i]:offsets[i+1] are the rows of ids[i]."""
    df = _host_columns(df.copy(), st)
    filters = list(st.list_filters)
    inv, ids = pd.factorize(df["SNID"].values, sort=True)     # ids sorted, inv = index into ids
    ids = np.asarray(ids)
    inv = np.ascontiguousarray(inv, dtype=np.int64)
    mjd = np.ascontiguousarray(df["MJD"].values, dtype=np.float64)
    flux = np.ascontiguousarray(df["FLUXCAL"].values, dtype=np.float64)
    err = np.ascontiguousarray(df["FLUXCALERR"].values, dtype=np.float64)
    fidx = np.ascontiguousarray(pd.Categorical(df["FLT"].values, categories=filters).codes, dtype=np.int64)  # -1: unknown
    n, F = len(mjd), len(filters)
    # Native passes (csrc/snn_featurize.hip): time order inside each light curve, the reference's sequential 8-hour
    # grouping, the output row of every observation; then per (row, filter) the observation with the lowest
    # FLUXCALERR (earliest on ties).  Filters the model does not know fill no column.
    from .. import _lib
    lib = _lib.load()
    order, row_of, row_mjd_of = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n)
    rows_per_curve = np.zeros(len(ids), np.int64)
    n_rows = _lib.check(lib.snn_featurize_plan(n, len(ids), inv.ctypes.data, mjd.ctypes.data, MIN_DT, order.ctypes.data,
                                               row_of.ctypes.data, row_mjd_of.ctypes.data, rows_per_curve.ctypes.data),
                        "snn_featurize_plan")
    fl, er = np.zeros((n_rows, F)), np.zeros((n_rows, F))
    present = np.zeros((n_rows, F), np.uint8)
    row_mjd = np.zeros(n_rows)
    _lib.check(lib.snn_featurize_fill(n, n_rows, F, order.ctypes.data, row_of.ctypes.data, row_mjd_of.ctypes.data,
                                      flux.ctypes.data, err.ctypes.data, fidx.ctypes.data, fl.ctypes.data, er.ctypes.data,
                                      present.ctypes.data, row_mjd.ctypes.data), "snn_featurize_fill")
    present = present.astype(bool)
    row_curve = np.repeat(np.arange(len(ids)), rows_per_curve)
    # a group that holds only filters the model does not know still survives the reference's pivot (its FLUXCAL_<f>
    # column exists there): zero fluxes, empty filter string, no one-hot bit (make_dataset.py:885-910)
    start = np.ones(n_rows, bool)
    start[1:] = row_curve[1:] != row_curve[:-1]
    gdt = np.diff(row_mjd, prepend=row_mjd[:1])
    gdt[start] = 0.0
    # filter-combination name of a row = its present filters concatenated in list_filters order; as a bit mask here
    bits = present @ (1 << np.arange(F))
    combo_bits = {c: m for m in range(1 << F)
                  for c in ["".join(f for k, f in enumerate(filters) if m >> k & 1)] if c in set(st.list_filters_combination)}
    cols = {f"FLUXCAL_{f}": fl[:, k] for k, f in enumerate(filters)}
    cols.update({f"FLUXCALERR_{f}": er[:, k] for k, f in enumerate(filters)})
    cols["delta_time"] = gdt
    host = [k for k in st.all_features if "HOST" in k]
    if host:
        first_of = np.zeros(len(ids), np.int64)         # host values are per light curve: take its first row's (in time)
        first_of[inv[order][::-1]] = order[::-1]
        for k in host:
            cols[k] = df[k].values.astype(np.float64)[first_of][row_curve]
    X = np.zeros((n_rows, len(st.all_features)), np.float32)
    for j, feat in enumerate(st.all_features):
        if feat in cols:
            X[:, j] = cols[feat].astype(np.float32)     # the reference casts float64 columns to float32 (:916-918)
        elif feat in combo_bits:
            X[:, j] = bits == combo_bits[feat]
    live_ids = ids[np.unique(row_curve)] if n_rows else ids[:0]
    counts = np.bincount(row_curve, minlength=len(ids))[np.unique(row_curve)] if n_rows else np.zeros(0, np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return live_ids, offsets, X


def format_data(df, settings):
    """DataFrame in the reference's format: index SNID, columns settings.all_features (validate_onthefly.py:65-120)."""
    ids, off, X = featurize(df, settings)
    idx = np.repeat(ids, np.diff(off))
    return pd.DataFrame(X, index=pd.Index(idx, name="SNID"), columns=list(settings.all_features))


def normalize(X_all, st):
    """Clip to the training minimum, log-standardise ("global" norm), keep the training features
    (validate_onthefly.py:160-181 + training_utils.py:23-86).  float64 arithmetic, float32 storage after the
    clip and after the standardisation, as the reference's float32 `sel.values` array implies.  The columns
    are addressed as the reference does: positions of the normalised features *within training_features*,
    applied to the all_features matrix (:138-142, :166-170) -- identical whenever the normalised features
    lead both lists, which every SuperNNova feature list does."""
    if st.norm not in ("global", "none"):
        raise ValueError(f"on-the-fly normalisation implements norm 'global'/'none', got {st.norm!r}")
    X = X_all.astype(np.float32).copy()
    if st.norm != "none":
        cols = [i for i, f in enumerate(st.training_features) if f in st.training_features_to_normalize]
        lo, mean, std = st.arr_norm[:, 0], st.arr_norm[:, 1], st.arr_norm[:, 2]
        v = np.clip(X[:, cols], lo, np.inf).astype(np.float32).astype(np.float64)
        X[:, cols] = ((np.log(v - lo + 1e-5) - mean) / std).astype(np.float32)
    keep = [i for i, f in enumerate(st.all_features) if f in st.training_features]
    return np.ascontiguousarray(X[:, keep])


def classify_lcs(df, model_file, device, rnn=None):
    """ids (sorted SNIDs), preds (B, num_inference_samples, nb_classes) -- validate_onthefly.py:123-236.
    `rnn` may carry an already-loaded model (brokers classify many alerts with one model)."""
    st = get_settings(model_file)
    st.use_cuda = "cuda" in str(device)
    st.random_length = False
    st.random_redshift = False
    np.random.seed(st.seed)
    torch.manual_seed(st.seed)
    if "vanilla" in st.pytorch_model_name:
        st.num_inference_samples = 1
    ids, off, X_all = featurize(df, st)
    X = normalize(X_all, st)
    if rnn is None:
        rnn = tu.get_model(st, len(st.training_features))
        state = torch.load(model_file, map_location="cpu", weights_only=True)
        try:
            rnn.load_state_dict(state)
        except Exception:
            raise ValueError("Model has less features than data (check model filters)"
                             if len(st.training_features) < len(st.all_features) else "Check correct model is chosen")
        rnn.to(device)
    rnn.eval()
    lengths = np.diff(off).astype(np.int32)
    rows = torch.from_numpy(X).to(device)
    curve_off = torch.from_numpy(off[:-1] * X.shape[1]).to(device)
    with torch.no_grad():
        preds = predict_mc(rnn, (rows, curve_off, lengths), st.num_inference_samples, device=device)
    return ids, preds.cpu().numpy()
