"""Hot-path subset of the reference's validation/validate_rnn.py.

Same helpers, same contracts:
  get_batch_predictions / _MFE      :43-86    one forward -> softmax -> numpy
  construct_predict_dict            :132-144
  find_idx                          :19-40
  get_aggregated                    :163-172  per-curve median / std (ddof=1) over the S samples
and the MC loop of `get_predictions` (:292-394) / `get_predictions_for_speed_benchmark`
(:513-584) re-shaped for the GPU: instead of S sequential forwards per batch of 128, ALL
curves x ALL S samples go through one persistent launch (`predict_mc`).
HDF5 loading, SNinfo lookup and pickling stay out of scope; callers pass `list_data`
(the list of (X_normed, target, SNID, ...) tuples tu.load_HDF5 returns).
"""
import math
from time import time

import numpy as np
import torch

from .. import engine as E
from ..utils import training_utils as tu

OFFSETS_VAL = [-7, -2, -1, 0, 1, 2, 30]        # utils/data_utils.py:13-19
OFFSETS_STR = ["-7", "-2", "-1", "", "+1", "+2", "+30"]
OOD_TYPES = ["random", "reverse", "shuffle", "sin"]


def find_idx(array, value):
    """Index of the element of sorted `array` closest to `value` (validate_rnn.py:19-40)."""
    idx = np.searchsorted(array, value, side="left")
    if idx > 0 and (idx == len(array) or math.fabs(value - array[idx - 1]) < math.fabs(value - array[idx])):
        return idx - 1
    return idx


def get_batch_predictions(rnn, X, target):
    """softmax(rnn.forward(X)) -> (arr_preds (B,C) float32, arr_target) (validate_rnn.py:43-63)."""
    out = rnn.forward(X)
    arr_preds = torch.nn.functional.softmax(out, dim=-1).data.cpu().numpy()
    return arr_preds, target.detach().cpu().numpy()


def get_batch_predictions_MFE(rnn, X, target):
    """Same with mean_field_inference=True (validate_rnn.py:66-86)."""
    out = rnn.forward(X, mean_field_inference=True)
    arr_preds = torch.nn.functional.softmax(out, dim=-1).data.cpu().numpy()
    return arr_preds, target.detach().cpu().numpy()


def construct_predict_dict(n_elements, n_samples, nb_classes, MFE=False):
    keys = ["all"] + [f"PEAKMJD{o}" for o in OFFSETS_STR if not MFE] + [f"all_{o}" for o in OOD_TYPES]
    d = {k: np.zeros((n_elements, n_samples, nb_classes), np.float32) for k in keys}
    d["target"] = np.zeros((n_elements, n_samples), np.int64)
    d["SNID"] = np.zeros((n_elements, n_samples)).astype(str)
    return d


def ragged_to_device(curves, device):
    """list of (len_i, D) float32 arrays -> (rows (sum len, D) on device, curve_off int64 device, lengths)."""
    lengths = np.fromiter((c.shape[0] for c in curves), dtype=np.int32, count=len(curves))
    rows = torch.from_numpy(np.concatenate(curves, axis=0).astype(np.float32, copy=False)).to(device)
    D = rows.size(1)
    off = np.zeros(len(curves), np.int64)
    np.cumsum(lengths[:-1].astype(np.int64) * D, out=off[1:])
    return rows, torch.from_numpy(off).to(device), lengths


def predict_mc(rnn, curves, n_samples, mean_field_inference=False, lengths=None, device=None):
    """Classify every curve n_samples times in one launch.

    curves: list of (len_i, D) arrays, or a tuple (rows, curve_off, lengths) already on the device.
    Returns probs as a device tensor (N, S, C) in input order -- the layout of d_pred["all"]
    (validate_rnn.py:135, 329).
    """
    device = device or next(rnn.parameters()).device
    if isinstance(curves, tuple):
        rows, curve_off, lengths = curves
    else:
        rows, curve_off, lengths = ragged_to_device(curves, device)
    wk, masks = rnn.mc_weights(n_samples, len(lengths), device, mean_field_inference)
    cfg = rnn.infer_cfg(masks is not None)
    plan = E.Plan(cfg, lengths, device)
    xp = E.pack_ragged(cfg, plan, rows, curve_off)
    _, probs = E.lstm_forward(cfg, plan, xp, wk, n_samples, masks=masks, want_logits=False, want_probs=True,
                              sample_major=False)
    return probs


def aggregate(probs):
    """Per-curve median and std over the S samples: the device-side equivalent of
    get_aggregated_dataframe's groupby(SNID).median()/.std() (validate_rnn.py:163-172;
    pandas std is ddof=1).  probs (N,S,C) -> (median (N,C), std (N,C))."""
    med = probs.median(dim=1).values if probs.size(1) % 2 == 1 else \
        0.5 * (probs.kthvalue(probs.size(1) // 2, dim=1).values + probs.kthvalue(probs.size(1) // 2 + 1, dim=1).values)
    std = probs.std(dim=1, unbiased=True) if probs.size(1) > 1 else torch.full_like(med, float("nan"))
    return med, std


def get_predictions_mc(settings, rnn, list_data, peak_mjds=None, times=None):
    """The MC part of get_predictions' _populate_predict_dict (validate_rnn.py:292-394) on in-memory
    data: fills "all", the PEAKMJD offsets (when peak_mjds/times are given) and the OOD variants,
    for S = settings.num_inference_samples draws, plus the MFE dict.  Returns (d_pred, d_pred_MFE)."""
    settings.random_length = False
    settings.random_redshift = False
    S = 1 if settings.model == "vanilla" else settings.num_inference_samples
    N, C = len(list_data), settings.nb_classes
    rnn.eval()
    targets = np.array([d[1] for d in list_data], np.int64)
    snids = np.array([str(d[2]) if len(d) > 2 else str(i) for i, d in enumerate(list_data)])

    def run(curves, n_samples, mfe):
        with torch.no_grad():
            return predict_mc(rnn, curves, n_samples, mean_field_inference=mfe).cpu().numpy()

    out = []
    for mfe, n_s in ((False, S), (True, 1)):
        d = construct_predict_dict(N, n_s, C, MFE=mfe)
        d["target"][:] = targets[:, None]
        d["SNID"][:] = snids[:, None]
        base = [x[0] for x in list_data]
        d["all"][:] = run(base, n_s, mfe)
        if not mfe and peak_mjds is not None and times is not None:
            for off, sfx in zip(OFFSETS_VAL, OFFSETS_STR):
                cut = np.array([find_idx(times[k], peak_mjds[k] + off) for k in range(N)])
                inb = np.where(cut >= 1)[0]
                d[f"PEAKMJD{sfx}"][:] = np.nan        # empty slices: no prediction (validate_rnn.py:372-374)
                if len(inb):
                    d[f"PEAKMJD{sfx}"][inb] = run([base[k][: cut[k]] for k in inb], n_s, mfe)
        for ood in ["random", "shuffle", "reverse", "sin"]:   # reference order (validate_rnn.py:380)
            if ood in ("random", "sin") and not hasattr(settings, "idx_flux"):
                continue
            d[f"all_{ood}"][:] = run([tu._ood(x, ood, settings) for x in base], n_s, mfe)
        out.append(d)
    return tuple(out)


def get_predictions_for_speed_benchmark(settings, rnn, list_data):
    """validate_rnn.py:513-584 without the CSV: supernova_per_s = N / wall time to produce
    S softmax vectors for every curve, host batching and the device->host copy included."""
    settings.random_length = False
    S = 1 if settings.model == "vanilla" else settings.num_inference_samples
    rnn.eval()
    start = time()
    with torch.no_grad():
        probs = predict_mc(rnn, [d[0] for d in list_data], S).cpu().numpy()
    total = time() - start
    return len(list_data) / total, probs


def flatten_to_dataframe(pred_dict):
    """(N, S, C) arrays -> one row per (curve, sample): columns `<key>_class<i>`, `target`, `SNID`
    (validate_rnn.py:147-160)."""
    import pandas as pd
    n, s, _ = pred_dict["all"].shape
    cols = {}
    for key, value in pred_dict.items():
        v = value.reshape(n * s, -1)
        if v.shape[1] == 1:
            cols[key] = v.ravel()
        else:
            for i in range(v.shape[1]):
                cols[f"{key}_class{i}"] = v[:, i]
    return pd.DataFrame.from_dict(cols)


def get_aggregated_dataframe(pred_df):
    """Per-SNID median and std (ddof = 1) of every column over the MC samples (validate_rnn.py:163-172)."""
    import pandas as pd
    g = pred_df.groupby("SNID")
    med, std = g.median(), g.std()
    med.columns = [f"{c}_median" for c in med.columns]
    std.columns = [f"{c}_std" for c in std.columns]
    agg = pd.merge(med, std, on="SNID")
    agg["SNID"] = agg.index
    agg["target"] = agg["target_median"]
    return agg


def get_predictions(settings, model_file=None, list_data=None, peak_mjds=None, times=None, rnn=None):
    """The RNN branch of validate_rnn.get_predictions (:229-442): load the state dict, run the MC and the
    mean-field passes, write `PRED_<model>.pickle` (one row per curve and sample) and, for the variational and
    Bayesian models, `PRED_<model>_aggregated.pickle`; returns the list of prediction files.
    `list_data` is the test list tu.load_HDF5(settings, test=True) would return (HDF5 is out of scope here);
    PEAKMJD offsets are filled when `peak_mjds` / `times` (per-curve arrays of observation times) are given.
    The state dict is read with weights_only=True."""
    import os
    if list_data is None:
        raise ValueError("supernnova_amd.validation.validate_rnn.get_predictions takes the test list of "
                         "tu.load_HDF5 as `list_data`: the HDF5 reader is not part of this package")
    if rnn is None:
        rnn = tu.get_model(settings, list_data[0][0].shape[1])
        if model_file is None:
            model_file = os.path.join(str(settings.rnn_dir), f"{settings.pytorch_model_name}.pt")
        rnn.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=True))
        rnn.to(getattr(settings, "device", "cuda"))
    rnn.eval()
    dump_dir = getattr(settings, "dump_dir", None) or os.path.dirname(os.path.abspath(model_file or str(settings.rnn_dir) + "/x"))
    d_pred, d_pred_mfe = get_predictions_mc(settings, rnn, list_data, peak_mjds, times)
    df = flatten_to_dataframe(d_pred)
    pred_file = os.path.join(dump_dir, f"PRED_{settings.pytorch_model_name}.pickle")
    df.to_pickle(pred_file, protocol=4)
    if settings.model in ("variational", "bayesian"):
        get_aggregated_dataframe(df).to_pickle(pred_file.replace(".pickle", "_aggregated.pickle"), protocol=4)
    return [pred_file]
