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
    rl = getattr(curves, "rows", None)
    if rl is not None and len(curves) == len(curves.lengths):       # tu.RaggedList: already concatenated
        return (torch.from_numpy(rl).to(device), torch.from_numpy(np.asarray(curves.curve_off, np.int64)).to(device),
                curves.lengths)
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


def predict_mc_prefixes(rnn, curves, cuts, n_samples, mean_field_inference=False, device=None):
    """Early classification in ONE launch: every (light curve k, prefix length cuts[v][k]) pair is a curve of its own
    for the kernel -- a view of the first cuts[v][k] rows of curve k (no row is copied: the views share `rows`), so
    the full curves and the 7 PEAKMJD-offset slices of validate_rnn.py:337-374 are one plan, one pack, one weight
    draw and one persistent launch instead of 8.  (Only layer 0's forward direction could be shared between the
    prefixes of a curve -- every other (layer, direction) sees the cut through the reverse pass -- so the prefixes run
    as independent work items.)  The S weight draws / W_hh row masks are shared by all prefixes of the launch; the
    reference draws anew for every forward, which leaves each key's statistics unchanged.

    curves: list of (len_k, D) arrays; cuts: list of int arrays (N,), entries < 1 = "slice empty, no prediction".
    Returns a list of (idx, probs (len(idx), S, C) device tensor) per entry of `cuts`, idx = curves with a slice."""
    device = device or next(rnn.parameters()).device
    rows, curve_off, lengths = ragged_to_device(curves, device)
    D = rows.size(1)
    off_host = np.zeros(len(curves), np.int64)
    np.cumsum(lengths[:-1].astype(np.int64) * D, out=off_host[1:])
    idxs, v_off, v_len = [], [], []
    for c in cuts:
        c = np.asarray(c)
        if (c > lengths).any():
            raise ValueError("a prefix length exceeds its light curve")
        inb = np.where(c >= 1)[0]
        idxs.append(inb)
        v_off.append(off_host[inb])
        v_len.append(c[inb].astype(np.int32))
    all_len = np.concatenate(v_len)
    if all_len.size == 0:
        return [(i, torch.empty((0, n_samples, rnn.cfg.nb_classes), device=device)) for i in idxs]
    wk, masks = rnn.mc_weights(n_samples, all_len.size, device, mean_field_inference)
    cfg = rnn.infer_cfg(masks is not None)
    plan = E.Plan(cfg, all_len, device)
    xp = E.pack_ragged(cfg, plan, rows, torch.from_numpy(np.concatenate(v_off)).to(device), shared_rows=True)
    _, probs = E.lstm_forward(cfg, plan, xp, wk, n_samples, masks=masks, want_logits=False, want_probs=True,
                              sample_major=False)
    out, a = [], 0
    for inb in idxs:
        out.append((inb, probs[a:a + len(inb)]))
        a += len(inb)
    return out


def aggregate(probs):
    """Per-curve median and std over the S samples: the device-side equivalent of
    get_aggregated_dataframe's groupby(SNID).median()/.std() (validate_rnn.py:163-172;
    pandas std is ddof=1).  probs (N,S,C) -> (median (N,C), std (N,C))."""
    med = probs.median(dim=1).values if probs.size(1) % 2 == 1 else \
        0.5 * (probs.kthvalue(probs.size(1) // 2, dim=1).values + probs.kthvalue(probs.size(1) // 2 + 1, dim=1).values)
    std = probs.std(dim=1, unbiased=True) if probs.size(1) > 1 else torch.full_like(med, float("nan"))
    return med, std


def get_predictions_mc(settings, rnn, list_data, peak_mjds=None, times=None, batch_size=None):
    """The MC part of get_predictions' _populate_predict_dict (validate_rnn.py:292-394) on in-memory
    data: fills "all", the PEAKMJD offsets (when peak_mjds/times are given) and the OOD variants,
    for S = settings.num_inference_samples draws, plus the MFE dict.  Returns (d_pred, d_pred_MFE).

    Batching.  The reference walks mini-batches of settings.batch_size, and every forward of a mini-batch draws new
    weights / masks.  In sampled mode (rnn.draws == "philox") all curves go through ONE launch per variant (the
    statistics are the same, the launch count is not).  In parity mode (rnn.draws == "host") -- or when `batch_size`
    is given -- the reference's mini-batches are kept, each variant of each mini-batch is one launch of S draws made
    with the reference's torch calls in the reference's order, and numpy's global RNG is consumed by the OOD
    transforms in the reference's order: seeded like a reference run, the dict replays that run."""
    settings.random_length = False
    settings.random_redshift = False
    S = 1 if settings.model == "vanilla" else settings.num_inference_samples
    N, C = len(list_data), settings.nb_classes
    rnn.eval()
    targets = np.array([d[1] for d in list_data], np.int64)
    snids = np.array([str(d[2]) if len(d) > 2 else str(i) for i, d in enumerate(list_data)])
    if batch_size is None and getattr(rnn, "draws", "philox") == "host":
        batch_size = settings.batch_size
    if batch_size is None:
        batches = [np.arange(N)]
    else:
        batches = np.array_split(np.arange(N), int(N / min(N, batch_size)))   # validate_rnn.py:285-287

    def run(curves, n_samples, mfe):
        with torch.no_grad():
            if batch_size is None:
                return predict_mc(rnn, curves, n_samples, mean_field_inference=mfe).cpu().numpy()
            # the reference draws its per-curve dropout masks for the length-SORTED mini-batch (get_data_batch,
            # training_utils.py:449-453: np.argsort(list_len)[::-1]); host-drawn masks are indexed by position in the
            # launch, so the launch takes the curves in that order and the rows are put back afterwards
            idx_sort = np.argsort([c.shape[0] for c in curves])[::-1]
            probs = predict_mc(rnn, [curves[k] for k in idx_sort], n_samples, mean_field_inference=mfe).cpu().numpy()
            return probs[np.argsort(idx_sort)]

    base = [x[0] for x in list_data]
    out = []
    for mfe, n_s in ((False, S), (True, 1)):
        d = construct_predict_dict(N, n_s, C, MFE=mfe)
        d["target"][:] = targets[:, None]
        d["SNID"][:] = snids[:, None]
        if batch_size is None and not mfe and peak_mjds is not None and times is not None:
            # sampled mode: full curves + the 7 early-classification slices in one launch, the 4 OOD variants in another
            cuts = [np.array([len(c) for c in base])]
            for off in OFFSETS_VAL:
                cuts.append(np.array([find_idx(times[k], peak_mjds[k] + off) for k in range(N)]))
            with torch.no_grad():
                res = predict_mc_prefixes(rnn, base, cuts, n_s, mean_field_inference=mfe)
            for key, (inb, pr) in zip(["all"] + [f"PEAKMJD{sfx}" for sfx in OFFSETS_STR], res):
                if len(inb) or key == "all":
                    if key != "all":
                        d[key][:] = np.nan
                    d[key][inb] = pr.cpu().numpy()
            oods = [o for o in ["random", "shuffle", "reverse", "sin"] if o in ("shuffle", "reverse") or hasattr(settings, "idx_flux")]
            if oods:
                both = [tu._ood(x, ood, settings) for ood in oods for x in base]
                pr = run(both, n_s, mfe)
                for j, ood in enumerate(oods):
                    d[f"all_{ood}"][:] = pr[j * N:(j + 1) * N]
            out.append(d)
            continue
        for idxs in batches:
            d["all"][idxs] = run([base[k] for k in idxs], n_s, mfe)
            if not mfe and peak_mjds is not None and times is not None:
                for off, sfx in zip(OFFSETS_VAL, OFFSETS_STR):
                    cut = np.array([find_idx(times[k], peak_mjds[k] + off) for k in idxs])
                    inb = np.where(cut >= 1)[0]
                    if len(inb):     # (a mini-batch whose slices are all empty is left untouched, as the reference's)
                        d[f"PEAKMJD{sfx}"][idxs[cut < 1]] = np.nan    # empty slices: no prediction (validate_rnn.py:372-374)
                        d[f"PEAKMJD{sfx}"][idxs[inb]] = run([base[idxs[j]][: cut[j]] for j in inb], n_s, mfe)
            for ood in ["random", "shuffle", "reverse", "sin"]:   # reference order (validate_rnn.py:380)
                if ood in ("random", "sin") and not hasattr(settings, "idx_flux"):
                    continue
                d[f"all_{ood}"][idxs] = run([tu._ood(base[k], ood, settings) for k in idxs], n_s, mfe)
        out.append(d)
    return tuple(out)


def get_predictions_for_speed_benchmark(settings):
    """The reference's own definition of the speed metric (validate_rnn.py:513-584), same signature and side effects:
    load the test list of `settings.processed_dir/database.h5` (tu.load_HDF5(test=True)), build a RANDOMLY INITIALISED
    model (:535-537), classify every test curve `num_inference_samples` times (1 for a vanilla model) mini-batch by
    mini-batch of `settings.batch_size` (:540-542), and record
        Supernova_per_s = num_elem / wall time        (host batching and the device->host copy inside the clock, :544-566)
    in `settings.stats_dir/rnn_speed.csv` with the columns Supernova_per_s (stored through float16, :569-571), model,
    batch_size, device, id = "<model>_<batch_size>_<device>"; an existing file has the row with that id replaced or
    appended, anything unreadable is overwritten (:578-584).
    What differs is inside a mini-batch: the S forwards of the reference's inner loop (:559-563) are ONE launch
    (predict_mc) whose (B, S, C) probabilities are copied to the host once -- batch_size stays the caller's knob
    ("Batch size can be controled to speed up predictions"): one batch holding the whole test set is one launch.
    Returns Supernova_per_s (the reference returns None)."""
    import os
    import pandas as pd
    settings.random_length = False
    if "vanilla" in settings.pytorch_model_name:
        settings.num_inference_samples = 1
    list_data_test = tu.load_HDF5(settings, test=True)
    rnn = tu.get_model(settings, len(settings.training_features))
    rnn.to(settings.device)
    rnn.eval()
    num_elem = len(list_data_test)
    num_batches = num_elem / min(num_elem, settings.batch_size)
    list_batches = np.array_split(np.arange(num_elem), int(num_batches))      # (array_split truncates the reference's float the same way)
    curves = [d[0] for d in list_data_test]
    start_time = time()
    for batch_idxs in list_batches:
        with torch.no_grad():
            arr_preds = predict_mc(rnn, [curves[k] for k in batch_idxs], settings.num_inference_samples,
                                   device=torch.device(settings.device)).cpu().numpy()
    total_time = time() - start_time
    assert arr_preds.shape[1:] == (settings.num_inference_samples, settings.nb_classes)
    supernova_per_s = num_elem / total_time
    model_id = f"{settings.model}_{settings.batch_size}_{settings.device}"
    df = pd.DataFrame(data=np.array([supernova_per_s]).astype(np.float16), columns=["Supernova_per_s"])
    df["model"] = settings.model
    df["batch_size"] = settings.batch_size
    df["device"] = settings.device
    df["id"] = model_id
    os.makedirs(str(settings.stats_dir), exist_ok=True)
    results_file = os.path.join(str(settings.stats_dir), "rnn_speed.csv")
    try:
        df_all = pd.read_csv(results_file, index_col="id")
        df_all.loc[model_id] = df.set_index("id").loc[model_id]
        df_all.reset_index().to_csv(results_file, index=False)
    except Exception:
        df.to_csv(results_file, index=False)
    return supernova_per_s


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
    """The RNN branch of validate_rnn.get_predictions (:229-442), reference signature `(settings, model_file=None)`:
    load the test list from `settings.processed_dir/database.h5` (tu.load_HDF5) with the peak dates of
    du.load_HDF5_SNinfo, load the state dict (weights_only), run the MC and the mean-field passes, write
    `PRED_<model>.pickle` (one row per curve and sample) and, for the variational and Bayesian models,
    `PRED_<model>_aggregated.pickle`; returns the list of prediction files.  Output directory as the reference:
    `models_dir/<name>` without a model file, `dump_dir/models/<name>` with one (:253-273).
    In-memory callers may pass `list_data` (the list tu.load_HDF5(settings, test=True) returns) and `peak_mjds` /
    `times` themselves, and a loaded `rnn`."""
    import os
    name = settings.pytorch_model_name
    if list_data is None:
        from ..utils import data_utils as du
        list_data = tu.load_HDF5(settings, test=True)
        info = du.load_HDF5_SNinfo(settings).set_index("SNID")
        peak_mjds = info.loc[[d[2] for d in list_data]]["PEAKMJDNORM"].values
        col = settings.d_feat_to_idx["delta_time"]
        times = [np.cumsum(d[3][:, col]) for d in list_data]            # validate_rnn.py:303-308
    if model_file is None and rnn is None:
        dump_dir = getattr(settings, "rnn_dir", None) or f"{settings.models_dir}/{name}"
        model_file = os.path.join(str(dump_dir), f"{name}.pt")
    elif getattr(settings, "dump_dir", None) and model_file is not None and getattr(settings, "models_dir", None):
        dump_dir = f"{settings.dump_dir}/models/{name}"
    else:
        dump_dir = getattr(settings, "dump_dir", None) or os.path.dirname(os.path.abspath(model_file or str(settings.rnn_dir) + "/x"))
    os.makedirs(str(dump_dir), exist_ok=True)
    if rnn is None:
        rnn = tu.get_model(settings, list_data[0][0].shape[1])
        rnn.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=True))
        rnn.to(getattr(settings, "device", "cuda"))
    rnn.eval()
    d_pred, d_pred_mfe = get_predictions_mc(settings, rnn, list_data, peak_mjds, times)
    df = flatten_to_dataframe(d_pred)
    pred_file = os.path.join(str(dump_dir), f"PRED_{name}.pickle")
    df.to_pickle(pred_file, protocol=4)
    if settings.model in ("variational", "bayesian"):
        get_aggregated_dataframe(df).to_pickle(pred_file.replace(".pickle", "_aggregated.pickle"), protocol=4)
    return [pred_file]
