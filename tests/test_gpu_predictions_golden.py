"""a10 / a12 / f2 against the reference's own host code (fixture g10_predictions, tools/make_golden.py:
get_evaluation_metrics, the nested _populate_predict_dict of get_predictions, flatten_to_dataframe and
get_aggregated_dataframe exec'd from the reference's files, driving its three model classes).  The modules here draw
eps / masks with the reference's torch calls in the reference's order (draws = "host"), so seeding torch and numpy like
the fixture's run replays it: metrics, every key of the prediction dicts (full curves, the 7 PEAKMJD offsets incl.
their NaN pattern, the 4 out-of-distribution variants, the mean-field pass) and the aggregated DataFrame."""
import numpy as np
import pytest
import torch

from test_host_logic import settings
from util import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sub(z, prefix):
    return {k[len(prefix):]: v for k, v in z.items() if k.startswith(prefix)}


@pytest.mark.parametrize("model", ["vanilla", "variational", "bayesian"])
def test_get_evaluation_metrics_replays_the_reference(model):
    from supernnova_amd.utils import training_utils as tu
    z = dict(np.load(__import__("os").path.join(__import__("util").GOLDEN, "g10_predictions.npz")))
    _, st, data, rnn = _setup_flat(z, model)
    for tag, np_seed, t_seed, kw in (("full", 17, 1700, {}), ("sub", 18, 1800, {"sample_size": 20})):
        np.random.seed(np_seed)
        torch.manual_seed(t_seed)
        m = tu.get_evaluation_metrics(st, data, rnn, **kw)
        auc, acc, loss, kl = z[f"{model}/metrics_{tag}"]
        assert m["Acc"] == acc
        assert abs(m["AUC"] - auc) < 1e-9
        assert abs(m["loss"] - loss) < 1e-6 * loss
        if model == "bayesian":
            assert abs(m["KL"] - kl) < 1e-5 * kl
        else:
            assert "KL" not in m


def _setup_flat(z, model):
    from supernnova_amd.utils import training_utils as tu
    feats = [str(f) for f in z["features"]]
    st = settings(model=model, use_cuda=True, nb_classes=2, batch_size=16, num_inference_samples=4, norm="global",
                  pytorch_model_name=f"{model}_S_0_CLF_2", redshift="none")
    st.training_features = st.all_features = feats
    st.idx_flux = [i for i, f in enumerate(feats) if "FLUXCAL_" in f]
    st.idx_fluxerr = [i for i, f in enumerate(feats) if "FLUXCALERR_" in f]
    st.idx_delta_time = [i for i, f in enumerate(feats) if "delta_time" in f]
    st.idx_features_to_normalize = st.idx_flux + st.idx_fluxerr + st.idx_delta_time
    st.arr_norm = z["arr_norm"]
    off = np.concatenate([[0], np.cumsum(z["lengths"])])
    data = [(z["x"][off[i]:off[i + 1]], int(z["target"][i]), str(z["snid"][i]), z["x_all"][off[i]:off[i + 1]])
            for i in range(int(z["n"]))]
    rnn = tu.get_model(st, len(feats))
    sd = _sub(z, f"{model}/sd/")
    rnn.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    rnn.to(DEV).eval()
    rnn.draws = "host"
    return z, st, data, rnn


@pytest.mark.parametrize("model", ["vanilla", "variational", "bayesian"])
def test_prediction_dicts_replay_the_reference(model):
    import os
    import util
    from supernnova_amd.validation import validate_rnn as vr
    z = dict(np.load(os.path.join(util.GOLDEN, "g10_predictions.npz")))
    _, st, data, rnn = _setup_flat(z, model)
    dt_col = st.idx_delta_time[0]
    times = [np.cumsum(d[3][:, dt_col]) for d in data]
    np.random.seed(19)
    torch.manual_seed(1900)
    d_pred, d_mfe = vr.get_predictions_mc(st, rnn, data, peak_mjds=z["peak"], times=times)
    ref, ref_mfe = _sub(z, f"{model}/pred/"), _sub(z, f"{model}/pred_mfe/")
    assert set(d_pred) == set(ref) and set(d_mfe) == set(ref_mfe)
    for got, want in ((d_pred, ref), (d_mfe, ref_mfe)):
        for k, v in want.items():
            if k == "SNID":
                assert np.array_equal(got[k].astype(str), v.astype(str))
            elif k == "target":
                assert np.array_equal(got[k], v)
            else:
                assert got[k].shape == v.shape, k
                assert np.array_equal(np.isnan(got[k]), np.isnan(v)), f"{k}: NaN pattern (empty early slices) differs"
                ok = ~np.isnan(v)
                rel = np.abs(got[k][ok] - v[ok]) / np.maximum(np.abs(v[ok]), 1e-30)
                assert rel.max() < util.PROB_RTOL, (k, rel.max())
    # the early-classification keys are exercised: some slices are empty, some are not, and they differ from "all"
    assert np.isnan(ref["PEAKMJD-7"]).any() and (~np.isnan(ref["PEAKMJD-7"])).any()
    # a12: the DataFrames
    df = vr.flatten_to_dataframe(d_pred)
    assert list(df.columns) == [str(c) for c in z[f"{model}/df_columns"]]
    np.testing.assert_allclose(df["all_class1"].values, z[f"{model}/df_all_class1"], rtol=util.PROB_RTOL)
    if model != "vanilla":
        agg = vr.get_aggregated_dataframe(df)
        assert list(agg.columns) == [str(c) for c in z[f"{model}/agg_columns"]]
        assert [str(i) for i in agg.index.values] == [str(i) for i in z[f"{model}/agg_index"]]
        num = [str(c) for c in z[f"{model}/agg_numeric_columns"]]
        want = z[f"{model}/agg_values"]
        got = agg[num].values.astype(np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        # medians to the probability bar; stds of S = 4 values differ by the probabilities' absolute error
        assert np.abs(got[ok] - want[ok]).max() < 2e-6
        # device-side aggregate (median / std over S) against the reference's pandas groupby
        med, std = vr.aggregate(torch.from_numpy(d_pred["all"]).to(DEV))
        cols = {c: i for i, c in enumerate(num)}
        order = np.argsort(np.argsort([str(s) for s in z["snid"]]))      # groupby sorts the SNIDs
        for q in range(2):
            np.testing.assert_allclose(med[:, q].cpu().numpy(), want[order, cols[f"all_class{q}_median"]], atol=2e-6)
            np.testing.assert_allclose(std[:, q].cpu().numpy(), want[order, cols[f"all_class{q}_std"]], atol=2e-6)


@pytest.mark.parametrize("model", ["vanilla", "bayesian"])
def test_multi_prefix_launch_equals_one_launch_per_slice(model):
    """f2: predict_mc_prefixes classifies the full curves and all early slices as overlapping views of one row buffer
    in ONE launch; per slice the probabilities are bit-identical to a launch of its own on the truncated copies with
    the same weight draws (same seed), and get_predictions_mc (sampled mode) fills the same keys with it."""
    import os
    import time
    import util
    from supernnova_amd.validation import validate_rnn as vr
    z = dict(np.load(os.path.join(util.GOLDEN, "g10_predictions.npz")))
    _, st, data, rnn = _setup_flat(z, model)
    rnn.draws = "philox"
    S = 1 if model == "vanilla" else 4
    base = [d[0] for d in data]
    times = [np.cumsum(d[3][:, st.idx_delta_time[0]]) for d in data]
    cuts = [np.array([len(c) for c in base])] + [np.array([vr.find_idx(times[k], z["peak"][k] + off) for k in range(len(base))])
                                                 for off in vr.OFFSETS_VAL]
    with torch.no_grad():
        torch.manual_seed(5)
        multi = vr.predict_mc_prefixes(rnn, base, cuts, S)
        for (inb, pr), cut in zip(multi, cuts):
            assert np.array_equal(inb, np.where(cut >= 1)[0])
            if len(inb) == 0:
                continue
            torch.manual_seed(5)
            single = vr.predict_mc(rnn, [base[k][: cut[k]] for k in inb], S)
            assert torch.equal(pr, single)
        # the dict through the one-launch path: same keys, NaN exactly where the slice is empty
        np.random.seed(1)
        torch.manual_seed(5)
        t0 = time.perf_counter()
        d_pred, d_mfe = vr.get_predictions_mc(st, rnn, data, peak_mjds=z["peak"], times=times)
        torch.cuda.synchronize()
        t_full = time.perf_counter() - t0
        t0 = time.perf_counter()
        vr.predict_mc(rnn, base, S).cpu()
        t_all = time.perf_counter() - t0
    print(f"{model}: full dict (all + 7 early slices + 4 OOD + mean-field pass) {t_full * 1e3:.1f} ms, 'all' only {t_all * 1e3:.1f} ms")
    for (inb, pr), sfx in zip(multi[1:], vr.OFFSETS_STR):
        got = d_pred[f"PEAKMJD{sfx}"]
        empty = np.ones(len(base), bool)
        empty[inb] = False
        assert np.isnan(got[empty]).all() and not np.isnan(got[inb]).any()
        assert np.array_equal(got[inb], pr.cpu().numpy())
    assert np.array_equal(d_pred["all"], multi[0][1].cpu().numpy())
    assert np.isfinite(d_pred["all_sin"]).all() and np.isfinite(d_mfe["all_reverse"]).all()


def test_early_prediction_per_time_step_in_one_launch():
    """f2, early_prediction.get_predictions (visualization/early_prediction.py:19-81): the per-time-step loop of one light
    curve as ONE launch over its T prefixes.  Vanilla: every prefix equals the reference's calling convention -- the dense
    un-packed slice X[:i] expanded to a batch -- through the module's forward, to the probability bar; Bayesian /
    variational: (T, S, C) with the percentiles of :72-79, the last prefix distributed like the full curve."""
    import os
    import util
    from supernnova_amd.visualization import early_prediction as ep
    z = dict(np.load(os.path.join(util.GOLDEN, "g10_predictions.npz")))
    models = {}
    for m in ("vanilla", "variational", "bayesian"):
        _, st, data, rnn = _setup_flat(z, m)
        rnn.draws = "philox"
        models[f"{m}_S_0_CLF_2"] = rnn
    st.num_inference_samples = 64
    k = int(np.argmax(z["lengths"]))
    X, target = data[k][0], data[k][1]
    torch.manual_seed(3)
    d_pred, X_out = ep.get_predictions(st, models, X, target)
    T = X.shape[0]
    assert np.array_equal(X_out, X) and set(d_pred) == set(models)
    van = models["vanilla_S_0_CLF_2"]
    with torch.no_grad():
        Xt = torch.from_numpy(X).to(DEV)[:, None, :]
        want = np.stack([torch.softmax(van(Xt[:i].expand(i, 1, X.shape[1])), -1).cpu().numpy() for i in range(1, T + 1)])
    got = d_pred["vanilla_S_0_CLF_2"]["prob"]
    assert got.shape == (T, 1, 2)
    assert (np.abs(got - want) / want).max() < util.PROB_RTOL
    for name in ("variational_S_0_CLF_2", "bayesian_S_0_CLF_2"):
        d = d_pred[name]
        assert d["prob"].shape == (T, 64, 2) and d["prob"].std(1).min() > 0
        np.testing.assert_allclose(d["prob"].sum(-1), 1.0, atol=1e-6)
        assert np.array_equal(d["median"], np.median(d["prob"], 1))
        assert np.array_equal(d["perc_16"], np.percentile(d["prob"], 16, axis=1))
        assert (d["perc_2"] <= d["perc_16"]).all() and (d["perc_84"] <= d["perc_98"]).all()
    # an out-of-distribution transform goes through get_data_batch as in the reference (:23)
    d_rev, X_rev = ep.get_predictions(st, {"vanilla_S_0_CLF_2": van}, X, target, OOD="reverse")
    assert not np.array_equal(X_rev, X) and d_rev["vanilla_S_0_CLF_2"]["prob"].shape == (T, 1, 2)
