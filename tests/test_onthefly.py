"""The broker path: DataFrame of photometry -> features -> normalised rows -> probabilities.

Fixture tests/golden/g7_onthefly.npz was produced by the reference's own format_data /
compute_delta_time / pivot_dataframe_single_from_df / normalize_arr code (tools/make_golden.py g7) on
tests/onthefly_lc/example_lc.csv and on a synthetic multi-curve alert batch; tests/golden/onthefly_model/
holds the reference's bundled model data files (weights, cli_args.json, data_norm.json).
Featurisation and normalisation are host code: bit-exact.  Classification: 1e-5 relative on probabilities."""
import glob
import os

import numpy as np
import pandas as pd
import pytest

from util import GOLDEN, load_golden, rel_err, PROB_RTOL

MODEL_FILE = glob.glob(os.path.join(GOLDEN, "onthefly_model", "*.pt"))[0]
CASES = ("example", "synthetic")


def _frame(g):
    return pd.DataFrame({"SNID": g["snid"], "MJD": g["mjd"], "FLUXCAL": g["flux"], "FLUXCALERR": g["err"],
                         "FLT": g["flt"]})


@pytest.mark.parametrize("case", CASES)
def test_format_and_normalise_bit_exact(case):
    from supernnova_amd.validation import validate_onthefly as vo
    g = load_golden("g7_onthefly")[case]
    st = vo.get_settings(MODEL_FILE)
    assert st.arr_norm.shape == (9, 3) and len(st.training_features) == 24
    df = _frame(g)
    before = df.copy()
    fd = vo.format_data(df, st)
    pd.testing.assert_frame_equal(df, before)                     # the caller's frame is not modified
    assert list(fd.columns) == list(st.all_features)
    assert list(fd.index.astype(str)) == list(g["fmt_index"])
    np.testing.assert_array_equal(fd.values.astype(np.float32), g["fmt_values"])
    ids, off, X = vo.featurize(df, st)
    assert list(ids) == list(g["ids"])
    np.testing.assert_array_equal(np.diff(off), g["lengths"])
    np.testing.assert_array_equal(vo.normalize(X, st), g["x_normed"])


def test_row_order_does_not_matter():
    from supernnova_amd.validation import validate_onthefly as vo
    # (rows with EQUAL MJD that sit in the middle of an 8-hour group are the exception, in the reference too: its
    # stable sort keeps their input order and the second one opens a new group; the synthetic case pins that above)
    g = load_golden("g7_onthefly")["example"]
    st = vo.get_settings(MODEL_FILE)
    df = _frame(g)
    a = vo.featurize(df, st)
    b = vo.featurize(df.iloc[np.random.default_rng(3).permutation(len(df))].reset_index(drop=True), st)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def test_missing_host_columns():
    from supernnova_amd.validation import validate_onthefly as vo
    g = load_golden("g7_onthefly")["example"]
    st = vo.get_settings(MODEL_FILE)
    st.redshift = "zspe"                                          # model wants HOSTGAL_SPECZ, frame has none
    with pytest.raises(ValueError):
        vo.format_data(_frame(g), st)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_classify_lcs(case):
    from supernnova_amd.validation import validate_onthefly as vo
    g = load_golden("g7_onthefly")[case]
    ids, preds = vo.classify_lcs(_frame(g), MODEL_FILE, "cuda:0")
    assert list(ids) == list(g["ids"])
    assert preds.shape == g["probs"].shape and preds.dtype == np.float32   # (B, 1, C): vanilla -> one sample
    assert rel_err(preds, g["probs"]) < PROB_RTOL


@pytest.mark.gpu
def test_classify_lcs_wrong_model():
    import json, shutil, tempfile
    from supernnova_amd.validation import validate_onthefly as vo
    g = load_golden("g7_onthefly")["example"]
    with tempfile.TemporaryDirectory() as d:
        for f in glob.glob(os.path.join(GOLDEN, "onthefly_model", "*")):
            shutil.copy(f, d)
        cli = json.load(open(os.path.join(d, "cli_args.json")))
        cli["hidden_dim"] = 16                                    # state dict no longer fits
        json.dump(cli, open(os.path.join(d, "cli_args.json"), "w"))
        with pytest.raises(ValueError):
            vo.classify_lcs(_frame(g), os.path.join(d, os.path.basename(MODEL_FILE)), "cuda:0")


def _rows_by_definition(df, filters, min_dt):
    """The reference's rule, one light curve at a time in plain Python (make_dataset.py:818-832, 885-910): sort by time,
    open a group at the first observation, at delta_time == 0, or more than min_dt after the running group's first
    time; per (group, filter) keep the lowest-error observation (earliest on ties)."""
    out = []
    for snid in sorted(set(df["SNID"])):
        sub = df[df["SNID"] == snid].sort_values("MJD", kind="stable")
        rows, last, prev_t, prev_g = [], 0.0, None, None
        for t, f, fl, er in zip(sub["MJD"], sub["FLT"], sub["FLUXCAL"], sub["FLUXCALERR"]):
            dt = 0.0 if prev_t is None else t - prev_t
            if dt == 0.0 or (t - last) > min_dt:
                g = last = t
            else:
                g = prev_g
            if prev_g is None or g != prev_g:
                rows.append({"_t": g})
            prev_t, prev_g = t, g
            if f in filters and (f not in rows[-1] or er < rows[-1][f][1]):
                rows[-1][f] = (fl, er)
        for i, r in enumerate(rows):
            out.append([snid, 0.0 if i == 0 else r["_t"] - rows[i - 1]["_t"]]
                       + [r.get(f, (0.0, 0.0))[0] for f in filters] + [r.get(f, (0.0, 0.0))[1] for f in filters]
                       + ["".join(f for f in filters if f in r)])
    return out


def test_native_featuriser_on_messy_tables():
    """snn_featurize_plan / snn_featurize_fill against the rule written out per light curve: unsorted rows, duplicate
    times, observations minutes apart, filters the model does not know, one-row curves."""
    from supernnova_amd.validation import validate_onthefly as vo
    st = vo.get_settings(MODEL_FILE)
    filters = list(st.list_filters)
    g = np.random.default_rng(3)
    for trial in range(12):
        n, nc = int(g.integers(1, 1500)), int(g.integers(1, 40))
        df = pd.DataFrame({"SNID": g.integers(0, nc, n).astype(str),
                           "MJD": 57000 + np.round(g.uniform(0, 25, n), int(g.integers(0, 3))),
                           "FLUXCAL": g.normal(0, 50, n), "FLUXCALERR": np.round(g.uniform(1, 5, n), 1),
                           "FLT": g.choice(list("grizY"), n)})
        ids, off, X = vo.featurize(df, st)
        want = _rows_by_definition(df, filters, vo.MIN_DT)
        assert len(want) == len(X) and list(np.repeat(ids, np.diff(off))) == [w[0] for w in want]
        col = {f: j for j, f in enumerate(st.all_features)}
        np.testing.assert_array_equal(X[:, col["delta_time"]], np.array([w[1] for w in want], np.float32))
        for k, f in enumerate(filters):
            np.testing.assert_array_equal(X[:, col[f"FLUXCAL_{f}"]], np.array([w[2 + k] for w in want], np.float32))
            np.testing.assert_array_equal(X[:, col[f"FLUXCALERR_{f}"]],
                                          np.array([w[2 + len(filters) + k] for w in want], np.float32))
        for c in st.list_filters_combination:
            np.testing.assert_array_equal(X[:, col[c]], np.array([w[-1] == c for w in want], np.float32))
