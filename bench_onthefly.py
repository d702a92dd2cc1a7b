#!/usr/bin/env python
"""bench_onthefly.py -- secondary benchmark: the broker path validate_onthefly.classify_lcs, end to end.

  python bench_onthefly.py [--curves 200000] [--samples 1] [--reps 3]

A synthetic alert table (DES-like: 4 filters, ~72 photometry rows per light curve over ~130 days, several filters per
night so the 8-hour grouping has work to do) goes through featurize (host, vectorised numpy) -> normalize -> one launch
over all curves x samples, with the model bundled in tests/golden/onthefly_model (the reference's own test model).
Prints one JSON line: light curves/s end to end and the split host featurisation / device classification."""
import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth_alerts(n, seed=0, rows=72):
    g = np.random.default_rng(seed)
    snid = np.repeat(np.arange(n), rows)
    night = np.tile(np.repeat(np.arange(rows // 4), 4), n) * 7.0          # 4 observations per night, weekly cadence
    mjd = 57000.0 + night + g.uniform(0, 0.2, n * rows) + np.repeat(g.uniform(0, 5, n), rows)
    flt = np.tile(np.array(list("griz")), n * rows // 4)
    flux = g.normal(50, 40, n * rows).astype(np.float32)
    err = g.uniform(2, 20, n * rows).astype(np.float32)
    return pd.DataFrame({"SNID": snid.astype(str), "MJD": mjd, "FLUXCAL": flux, "FLUXCALERR": err, "FLT": flt})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curves", type=int, default=200_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", choices=["f32", "f16x2"], default="f32")
    args = ap.parse_args()
    from supernnova_amd.validation import validate_onthefly as vo
    from supernnova_amd.utils import training_utils as tu
    model_dir = os.path.join(ROOT, "tests", "golden", "onthefly_model")
    model_file = [os.path.join(model_dir, f) for f in os.listdir(model_dir) if f.endswith(".pt")][0]
    st = vo.get_settings(model_file)
    st.use_cuda = True
    rnn = tu.get_model(st, len(st.training_features))
    rnn.load_state_dict(torch.load(model_file, map_location="cpu", weights_only=True))
    rnn.cuda().eval()
    rnn.inference_precision = args.precision
    df = synth_alerts(args.curves)
    t_feat, t_all = [], []
    for rep in range(args.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f0 = time.perf_counter()
        ids, off, X = vo.featurize(df.copy(), st)
        t1 = time.perf_counter() - f0
        ids2, probs = vo.classify_lcs(df.copy(), model_file, "cuda", rnn=rnn)
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0 - t1            # one full classify_lcs (it featurises again itself)
        if rep:
            t_feat.append(t1)
            t_all.append(t2)
    assert np.isfinite(probs).all() and probs.shape[0] == args.curves
    ta, tf = float(np.median(t_all)), float(np.median(t_feat))
    print(json.dumps({"metric": "light curves/s classified on the fly (classify_lcs, end to end from a DataFrame)",
                      "value": args.curves / ta, "unit": "light curves/s", "curves": args.curves,
                      "photometry_rows": len(df), "steps_per_curve_after_8h_grouping": float((off[1:] - off[:-1]).mean()),
                      "s_total": ta, "s_featurize_host": tf, "s_normalize_pack_classify": ta - tf,
                      "precision": args.precision, "model": os.path.basename(model_file), "data": "synthetic"}))


if __name__ == "__main__":
    main()
