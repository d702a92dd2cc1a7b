"""The forward caller of the reference's visualization/early_prediction.py: `get_predictions` :19-81, the
per-time-step prediction loop behind the early-classification plots.  The plots themselves are out of scope.

The reference runs, for every prefix X[:i] of one light curve and every model, `num_inference_samples` forwards (Bayesian,
SWAG: one per draw, :42-61) or one forward over a batch of that many copies (variational, :39-40, :63-66) -- T x S small
launches per model.  Here the T prefixes are T work items of ONE launch per model (validate_rnn.predict_mc_prefixes:
views of the same rows, one plan, one weight draw per MC sample shared by the prefixes), and the percentiles are taken
on the host from the (T, S, C) result exactly as :72-79 does.
"""
import numpy as np
import torch

from ..utils import training_utils as tu
from ..validation import validate_rnn as vr


def _n_samples(model_type, settings):
    # :39 names "variational"; :42 "bayesian" and :52 "swag" loop num_inference_samples times; anything else is one forward
    return settings.num_inference_samples if any(k in model_type for k in ("variational", "bayesian", "swag")) else 1


def get_predictions(settings, dict_rnn, X, target, OOD=None):
    """-> (d_pred, X_array): d_pred[model_type] = {"prob": (T, S, C), "median", "perc_16", "perc_84", "perc_2", "perc_98":
    (T, C)}; X_array = the (T, D) input after the optional OOD transform (early_prediction.py:19-81)."""
    _, X_tensor, *_ = tu.get_data_batch([(X.copy(), target)], [0], settings, OOD=OOD)
    curve = X_tensor[:, 0, :].cpu().numpy().astype(np.float32, copy=False)
    T = curve.shape[0]
    cuts = [np.array([i], np.int64) for i in range(1, T + 1)]
    d_pred = {}
    for model_type, rnn in dict_rnn.items():
        rnn.eval()
        with torch.no_grad():
            per_prefix = vr.predict_mc_prefixes(rnn, [curve], cuts, _n_samples(model_type, settings))
            arr = torch.cat([p for _, p in per_prefix], dim=0).cpu().numpy()          # (T, S, C)
        d_pred[model_type] = {"prob": arr, "median": np.median(arr, axis=1)}
        for q in (16, 84, 2, 98):
            d_pred[model_type][f"perc_{q}"] = np.percentile(arr, q, axis=1)
    return d_pred, X_tensor.squeeze().detach().cpu().numpy()
