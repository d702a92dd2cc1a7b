"""GPU: training half of the hot path (forward-train tape, BPTT, BBB backward, CE, Adam) against the
reference's train_step fixtures (utils/training_utils.py:484-525 driven by tools/make_golden.py):
loss, every parameter gradient, and the parameters after two Adam steps."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence

from supernnova_amd import layout
from util import load_golden, flat, LOGIT_ATOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_RTOL = 1e-4      # grads <= 1e-4 relative (to the tensor's max |grad|; summation order differs)


def _t(a, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(DEV)


def _grad_check(got, ref, name):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max() / scale
    assert err < GRAD_RTOL, f"{name}: {err:.2e}"


def _masks(E, g, model, step):
    if model == "bayesian":
        return None
    if model == "variational":
        return E.Masks(mode=1, m_in=_t(g["mask_in"][step][None]), m_whh=_t(g["mask_whh"][step][None]),
                       m_mid=_t(g["mask_mid"][step][None, :, None]), m_out=_t(g["mask_out"][step][None]), mid_T=1)
    mid = g["mask_mid"][step]                      # (1, T, B, 2H) per-step mask of torch.nn.LSTM(dropout)
    return E.Masks(mode=1, m_mid=_t(mid[None]), m_out=_t(g["mask_out"][step][None]), mid_T=mid.shape[1])


@pytest.mark.parametrize("model", ["vanilla", "variational", "bayesian"])
def test_engine_backward_matches_reference_grads(model):
    from supernnova_amd import engine as E
    g = load_golden(f"g4_train_{model}")
    cfg = E.Config(24, 32, 2, True, 2, "mean")
    B, nb = int(g["batch_size"]), int(g["num_batches"])
    plan = E.Plan(cfg, g["lengths"], DEV)
    xp = E.pack_x(cfg, plan, _t(g["x"]))
    target = _t(g["target"], np.int64)
    masks = _masks(E, g, model, 0)
    sd = g["sd0"]
    if model == "bayesian":
        mu = _t(flat(sd, layout.bayesian_keys(2, 2, "mu")))
        rho = _t(flat(sd, layout.bayesian_keys(2, 2, "rho")))
        eps = _t(g["eps"][0])
        wk, kl, terms, mode = E.bbb_sample_train(cfg, mu, rho, E.Prior(), eps=eps)
    else:
        keys = layout.vanilla_keys(2, 2) if model == "vanilla" else layout.variational_keys(2, 2)
        wk = E.prepare_weights(cfg, _t(flat(sd, keys)))
    logits, tape = E.lstm_forward_train(cfg, plan, xp, wk, masks)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"][0], atol=LOGIT_ATOL, rtol=0)
    ce, dlogits = E.ce_loss_grad(logits, target)
    dflat = E.lstm_backward(cfg, plan, xp, wk, tape, dlogits, masks)
    torch.cuda.synchronize()
    if model == "bayesian":
        loss = float(ce) + float(kl) / (nb * B)
        dmu, drho = E.bbb_backward(cfg, mu, rho, E.Prior(), terms, dflat, mode, eps=eps, kl_scale=1.0 / (nb * B))
        gm, gr = dmu.cpu().numpy(), drho.cpu().numpy()
        off = 0
        for k in layout.bayesian_keys(2, 2, "mu"):
            n = g["grads0"][k.replace("output_layer.", "output_layer.")].size
            _grad_check(gm[off:off + n], g["grads0"][k], k)
            off += n
        off = 0
        for k in layout.bayesian_keys(2, 2, "rho"):
            n = g["grads0"][k].size
            _grad_check(gr[off:off + n], g["grads0"][k], k)
            off += n
    else:
        loss = float(ce)
        gw = dflat.cpu().numpy()
        off = 0
        for k in keys:
            n = g["grads0"][k].size
            _grad_check(gw[off:off + n], g["grads0"][k], k)
            off += n
    assert abs(loss - g["loss"][0]) < 2e-6 * max(1.0, abs(g["loss"][0]))
