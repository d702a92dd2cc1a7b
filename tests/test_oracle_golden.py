"""CPU: the C oracle reproduces every golden vector generated from the reference
(tools/make_golden.py).  This pins the oracle; the GPU tests then compare the
HIP path with the oracle and with the same fixtures."""
import numpy as np
import pytest

import oracle
from supernnova_amd import layout
from util import load_golden, flat, LOGIT_ATOL


def _cfg(sd_shapes, D, H, C, pool):
    return oracle.make_cfg(D, H, 2, True, C, pool)


def test_vanilla_mean_golden():
    g = load_golden("g1_vanilla_mean")
    cfg = oracle.make_cfg(24, 32, 2, True, 2, "mean")
    w = flat(g["sd"], layout.vanilla_keys(2, 2))
    o = oracle.forward(cfg, g["x"], g["lengths"], w)
    np.testing.assert_allclose(o["logits"], g["logits"], atol=LOGIT_ATOL, rtol=0)
    np.testing.assert_allclose(o["probs"], g["probs"], rtol=1e-5, atol=0)
    T, B, _ = g["x"].shape
    od = oracle.forward(cfg, g["x"], np.full(B, T), w)
    np.testing.assert_allclose(od["logits"], g["logits_dense"], atol=LOGIT_ATOL, rtol=0)
    o1 = oracle.forward(cfg, g["x_b1"], g["lengths_b1"], w)
    np.testing.assert_allclose(o1["logits"], g["logits_b1"], atol=LOGIT_ATOL, rtol=0)


def test_vanilla_standard_golden():
    g = load_golden("g1_vanilla_standard")
    cfg = oracle.make_cfg(26, 32, 2, True, 3, "standard")
    o = oracle.forward(cfg, g["x"], g["lengths"], flat(g["sd"], layout.vanilla_keys(2, 2)))
    np.testing.assert_allclose(o["logits"], g["logits"], atol=LOGIT_ATOL, rtol=0)


@pytest.mark.parametrize("tag,D,C,pool", [("mean", 24, 2, "mean"), ("standard", 26, 4, "standard")])
def test_variational_golden(tag, D, C, pool):
    g = load_golden(f"g2_variational_{tag}")
    cfg = oracle.make_cfg(D, 32, 2, True, C, pool)
    w = flat(g["sd"], layout.variational_keys(2, 2))
    for s in range(g["logits"].shape[0]):
        o = oracle.forward(cfg, g["x"], g["lengths"], w, g["mask_in"][s], g["mask_whh"][s],
                           g["mask_mid"][s], g["mask_out"][s])
        np.testing.assert_allclose(o["logits"], g["logits"][s], atol=LOGIT_ATOL, rtol=0)
    o = oracle.forward(cfg, g["x"], g["lengths"], w)
    np.testing.assert_allclose(o["logits"], g["logits_mfe"], atol=LOGIT_ATOL, rtol=0)


@pytest.mark.parametrize("tag,D,H,C,pool", [("mean", 24, 32, 2, "mean"), ("standard", 26, 32, 3, "standard"),
                                            ("h64", 24, 64, 5, "mean"), ("h128_d78", 78, 128, 14, "mean")])
def test_bayesian_golden(tag, D, H, C, pool):
    g = load_golden(f"g3_bayesian_{tag}")
    cfg = oracle.make_cfg(D, H, 2, True, C, pool)
    mu = flat(g["sd"], layout.bayesian_keys(2, 2, "mu"))
    rho = flat(g["sd"], layout.bayesian_keys(2, 2, "rho"))
    n = rho.size
    for s in range(g["logits"].shape[0]):
        w, sigma = oracle.bbb_sample(mu[:n], rho, g["eps"][s])
        kl = oracle.kl(cfg, w, mu[:n], sigma)
        assert abs(kl - g["kl"][s]) / abs(g["kl"][s]) < 2e-6
        o = oracle.forward(cfg, g["x"], g["lengths"], np.concatenate([w, mu[n:]]))
        np.testing.assert_allclose(o["logits"], g["logits"][s], atol=LOGIT_ATOL, rtol=0)
    w, sigma = oracle.bbb_sample(mu[:n], rho, None)
    assert abs(oracle.kl(cfg, w, mu[:n], sigma) - g["kl_mfe"]) / abs(g["kl_mfe"]) < 2e-6
    o = oracle.forward(cfg, g["x"], g["lengths"], np.concatenate([w, mu[n:]]))
    np.testing.assert_allclose(o["logits"], g["logits_mfe"], atol=LOGIT_ATOL, rtol=0)


@pytest.mark.parametrize("cell", ["gru", "rnn"])
@pytest.mark.parametrize("pool,D,C", [("mean", 24, 2), ("standard", 26, 3)])
def test_gru_and_rnn_layer_types_golden(cell, pool, D, C):
    """VanillaRNN with layer_type gru / rnn (vanilla_rnn.py:26-32): the oracle's GRU and tanh-RNN cells against
    the reference's torch.nn.GRU / torch.nn.RNN logits, packed and dense input."""
    g = load_golden(f"g9_{cell}_{pool}")
    cfg = oracle.make_cfg(D, 32, 2, True, C, pool, cell=cell)
    w = flat(g["sd"], layout.vanilla_keys(2, 2))
    assert w.size == oracle.param_count(cfg)
    o = oracle.forward(cfg, g["x"], g["lengths"], w)
    np.testing.assert_allclose(o["logits"], g["logits"], atol=LOGIT_ATOL, rtol=0)
    np.testing.assert_allclose(o["probs"], g["probs"], rtol=1e-5, atol=0)
    T, B, _ = g["x"].shape
    od = oracle.forward(cfg, g["x"], np.full(B, T), w)
    np.testing.assert_allclose(od["logits"], g["logits_dense"], atol=LOGIT_ATOL, rtol=0)
