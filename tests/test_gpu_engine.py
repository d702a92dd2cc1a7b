"""GPU: engine-level parity of the HIP kernels (through the C ABI) against the
golden vectors from the reference and against the C oracle."""
import numpy as np
import pytest
import torch

import oracle
from supernnova_amd import layout
from util import load_golden, flat, rel_err, make_curves, PROB_RTOL, LOGIT_ATOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eng():
    from supernnova_amd import engine
    return engine


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def _run(E, cfg, x, lengths, wk, S, masks=None):
    plan = E.Plan(cfg, lengths, DEV)
    xp = E.pack_x(cfg, plan, _t(x))
    logits, probs = E.lstm_forward(cfg, plan, xp, wk, S, masks=masks)
    torch.cuda.synchronize()
    return logits.cpu().numpy(), probs.cpu().numpy()


def _check(logits, probs, ref_logits, ref_probs=None):
    assert np.isfinite(logits).all()
    np.testing.assert_allclose(logits, ref_logits, atol=LOGIT_ATOL, rtol=0)
    if ref_probs is not None:
        assert rel_err(probs, ref_probs) < PROB_RTOL


def test_vanilla_golden():
    E = _eng()
    g = load_golden("g1_vanilla_mean")
    cfg = E.Config(24, 32, 2, True, 2, "mean")
    wk = E.prepare_weights(cfg, _t(flat(g["sd"], layout.vanilla_keys(2, 2))))
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, 1)
    _check(lo[0], pr[0], g["logits"], g["probs"])
    T, B, _ = g["x"].shape
    lo, pr = _run(E, cfg, g["x"], np.full(B, T), wk, 1)
    _check(lo[0], pr[0], g["logits_dense"])
    lo, pr = _run(E, cfg, g["x_b1"], g["lengths_b1"], wk, 1)
    _check(lo[0], pr[0], g["logits_b1"], g["probs_b1"])


def test_vanilla_standard_golden():
    E = _eng()
    g = load_golden("g1_vanilla_standard")
    cfg = E.Config(26, 32, 2, True, 3, "standard")
    wk = E.prepare_weights(cfg, _t(flat(g["sd"], layout.vanilla_keys(2, 2))))
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, 1)
    _check(lo[0], pr[0], g["logits"], g["probs"])


@pytest.mark.parametrize("tag,D,C,pool", [("mean", 24, 2, "mean"), ("standard", 26, 4, "standard")])
def test_variational_golden(tag, D, C, pool):
    E = _eng()
    g = load_golden(f"g2_variational_{tag}")
    cfg = E.Config(D, 32, 2, True, C, pool)
    wk = E.prepare_weights(cfg, _t(flat(g["sd"], layout.variational_keys(2, 2))))
    S = g["logits"].shape[0]
    masks = E.Masks(mode=1, m_in=_t(g["mask_in"]), m_whh=_t(g["mask_whh"]),
                    m_mid=_t(g["mask_mid"][:, :, None]), m_out=_t(g["mask_out"]), mid_T=1)
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, S, masks)
    _check(lo, pr, g["logits"], g["probs"])
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, 1)
    _check(lo[0], pr[0], g["logits_mfe"])


@pytest.mark.parametrize("tag,D,H,C,pool", [("mean", 24, 32, 2, "mean"), ("standard", 26, 32, 3, "standard"),
                                            ("h64", 24, 64, 5, "mean")])
def test_bayesian_golden(tag, D, H, C, pool):
    E = _eng()
    g = load_golden(f"g3_bayesian_{tag}")
    cfg = E.Config(D, H, 2, True, C, pool)
    mu = _t(flat(g["sd"], layout.bayesian_keys(2, 2, "mu")))
    rho = _t(flat(g["sd"], layout.bayesian_keys(2, 2, "rho")))
    S = g["logits"].shape[0]
    wk, kl = E.bbb_sample(cfg, mu, rho, S, E.Prior(), eps=_t(g["eps"]))
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, S)
    _check(lo, pr, g["logits"], g["probs"])
    assert rel_err(kl.cpu().numpy(), g["kl"]) < 1e-5
    wk, kl = E.bbb_sample(cfg, mu, rho, 1, E.Prior(), mean_field=True)
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, 1)
    _check(lo[0], pr[0], g["logits_mfe"])
    assert rel_err(kl.cpu().numpy(), [g["kl_mfe"]]) < 1e-5


def test_bayesian_vs_oracle_ragged():
    """Seeded ragged batch (unsorted lengths, N not a multiple of 32) vs the C oracle."""
    E = _eng()
    D, H, C, S, N, T = 24, 32, 2, 4, 203, 40
    cfg = E.Config(D, H, 2, True, C, "mean")
    ocfg = oracle.make_cfg(D, H, 2, True, C, "mean")
    g = np.random.default_rng(5)
    P = cfg.P
    mu = g.uniform(-0.3, 0.3, P).astype(np.float32)
    rho = g.uniform(-4, -2, P - C).astype(np.float32)
    eps = g.standard_normal((S, P - C)).astype(np.float32)
    x, lengths = make_curves(6, N, T, D, min_len=1)
    wk, kl = E.bbb_sample(cfg, _t(mu), _t(rho), S, E.Prior(), eps=_t(eps))
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    for s in range(S):
        w, sigma = oracle.bbb_sample(mu[:P - C], rho, eps[s])
        o = oracle.forward(ocfg, x, lengths, np.concatenate([w, mu[P - C:]]))
        _check(lo[s], pr[s], o["logits"], o["probs"])
        assert abs(float(kl[s]) - oracle.kl(ocfg, w, mu[:P - C], sigma)) / abs(float(kl[s])) < 1e-5
