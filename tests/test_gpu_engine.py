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
                                            ("h64", 24, 64, 5, "mean"), ("h128_d78", 78, 128, 14, "mean")])
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


def _random_model(cfg, seed, scale=0.3):
    g = np.random.default_rng(seed)
    return g.uniform(-scale, scale, cfg.P).astype(np.float32)


@pytest.mark.parametrize("D,H,L,bidir,C,pool,N,T,min_len", [
    (24, 32, 1, True, 2, "mean", 40, 12, 1),        # single layer
    (24, 32, 3, True, 3, "standard", 33, 9, 1),     # three layers (y scratch ping-pong), standard pooling
    (24, 32, 2, False, 2, "mean", 50, 15, 1),       # unidirectional
    (23, 32, 2, True, 2, "mean", 31, 10, 2),        # odd D (zero-padded k-step)
    (25, 64, 2, True, 7, "standard", 20, 8, 1),     # H=64, odd D, 7 classes
    (24, 32, 2, True, 20, "mean", 64, 40, 40),      # 20 classes, all curves full length
    (24, 32, 2, True, 2, "mean", 1, 1, 1),          # one curve of one step
    (24, 32, 2, True, 2, "mean", 97, 300, 250),     # long sequences (T = 300)
])
def test_edge_configs_vs_oracle(D, H, L, bidir, C, pool, N, T, min_len):
    E = _eng()
    cfg = E.Config(D, H, L, bidir, C, pool)
    assert cfg.supported()
    ocfg = oracle.make_cfg(D, H, L, bidir, C, pool)
    w = _random_model(cfg, 11, 0.3 if T < 100 else 0.15)
    x, lengths = make_curves(12, N, T, D, min_len=min_len)
    wk = E.prepare_weights(cfg, _t(w))
    lo, pr = _run(E, cfg, x, lengths, wk, 1)
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])


def test_masks_all_layers_vs_oracle():
    """Variational masks with L=3 / standard pooling / per-sample masks, vs the oracle."""
    E = _eng()
    D, H, L, C, N, T, S = 26, 32, 3, 4, 45, 14, 3
    cfg = E.Config(D, H, L, True, C, "standard")
    ocfg = oracle.make_cfg(D, H, L, True, C, "standard")
    g = np.random.default_rng(3)
    w = _random_model(cfg, 4)
    x, lengths = make_curves(5, N, T, D, min_len=1)
    p = 0.2
    bern = lambda *s: (g.random(s) < 1 - p).astype(np.float32) / np.float32(1 - p)
    m_in, m_whh, m_mid, m_out = bern(S, N, D), bern(S, L, 4 * H), bern(S, L - 1, 1, N, 2 * H), bern(S, N, cfg.F)
    masks = E.Masks(mode=1, m_in=_t(m_in), m_whh=_t(m_whh), m_mid=_t(m_mid), m_out=_t(m_out), mid_T=1)
    lo, pr = _run(E, cfg, x, lengths, E.prepare_weights(cfg, _t(w)), S, masks)
    for s in range(S):
        o = oracle.forward(ocfg, x, lengths, w, m_in[s], m_whh[s], m_mid[s], m_out[s])
        _check(lo[s], pr[s], o["logits"], o["probs"])


def test_large_batch_properties():
    """At a size the oracle cannot cover: (a) permutation equivariance -- shuffling the curves permutes the
    outputs bit-for-bit (tiling / sorting / scatter are consistent); (b) a sampled subset matches the oracle."""
    E = _eng()
    D, H, C, S, N, T = 24, 32, 2, 3, 20011, 40
    cfg = E.Config(D, H, 2, True, C, "mean")
    ocfg = oracle.make_cfg(D, H, 2, True, C, "mean")
    g = np.random.default_rng(8)
    mu = g.uniform(-0.3, 0.3, cfg.P).astype(np.float32)
    rho = g.uniform(-4, -2, cfg.P - C).astype(np.float32)
    eps = g.standard_normal((S, cfg.P - C)).astype(np.float32)
    x, lengths = make_curves(9, N, T, D, min_len=3)
    wk, _ = E.bbb_sample(cfg, _t(mu), _t(rho), S, E.Prior(), eps=_t(eps))
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    perm = g.permutation(N)
    lo2, pr2 = _run(E, cfg, np.ascontiguousarray(x[:, perm]), lengths[perm], wk, S)
    assert np.array_equal(pr2, pr[:, perm]) and np.array_equal(lo2, lo[:, perm])
    sub = g.choice(N, 64, replace=False)
    for s in range(S):
        w, _ = oracle.bbb_sample(mu[:cfg.P - C], rho, eps[s])
        o = oracle.forward(ocfg, np.ascontiguousarray(x[:, sub]), lengths[sub], np.concatenate([w, mu[cfg.P - C:]]))
        _check(lo[s][sub], pr[s][sub], o["logits"], o["probs"])


@pytest.mark.parametrize("D,C,pool,N,T", [(78, 14, "mean", 70, 300), (76, 20, "standard", 33, 25), (24, 2, "mean", 40, 40)])
def test_h128_two_pass_kernel_vs_oracle(D, C, pool, N, T):
    """Config 5 shape (hidden 128, 6 filters + redshift, multi-class, 300-step sequences) on the two-pass
    kernel, with variational masks, vs the C oracle."""
    E = _eng()
    H, L, S = 128, 2, 2
    cfg = E.Config(D, H, L, True, C, pool)
    assert cfg.supported()
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    g = np.random.default_rng(21)
    w = _random_model(cfg, 22, 0.08)
    x, lengths = make_curves(23, N, T, D, min_len=max(1, T // 3))
    p = 0.1
    bern = lambda *s: (g.random(s) < 1 - p).astype(np.float32) / np.float32(1 - p)
    m_in, m_whh, m_mid, m_out = bern(S, N, D), bern(S, L, 4 * H), bern(S, L - 1, 1, N, 2 * H), bern(S, N, cfg.F)
    masks = E.Masks(mode=1, m_in=_t(m_in), m_whh=_t(m_whh), m_mid=_t(m_mid), m_out=_t(m_out), mid_T=1)
    wk = E.prepare_weights(cfg, _t(w))
    lo, pr = _run(E, cfg, x, lengths, wk, S, masks)
    for s in range(S):
        o = oracle.forward(ocfg, x, lengths, w, m_in[s], m_whh[s], m_mid[s], m_out[s])
        _check(lo[s], pr[s], o["logits"], o["probs"])
    lo, pr = _run(E, cfg, x, lengths, wk, 1)
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])


@pytest.mark.parametrize("D,C,pool,L,N,T", [(24, 2, "mean", 2, 70, 40), (26, 5, "standard", 2, 97, 17),
                                            (24, 3, "mean", 1, 64, 9), (24, 2, "standard", 3, 131, 23),
                                            (24, 2, "mean", 2, 33, 1)])
def test_paired_tile_kernel_matches_single_tile_kernel_and_oracle(D, C, pool, L, N, T):
    """H=32 bidirectional inference runs the two-tiles-per-workgroup pipeline (snn_forward_pair.hip) at every size: a
    batch of <= 32 curves as a tile with an empty partner, pairs of equally long curves through its UNIFORM
    instantiation.  The same curves must give bit-identical results whichever partner, position and instantiation they
    meet (odd tile counts, partners of different length, 1..3 layers, both poolings), and must match the oracle (which
    is what ties the pair kernel, with its exp2 factors folded into the weights, to the reference)."""
    E = _eng()
    H, S = 32, 2
    cfg = E.Config(D, H, L, True, C, pool)
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    w = _random_model(cfg, 31, 0.15)
    x, lengths = make_curves(32, N, T, D, min_len=1)
    wk = E.prepare_weights(cfg, _t(w))
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    for a in range(0, N, 32):
        lo1, pr1 = _run(E, cfg, np.ascontiguousarray(x[:, a:a + 32]), lengths[a:a + 32], wk, S)
        assert np.array_equal(lo1, lo[:, a:a + 32]) and np.array_equal(pr1, pr[:, a:a + 32])
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])
    _check(lo[1], pr[1], o["logits"], o["probs"])


@pytest.mark.parametrize("D,H,bidir", [(8, 32, True), (14, 64, True), (42, 32, True), (44, 64, False), (76, 32, True),
                                       (78, 64, True), (42, 128, True), (30, 32, False), (78, 128, False)])
def test_feature_width_buckets_vs_oracle(D, H, bidir):
    """Any input width up to 78 runs: the layer-0 k-steps are zero-padded up to the next instantiated width
    (snn_layout.h: 24, 26, 48, 76, 78), for 1..6 filters with or without redshift, both direction counts."""
    E = _eng()
    L, C, S, N, T = 2, 3, 1, 45, 11
    cfg = E.Config(D, H, L, bidir, C, "mean")
    assert cfg.supported()
    ocfg = oracle.make_cfg(D, H, L, bidir, C, "mean")
    w = _random_model(cfg, 41, 0.12)
    x, lengths = make_curves(42, N, T, D, min_len=1)
    lo, pr = _run(E, cfg, x, lengths, E.prepare_weights(cfg, _t(w)), S)
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])


def test_unsupported_shapes_fail_loudly():
    E = _eng()
    from supernnova_amd import _lib
    for cfg in (E.Config(80, 32, 2, True, 2, "mean"), E.Config(24, 48, 2, True, 2, "mean"),
                E.Config(24, 256, 2, True, 2, "mean")):
        assert not cfg.supported()
        with pytest.raises(_lib.SnnError):
            E.lstm_forward_workspace_bytes(cfg, 10) if hasattr(E, "lstm_forward_workspace_bytes") else \
                _lib.check(_lib.load().snn_lstm_forward_workspace_bytes(__import__("ctypes").byref(cfg.c), 10), "ws")


@pytest.mark.parametrize("prec", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("D,C,pool,L,N,T", [(24, 2, "mean", 2, 70, 40), (26, 5, "standard", 2, 97, 17),
                                            (24, 3, "mean", 1, 64, 9), (24, 2, "standard", 3, 131, 23),
                                            (42, 2, "mean", 2, 50, 12), (78, 14, "mean", 2, 40, 30),
                                            (24, 2, "mean", 2, 21, 40), (24, 2, "mean", 2, 1, 1),
                                            (24, 2, "mean", 2, 97, 300), (78, 14, "standard", 2, 33, 300)])
def test_split_precision_vs_oracle(prec, D, C, pool, L, N, T):
    """SNN_PREC_BF16X3: every fp32 product rebuilt from six bf16 piece products on the bf16 matrix pipe;
    SNN_PREC_F16X2: from three fp16 piece products.  Same parity bar as the fp32 kernels (logits 5e-6 absolute,
    probabilities 1e-5 relative) against the C oracle, and an error against the float64 oracle of the same size
    as the fp32 kernel's own."""
    E = _eng()
    H, S = 32, 2
    cfg32 = E.Config(D, H, L, True, C, pool)
    cfg = cfg32.with_precision(prec)
    assert cfg.supported()
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    w = _random_model(cfg, 31, 0.15)
    x, lengths = make_curves(32, N, T, D, min_len=1)
    lo, pr = _run(E, cfg, x, lengths, E.prepare_weights(cfg, _t(w)), S)
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])
    _check(lo[1], pr[1], o["logits"], o["probs"])
    lo32, _ = _run(E, cfg32, x, lengths, E.prepare_weights(cfg32, _t(w)), 1)
    o64 = oracle.forward(ocfg, x, lengths, w, f64=True)["logits"]
    e3, e32 = np.abs(lo[0] - o64).max(), np.abs(lo32[0] - o64).max()
    assert e3 < 3 * e32 + 2e-7, (e3, e32)


@pytest.mark.parametrize("prec", ["bf16x3", "f16x2"])
def test_split_precision_large_batch_properties(prec):
    """bf16x3 / f16x2 at a size with many tile groups spread over the per-XCD queues: shuffling the curves permutes
    the outputs bit-for-bit, and the probabilities agree with the exact-fp32 launch to the parity bar."""
    E = _eng()
    D, H, C, S, N, T = 24, 32, 2, 3, 9001, 40
    cfg32 = E.Config(D, H, 2, True, C, "mean")
    cfg = cfg32.with_precision(prec)
    g = np.random.default_rng(18)
    mu = g.uniform(-0.3, 0.3, cfg.P).astype(np.float32)
    rho = g.uniform(-4, -2, cfg.P - C).astype(np.float32)
    eps = g.standard_normal((S, cfg.P - C)).astype(np.float32)
    x, lengths = make_curves(19, N, T, D, min_len=3)
    wk, kl = E.bbb_sample(cfg, _t(mu), _t(rho), S, E.Prior(), eps=_t(eps))
    wk32, kl32 = E.bbb_sample(cfg32, _t(mu), _t(rho), S, E.Prior(), eps=_t(eps))
    assert torch.equal(kl, kl32)                       # the KL does not depend on the weight image format
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    perm = g.permutation(N)
    lo2, pr2 = _run(E, cfg, np.ascontiguousarray(x[:, perm]), lengths[perm], wk, S)
    assert np.array_equal(pr2, pr[:, perm]) and np.array_equal(lo2, lo[:, perm])
    lo32, pr32 = _run(E, cfg32, x, lengths, wk32, S)
    np.testing.assert_allclose(lo, lo32, atol=LOGIT_ATOL, rtol=0)
    assert rel_err(pr, pr32) < PROB_RTOL


@pytest.mark.parametrize("D,C,pool,L,N,T", [(24, 2, "mean", 2, 70, 40), (26, 5, "standard", 2, 97, 17),
                                            (24, 3, "mean", 1, 64, 9), (24, 2, "standard", 3, 131, 23),
                                            (24, 2, "mean", 2, 33, 1), (78, 14, "mean", 2, 161, 12)])
@pytest.mark.parametrize("prec", ["bf16x3", "f16x2"])
def test_split_precision_results_do_not_depend_on_tiling(prec, D, C, pool, L, N, T):
    """bf16x3 over more than one tile runs the two-tiles-per-workgroup pipeline (snn_forward_b3p.hip); a batch of
    <= 32 curves runs the one-tile kernel (snn_forward_b3.hip).  Same products in the same order: the same curves
    must give bit-identical results through both (odd tile counts, groups of 1..4 tiles, partners of different
    length, 1..3 layers, both poolings).  f16x2 has one kernel (snn_forward_h2.hip) and the same property."""
    E = _eng()
    H, S = 32, 2
    cfg = E.Config(D, H, L, True, C, pool).with_precision(prec)
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    w = _random_model(cfg, 33, 0.15)
    x, lengths = make_curves(34, N, T, D, min_len=1)
    wk = E.prepare_weights(cfg, _t(w))
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    for a in range(0, N, 32):
        lo1, pr1 = _run(E, cfg, np.ascontiguousarray(x[:, a:a + 32]), lengths[a:a + 32], wk, S)
        assert np.array_equal(lo1, lo[:, a:a + 32]) and np.array_equal(pr1, pr[:, a:a + 32])
    o = oracle.forward(ocfg, x, lengths, w)
    _check(lo[0], pr[0], o["logits"], o["probs"])


def _fold_keep_scale(cfg, w, p):
    """What SNN_PREC_F16X2 asks of the caller of a masked launch (include/snn_hip.h): 1/(1-p) inside W_ih of every
    layer and W_hh of the forward directions."""
    w = w.copy()
    s, H, off = np.float32(1) / np.float32(1 - p), cfg.hidden_size, 0
    for l in range(cfg.num_layers):
        I = cfg.input_size if l == 0 else H * cfg.dirs
        for d in range(cfg.dirs):
            w[off:off + 4 * H * I] *= s
            off += 4 * H * I
            if d == 0:
                w[off:off + 4 * H * H] *= s
            off += 4 * H * H + 8 * H
    return w


@pytest.mark.parametrize("D,C,pool,L,N,T", [(24, 2, "mean", 2, 70, 40), (42, 5, "standard", 2, 97, 17),
                                            (26, 3, "mean", 1, 33, 9), (48, 2, "standard", 3, 131, 23),
                                            (24, 2, "mean", 2, 1, 1)])
def test_f16x2_with_variational_masks_vs_oracle(D, C, pool, L, N, T):
    """Masked f16x2 launches (host-drawn mask arrays): mask bits applied on the fp16 pieces, keep-scale folded into
    the weights -- against the C oracle's masked forward at the fp32 parity bar; odd tile counts, a dead partner
    tile, 1..3 layers, both poolings, both instantiated input widths (XB0 = 2, 3).  Wider inputs are refused."""
    E = _eng()
    H, S, p = 32, 3, 0.1
    cfg32 = E.Config(D, H, L, True, C, pool)
    cfg = cfg32.with_precision("f16x2")
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    g = np.random.default_rng(51)
    w = _random_model(cfg, 52, 0.15)
    x, lengths = make_curves(53, N, T, D, min_len=1)
    bern = lambda *s: (g.random(s) < 1 - p).astype(np.float32) / np.float32(1 - p)
    m_in, m_whh, m_out = bern(S, N, D), bern(S, L, 4 * H), bern(S, N, cfg.F)
    m_mid = bern(S, L - 1, 1, N, 2 * H) if L > 1 else None
    masks = E.Masks(mode=1, m_in=_t(m_in), m_whh=_t(m_whh), m_mid=_t(m_mid) if L > 1 else None, m_out=_t(m_out), mid_T=1)
    lo, pr = _run(E, cfg, x, lengths, E.prepare_weights(cfg, _t(_fold_keep_scale(cfg, w, p))), S, masks)
    for s in range(S):
        o = oracle.forward(ocfg, x, lengths, w, m_in[s], m_whh[s], m_mid[s] if L > 1 else None, m_out[s])
        _check(lo[s], pr[s], o["logits"], o["probs"])


def test_f16x2_masks_refused_for_wide_inputs():
    E = _eng()
    from supernnova_amd import _lib
    cfg = E.Config(78, 32, 2, True, 2, "mean").with_precision("f16x2")
    x, lengths = make_curves(54, 40, 8, 78, min_len=1)
    w = _random_model(cfg, 55, 0.1)
    masks = E.Masks(mode=2, which=0xF, dropout=0.1, seed=3)
    with pytest.raises(_lib.SnnError):
        _run(E, cfg, x, lengths, E.prepare_weights(cfg, _t(w)), 2, masks)


# ---------------------------------------------------------------- layer_type gru / rnn (vanilla_rnn.py:26-32)
@pytest.mark.parametrize("cell", ["gru", "rnn"])
@pytest.mark.parametrize("pool,D,C", [("mean", 24, 2), ("standard", 26, 3)])
def test_gru_rnn_golden(cell, pool, D, C):
    """torch.nn.GRU / torch.nn.RNN stacks of the reference's VanillaRNN: fixture logits / probabilities, packed and dense."""
    E = _eng()
    g = load_golden(f"g9_{cell}_{pool}")
    cfg = E.Config(D, 32, 2, True, C, pool, cell=cell)
    w = flat(g["sd"], layout.vanilla_keys(2, 2))
    assert cfg.supported() and cfg.P == w.size
    wk = E.prepare_weights(cfg, _t(w))
    lo, pr = _run(E, cfg, g["x"], g["lengths"], wk, 1)
    _check(lo[0], pr[0], g["logits"], g["probs"])
    T, B, _ = g["x"].shape
    lo, pr = _run(E, cfg, g["x"], np.full(B, T), wk, 1)
    _check(lo[0], pr[0], g["logits_dense"])


@pytest.mark.parametrize("cell", ["gru", "rnn"])
@pytest.mark.parametrize("D,H,L,bidir,C,pool", [(24, 32, 2, True, 2, "mean"), (26, 64, 1, False, 3, "standard"),
                                                (44, 32, 3, False, 14, "mean"), (78, 64, 2, True, 20, "standard")])
def test_gru_rnn_vs_oracle_ragged(cell, D, H, L, bidir, C, pool):
    """Seeded ragged batches (lengths 1..T, N not a multiple of 32, several weight sets in one launch) vs the C oracle."""
    E = _eng()
    S, N, T = 3, 139, 45
    cfg = E.Config(D, H, L, bidir, C, pool, cell=cell)
    ocfg = oracle.make_cfg(D, H, L, bidir, C, pool, cell=cell)
    assert cfg.P == oracle.param_count(ocfg)
    g = np.random.default_rng(11)
    ws = g.uniform(-0.2, 0.2, (S, cfg.P)).astype(np.float32)   # logits of order 1: LOGIT_ATOL is an absolute bound
    x, lengths = make_curves(12, N, T, D, min_len=1)
    wk = E.prepare_weights_multi(cfg, _t(ws))
    lo, pr = _run(E, cfg, x, lengths, wk, S)
    for s in range(S):
        o = oracle.forward(ocfg, x, lengths, ws[s])
        _check(lo[s], pr[s], o["logits"], o["probs"])


@pytest.mark.parametrize("cell", ["gru", "rnn"])
@pytest.mark.parametrize("D,L,bidir,C,pool,N,T", [(24, 2, True, 2, "mean", 1000, 40), (78, 3, True, 5, "standard", 131, 23),
                                                  (26, 1, False, 3, "mean", 33, 7)])
def test_wave_per_tile_kernel_matches_the_lstm_shaped_one(cell, D, L, bidir, C, pool, N, T):
    """Un-masked GRU / tanh-RNN inference at H = 32 runs wpt_fwd_kernel (snn_forward_wpt.hip: one wave per tile, unit-major
    row tiles, no workgroup barrier); SNN_NO_WPT=1 selects lstm_fwd_kernel<.., CELL>.  Same products in the same order per
    gate pre-activation; the head sums 16 units per lane instead of 4: logits within LOGIT_ATOL, several weight sets,
    tile groups with missing tiles (N / 32 not a multiple of 4), three layers (two y buffers)."""
    import os
    E = _eng()
    S = 2
    cfg = E.Config(D, 32, L, bidir, C, pool, cell=cell)
    g = np.random.default_rng(21)
    ws = g.uniform(-0.2, 0.2, (S, cfg.P)).astype(np.float32)
    x, lengths = make_curves(13, N, T, D, min_len=1)
    wk = E.prepare_weights_multi(cfg, _t(ws))
    res = {}
    for v in (None, "1"):
        if v:
            os.environ["SNN_NO_WPT"] = v
        try:
            lo, pr = _run(E, cfg, x, lengths, wk, S)
            name = E.last_forward_kernel()
        finally:
            os.environ.pop("SNN_NO_WPT", None)
        assert ("wpt_fwd_kernel" in name) == (v is None), name
        res[v] = (lo, pr)
    for s in range(S):
        _check(res[None][0][s], res[None][1][s], res["1"][0][s], res["1"][1][s])


def test_gru_rnn_cells_refuse_what_they_do_not_implement():
    """GRU / tanh-RNN cells run inference and training (tests/test_gpu_train.py), but have no Bayes-by-Backprop path
    (bayesian_rnn.py:47-69 registers the LSTM only), no split-precision kernel and no masked *inference* kernel: those
    entry points refuse the cells with a clear error instead of computing something else."""
    E = _eng()
    from supernnova_amd import _lib as L
    cfg = E.Config(24, 32, 2, True, 2, "mean", cell="gru")
    assert not cfg.with_precision("f16x2").supported()
    x, lengths = make_curves(3, 40, 12, 24, min_len=1)
    plan = E.Plan(cfg, lengths, DEV)
    xp = E.pack_x(cfg, plan, _t(x))
    wk = E.prepare_weights(cfg, torch.zeros(cfg.P, device=DEV))
    with pytest.raises(L.SnnError, match="inference-only"):          # Bayes-by-Backprop exists for the LSTM cell only
        E.bbb_sample(cfg, torch.zeros(cfg.P, device=DEV), torch.zeros(cfg.P - 2, device=DEV), 1, E.Prior())
    masks = E.Masks(mode=L.SNN_MASK_PHILOX, which=8, dropout=0.1, seed=1)
    with pytest.raises(L.SnnError, match="masked"):
        E.lstm_forward(cfg, plan, xp, wk, 1, masks=masks)


@pytest.mark.parametrize("D,C,pool,L,N,T", [(24, 2, "mean", 2, 97, 17), (26, 3, "standard", 3, 64, 9), (24, 2, "mean", 1, 40, 12)])
def test_masked_paired_tile_kernel_matches_single_tile_kernel_and_oracle(D, C, pool, L, N, T):
    """Locked variational masks run the MASKED two-tiles-per-workgroup kernel at every size (<= 32 curves: a tile with an
    empty partner).  Same curves, same mask values: bit-identical results in a multi-tile launch and in launches of one
    tile each, and both match the oracle; Philox masks (keyed by curve index) give the same bits in a 32-curve and in a
    multi-tile launch."""
    E = _eng()
    H, S, p = 32, 3, 0.15
    cfg = E.Config(D, H, L, True, C, pool)
    ocfg = oracle.make_cfg(D, H, L, True, C, pool)
    g = np.random.default_rng(17)
    w = _random_model(cfg, 33, 0.15)
    x, lengths = make_curves(34, N, T, D, min_len=1)
    wk = E.prepare_weights(cfg, _t(w))
    bern = lambda *s: (g.random(s) < 1 - p).astype(np.float32) / np.float32(1 - p)
    m_in, m_whh, m_out = bern(S, N, D), bern(S, L, 4 * H), bern(S, N, cfg.F)
    m_mid = bern(S, max(L - 1, 1), 1, N, 2 * H)

    def masks(a=0, b=N):
        return E.Masks(mode=1, m_in=_t(m_in[:, a:b]), m_whh=_t(m_whh), m_mid=_t(m_mid[:, :, :, a:b]) if L > 1 else None,
                       m_out=_t(m_out[:, a:b]), mid_T=1)
    lo, pr = _run(E, cfg, x, lengths, wk, S, masks())
    for a in range(0, N, 32):
        lo1, pr1 = _run(E, cfg, np.ascontiguousarray(x[:, a:a + 32]), lengths[a:a + 32], wk, S, masks(a, a + 32))
        assert np.array_equal(lo1, lo[:, a:a + 32]) and np.array_equal(pr1, pr[:, a:a + 32])
    for s in range(S):
        o = oracle.forward(ocfg, x, lengths, w, m_in[s], m_whh[s], m_mid[s] if L > 1 else None, m_out[s])
        _check(lo[s], pr[s], o["logits"], o["probs"])
    ph = lambda: E.Masks(mode=2, which=15, dropout=p, seed=5, mid_T=1)
    lo2, _ = _run(E, cfg, x, lengths, wk, S, ph())
    lo3, _ = _run(E, cfg, np.ascontiguousarray(x[:, :32]), lengths[:32], wk, S, ph())
    assert np.array_equal(lo3, lo2[:, :32])


def test_full_shard_size_properties():
    """BASELINE configs[2] at its per-GPU size (1.25 M ragged light curves, lengths U{3..40}, BayesianRNN weights drawn
    per sample, S = 50 draws): (a) a random subset re-run in its own launch gives the same bits (tiling, work queue
    and scatter are consistent at 39063 tiles / 32-bit-overflowing offsets); (b) that subset matches the oracle; (c)
    probabilities are normalised and finite everywhere."""
    E = _eng()
    D, H, C, S, N, T = 24, 32, 2, 50, 1_250_000, 40
    cfg = E.Config(D, H, 2, True, C, "mean")
    ocfg = oracle.make_cfg(D, H, 2, True, C, "mean")
    g = np.random.default_rng(2024)
    lengths = g.integers(3, T + 1, size=N).astype(np.int32)
    off = np.zeros(N, np.int64)
    np.cumsum(lengths[:-1].astype(np.int64) * D, out=off[1:])
    gen = torch.Generator(device=DEV).manual_seed(7)
    rows = torch.randn(int(lengths.sum()), D, device=DEV, generator=gen)
    assert rows.numel() * 4 > 2 ** 31                       # float offsets do not fit 32 bits
    mu = g.uniform(-0.3, 0.3, cfg.P).astype(np.float32)
    rho = g.uniform(-4, -2, cfg.P - C).astype(np.float32)
    eps = g.standard_normal((S, cfg.P - C)).astype(np.float32)
    wk, _ = E.bbb_sample(cfg, _t(mu), _t(rho), S, E.Prior(), eps=_t(eps))
    plan = E.Plan(cfg, lengths, DEV)
    xp = E.pack_ragged(cfg, plan, rows, torch.from_numpy(off).to(DEV))
    lo, pr = E.lstm_forward(cfg, plan, xp, wk, S)
    assert bool(torch.isfinite(pr).all()) and float((pr.sum(-1) - 1).abs().max()) < 1e-6
    sub = np.sort(g.choice(N, 96, replace=False))
    sub[0], sub[-1] = 0, N - 1
    sub_rows = torch.cat([rows[off[i] // D: off[i] // D + lengths[i]] for i in sub])
    sub_off = np.zeros(len(sub), np.int64)
    np.cumsum(lengths[sub][:-1].astype(np.int64) * D, out=sub_off[1:])
    plan2 = E.Plan(cfg, lengths[sub], DEV)
    xp2 = E.pack_ragged(cfg, plan2, sub_rows, torch.from_numpy(sub_off).to(DEV))
    lo2, pr2 = E.lstm_forward(cfg, plan2, xp2, wk, S)
    idx = torch.from_numpy(sub).to(DEV)
    assert torch.equal(lo[:, idx], lo2) and torch.equal(pr[:, idx], pr2)
    x_pad = np.zeros((T, len(sub), D), np.float32)
    host = sub_rows.cpu().numpy()
    for k, i in enumerate(sub):
        a = sub_off[k] // D
        x_pad[:lengths[i], k] = host[a:a + lengths[i]]
    lo2, pr2 = lo2.cpu().numpy(), pr2.cpu().numpy()
    for s in range(0, S, 9):
        w, _ = oracle.bbb_sample(mu[:cfg.P - C], rho, eps[s])
        o = oracle.forward(ocfg, x_pad, lengths[sub], np.concatenate([w, mu[cfg.P - C:]]))
        _check(lo2[s], pr2[s], o["logits"], o["probs"])


@pytest.mark.parametrize("name,prec,D,H,C,N,T,full", [("configs[1] variational", "f32", 24, 32, 2, 1_000_000, 40, False),
                                                      ("configs[1] variational", "f16x2", 24, 32, 2, 1_000_000, 40, False),
                                                      ("configs[4] elasticc-like", "f32", 78, 128, 14, 8192, 300, True)])
def test_other_configs_at_full_size(name, prec, D, H, C, N, T, full):
    """BASELINE configs[1] (VariationalRNN, Philox masks keyed by curve index, 1 M curves) and configs[4] (H = 128, D = 78,
    300-step sequences) at full size, S = 50: a contiguous block of curves re-run in its own launch (curve_base = first
    index, so the mask keys agree) gives the same bits; probabilities finite and normalised."""
    E = _eng()
    S, var = 50, name.startswith("configs[1]")
    cfg = E.Config(D, H, 2, True, C, "mean", precision=prec)
    g = np.random.default_rng(99)
    lengths = (np.full(N, T) if full else g.integers(3, T + 1, size=N)).astype(np.int32)
    off = np.zeros(N, np.int64)
    np.cumsum(lengths[:-1].astype(np.int64) * D, out=off[1:])
    rows = torch.randn(int(lengths.sum()), D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    if var:
        wk = E.prepare_weights(cfg, _t(_random_model(cfg, 5, 0.25)))
    else:
        wk = E.prepare_weights_multi(cfg, _t(g.uniform(-0.1, 0.1, (S, cfg.P))))
    mk = (lambda base: E.Masks(mode=2, which=15, dropout=0.05, seed=11, mid_T=1, curve_base=base)) if var else (lambda base: None)
    plan = E.Plan(cfg, lengths, DEV)
    xp = E.pack_ragged(cfg, plan, rows, torch.from_numpy(off).to(DEV))
    lo, pr = E.lstm_forward(cfg, plan, xp, wk, S, masks=mk(0))
    assert bool(torch.isfinite(pr).all()) and float((pr.sum(-1) - 1).abs().max()) < 1e-6
    a, n = N // 3, 80
    r0, r1 = off[a] // D, off[a + n] // D
    plan2 = E.Plan(cfg, lengths[a:a + n], DEV)
    xp2 = E.pack_ragged(cfg, plan2, rows[r0:r1].contiguous(), torch.from_numpy(off[a:a + n] - off[a]).to(DEV))
    lo2, pr2 = E.lstm_forward(cfg, plan2, xp2, wk, S, masks=mk(a))
    assert torch.equal(lo[:, a:a + n], lo2) and torch.equal(pr[:, a:a + n], pr2)


@pytest.mark.parametrize("D,H,N,T", [(24, 32, 70, 40), (23, 32, 1, 1), (78, 128, 45, 19), (26, 64, 9000, 13)])
def test_pack_image_is_the_documented_layout(D, H, N, T):
    """snn_pack_x (fp32 image), padded and ragged sources: xp[tile_off[tile] + (t*Dp + k)*32 + c] = x[t, order[tile*32+c], k]
    for t < length, 0 in the padding (k >= D, t >= length, missing curves) -- bit for bit, for a handful of tiles
    (steps dealt over 8 workgroups per tile) and for more than 256 tiles (one workgroup per tile)."""
    E = _eng()
    cfg = E.Config(D, H, 2, True, 2, "mean")
    x, lengths = make_curves(21, N, T, D, min_len=1)
    x = x + (np.arange(T)[:, None, None] >= lengths[None, :, None]) * 7.0     # junk in the padding must not leak
    x = x.astype(np.float32)
    plan = E.Plan(cfg, lengths, "cuda:0")
    Dp = cfg.x_step_floats // 32
    assert Dp >= D
    want = np.zeros(plan.packed_floats(cfg), np.float32)
    for tile in range(plan.n_tiles):
        Tt = int(plan.tile_T[tile])
        img = want[plan.tile_off[tile]:plan.tile_off[tile] + Tt * Dp * 32].reshape(Tt, Dp, 32)
        for c in range(32):
            o = int(plan.order[tile * 32 + c])
            if o >= 0:
                img[:lengths[o], :D, c] = x[:lengths[o], o, :]
    got = E.pack_x(cfg, plan, _t(x)).cpu().numpy()
    assert np.array_equal(got, want)
    got_bt = E.pack_x(cfg, plan, _t(np.ascontiguousarray(x.transpose(1, 0, 2))), time_major=False).cpu().numpy()
    assert np.array_equal(got_bt, want)
    rows = np.concatenate([x[:lengths[i], i] for i in range(N)])
    off = np.concatenate([[0], np.cumsum(lengths.astype(np.int64))[:-1]]) * D
    got_r = E.pack_ragged(cfg, plan, _t(rows), torch.from_numpy(off.astype(np.int64)).to("cuda:0")).cpu().numpy()
    assert np.array_equal(got_r, want)


@pytest.mark.parametrize("prec", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("D,N,T", [(24, 70, 40), (23, 1, 1), (42, 300, 9)])
def test_split_precision_pack_image_is_the_documented_layout(prec, D, N, T):
    """snn_pack_x, split-precision images (include/snn_hip.h): per (tile, t) XB blocks of NPC pieces of 64 lanes x 8
    16-bit values; lane (curve = lane & 31, khalf = lane >> 5) holds features 16 b + 8 khalf + e.  f16x2 pieces:
    hi = fp16(x), lo = fp16((x - hi) * 2048); bf16x3: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) --
    round-to-nearest-even, bit for bit; zeros in every padding position."""
    E = _eng()
    cfg = E.Config(D, 32, 2, True, 2, "mean").with_precision(prec)
    x, lengths = make_curves(22, N, T, D, min_len=1)
    plan = E.Plan(cfg, lengths, "cuda:0")
    npc, dt = (2, torch.float16) if prec == "f16x2" else (3, torch.bfloat16)
    XB = cfg.x_step_floats * 2 // (npc * 64 * 8)
    assert XB * 16 >= D
    xt = torch.from_numpy(x)
    hi = xt.to(dt)
    if prec == "f16x2":
        pieces = [hi, ((xt - hi.float()) * 2048.0).to(dt)]
    else:
        r1 = xt - hi.float()
        mid = r1.to(dt)
        pieces = [hi, mid, (r1 - mid.float()).to(dt)]
    pieces = [p.view(torch.int16).numpy() for p in pieces]
    fp32_cfg = cfg.with_precision("f32")
    step_off = (plan.tile_off // (fp32_cfg.x_step_floats)).astype(np.int64)     # first (tile, t) step of each tile
    want = np.zeros((plan.total_steps, XB, npc, 64, 8), np.int16)
    for tile in range(plan.n_tiles):
        for c in range(32):
            o = int(plan.order[tile * 32 + c])
            if o < 0:
                continue
            n = int(lengths[o])
            for k in range(D):
                b, kh, e = k // 16, (k % 16) // 8, k % 8
                for p in range(npc):
                    want[step_off[tile]:step_off[tile] + n, b, p, c + 32 * kh, e] = pieces[p][:n, o, k]
    got = E.pack_x(cfg, plan, _t(x)).view(torch.int16).cpu().numpy().reshape(want.shape)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("prec", ["f32", "f16x2"])
@pytest.mark.parametrize("enforce_sorted", [True, False])
def test_pack_from_packed_sequence_rows_equals_pack_from_padded(prec, enforce_sorted):
    """snn_pack_x_steps reads a PackedSequence's rows in place (step t at step_off[t], curve n in column
    unsorted_indices[n]); the tile image is bit-identical to snn_pack_x over the padded tensor pad_packed_sequence
    would have made -- 1 curve, a partial tile, several tiles; also through the modules (logits bit-identical)."""
    from torch.nn.utils.rnn import pack_padded_sequence
    from supernnova_amd import engine as E
    from supernnova_amd.modules import unpack_packed_sequence
    g = torch.Generator().manual_seed(5)
    for B, T in [(1, 7), (33, 40), (700, 25)]:
        cfg = E.Config(24, 32, 2, True, 2, "mean").with_precision(prec)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        if enforce_sorted:
            lens = lens.sort(descending=True).values
        lens[0 if enforce_sorted else int(torch.randint(0, B, (1,), generator=g))] = T
        x = torch.randn(T, B, 24, generator=g)
        for b in range(B):
            x[lens[b]:, b] = 0
        pk = pack_padded_sequence(x.to(DEV), lens, enforce_sorted=enforce_sorted)
        rows, lengths = unpack_packed_sequence(pk)
        assert np.array_equal(lengths, lens.numpy())
        plan = E.Plan(cfg, lengths, DEV)
        assert torch.equal(E.pack_x(cfg, plan, rows), E.pack_x(cfg, plan, x.to(DEV)))
