"""GPU: the module API (drop-in boundary) against the reference's golden vectors.

With `draws = "host"` the modules draw eps / masks on the host with the reference's torch calls
in the reference's order, so seeding exactly like the reference run that produced the fixture
(tools/make_golden.py) must reproduce its logits to fp32 rounding -- the parity mode of
BASELINE.json's north_star (probabilities within 1e-5 relative)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence

import oracle
from supernnova_amd import layout
from util import load_golden, flat, rel_err, softmax_np, make_curves, PROB_RTOL, LOGIT_ATOL
from test_host_logic import settings

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(name, sd, D, **kw):
    from supernnova_amd.utils import training_utils as tu
    rnn = tu.get_model(settings(model=name, **kw), D)
    rnn.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return rnn.to(DEV).eval()


def _packed(g, xk="x", lk="lengths"):
    return pack_padded_sequence(torch.from_numpy(g[xk]).to(DEV), torch.from_numpy(g[lk].astype(np.int64)))


def _close(logits, ref):
    logits = logits.detach().cpu().numpy()
    np.testing.assert_allclose(logits, ref, atol=LOGIT_ATOL, rtol=0)
    assert rel_err(softmax_np(logits), softmax_np(ref)) < PROB_RTOL


def test_vanilla_module():
    from supernnova_amd.validation import validate_rnn as vr
    g = load_golden("g1_vanilla_mean")
    rnn = _model("vanilla", g["sd"], 24)
    with torch.no_grad():
        _close(rnn(_packed(g)), g["logits"])
        _close(rnn(torch.from_numpy(g["x"]).to(DEV)), g["logits_dense"])
        _close(rnn(_packed(g, "x_b1", "lengths_b1")), g["logits_b1"])
        preds, tgt = vr.get_batch_predictions(rnn, _packed(g), torch.zeros(16, dtype=torch.long, device=DEV))
    assert preds.dtype == np.float32 and rel_err(preds, g["probs"]) < PROB_RTOL


@pytest.mark.parametrize("prec", ["f32", "f16x2"])
@pytest.mark.parametrize("tag,kw,D", [("mean", {}, 24), ("standard", dict(rnn_output_option="standard", nb_classes=4), 26)])
def test_variational_module_seed_replay(tag, kw, D, prec):
    """Host-drawn masks in the reference's order reproduce the reference's stochastic logits; in f16x2 the kernel
    applies the mask bits on the fp16 pieces and the 1/(1-p) factors sit in the weights (same parity bar)."""
    g = load_golden(f"g2_variational_{tag}")
    rnn = _model("variational", g["sd"], D, **kw)
    rnn.draws = "host"
    rnn.inference_precision = prec
    assert rnn.infer_cfg(True).precision == prec
    with torch.no_grad():
        for s in range(g["logits"].shape[0]):
            torch.manual_seed(1000 + s)
            _close(rnn(_packed(g)), g["logits"][s])
        _close(rnn(_packed(g), mean_field_inference=True), g["logits_mfe"])
        torch.manual_seed(77)
        _close(rnn(torch.from_numpy(g["x"]).to(DEV)), g["logits_dense"])


@pytest.mark.parametrize("tag,kw,D", [("mean", {}, 24), ("standard", dict(rnn_output_option="standard", nb_classes=3), 26),
                                      ("h64", dict(hidden_dim=64, nb_classes=5), 24)])
def test_bayesian_module_seed_replay(tag, kw, D):
    g = load_golden(f"g3_bayesian_{tag}")
    rnn = _model("bayesian", g["sd"], D, **kw)
    rnn.draws = "host"
    with torch.no_grad():
        for s in range(g["logits"].shape[0]):
            torch.manual_seed(2000 + s)
            _close(rnn(_packed(g)), g["logits"][s])
            assert abs(float(rnn.kl) - g["kl"][s]) / g["kl"][s] < 1e-5
        _close(rnn(_packed(g), mean_field_inference=True), g["logits_mfe"])
        assert abs(float(rnn.kl) - g["kl_mfe"]) / g["kl_mfe"] < 1e-5
        torch.manual_seed(2100)
        _close(rnn(torch.from_numpy(g["x"]).to(DEV)), g["logits_dense"])


def test_forward_mc_and_ragged_predict_match_sequential_forwards():
    """forward_mc / predict_mc (one launch, MC folded into the batch) == S sequential forwards."""
    from supernnova_amd.validation import validate_rnn as vr
    g = load_golden("g3_bayesian_mean")
    rnn = _model("bayesian", g["sd"], 24)
    S = 5
    eps = rnn.host_eps(S, DEV)
    with torch.no_grad():
        mc = rnn.forward_mc(_packed(g), S, eps=eps).cpu().numpy()            # (B,S,C)
        kl_mc = rnn.kl.cpu().numpy()
        for s in range(S):
            lo = rnn(_packed(g), eps=eps[s:s + 1])
            assert rel_err(mc[:, s], softmax_np(lo.cpu().numpy())) < PROB_RTOL
            assert abs(float(rnn.kl) - kl_mc[s]) == 0.0
        # ragged list-of-curves entry point, unsorted order
        perm = np.random.default_rng(0).permutation(g["x"].shape[1])
        curves = [g["x"][: g["lengths"][b], b] for b in perm]
        wk, _ = rnn.mc_weights(S, len(curves), DEV, eps=eps)
        from supernnova_amd import engine as E
        rows, off, lens = vr.ragged_to_device(curves, DEV)
        plan = E.Plan(rnn.cfg, lens, DEV)
        _, pr = E.lstm_forward(rnn.cfg, plan, E.pack_ragged(rnn.cfg, plan, rows, off), wk, S, sample_major=False)
    assert np.array_equal(pr.cpu().numpy(), mc[perm])


@pytest.mark.parametrize("model,prec", [("bayesian", "f32"), ("variational", "f32"), ("variational", "f16x2")])
def test_sampled_mode_statistics(model, prec):
    """Sampled (Philox) mode vs the reference's predictive mean/std (2000 reference draws, fixture g5).
    Tolerances: per-curve |mean - mean_ref| <= 4*sqrt(2)*std_ref/sqrt(S) for >= 99.9 % of curves
    (+1e-4 absolute floor), pooled mean bias < 1e-3, pooled predictive std within +-10 %."""
    from supernnova_amd.validation import validate_rnn as vr
    g = load_golden(f"g5_mcstats_{model}")
    rnn = _model(model, g["sd"], 24, dropout=float(g["dropout"]))
    rnn.inference_precision = prec
    S = 2000
    torch.manual_seed(123)
    curves = [g["x"][: g["lengths"][b], b] for b in range(g["x"].shape[1])]
    with torch.no_grad():
        probs = vr.predict_mc(rnn, curves, S).cpu().numpy().astype(np.float64)   # (N,S,C)
    mean, std = probs.mean(1), probs.std(1, ddof=1)
    ref_m, ref_s = g["mean"], g["std"]
    n_ref = int(g["n_ref"])
    bound = 4 * np.sqrt(2) * ref_s / np.sqrt(min(S, n_ref)) + 1e-4
    ok = (np.abs(mean - ref_m) <= bound).mean()
    assert ok >= 0.999, ok
    assert abs((mean - ref_m).mean()) < 1e-3
    ratio = std[:, 1].mean() / ref_s[:, 1].mean()
    assert 0.9 < ratio < 1.1, ratio
    # two different seeds give different draws; the same seed reproduces bit-for-bit
    with torch.no_grad():
        torch.manual_seed(5); a = vr.predict_mc(rnn, curves, 4).cpu().numpy()
        torch.manual_seed(5); b = vr.predict_mc(rnn, curves, 4).cpu().numpy()
        torch.manual_seed(6); c = vr.predict_mc(rnn, curves, 4).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    if prec != "f32":   # same seed, same Philox mask bits in both kernels: the precisions agree to the parity bar
        rnn.inference_precision = "f32"
        with torch.no_grad():
            torch.manual_seed(5); a32 = vr.predict_mc(rnn, curves, 4).cpu().numpy()
        assert rel_err(a, a32) < PROB_RTOL


@pytest.mark.parametrize("model,prec", [("bayesian", "f32"), ("bayesian", "f16x2"), ("variational", "f32"),
                                        ("variational", "f16x2")])
def test_sampled_mode_gate(model, prec):
    """The sampled-mode gate exactly as SURVEY 8d writes it: 10 000 curves, S = 50 Philox draws each, against the
    reference's own S = 50 run A of fixture g5_gate_<model> (mini-batches of 5 curves share a weight draw / W_hh row
    mask on both sides, every mini-batch is a launch with a fresh seed): >= 99.9 % of curves within
    4*sqrt(2)*std_ref/sqrt(S), pooled bias < 1e-3, pooled predictive std within +-10 %, KS test of the standardised
    differences against t(S-1) p > 0.01, and a two-sample KS against the reference's own run B vs A p > 0.01."""
    from util import make_curves, sampled_gate, assert_sampled_gate
    from supernnova_amd.validation import validate_rnn as vr
    g = load_golden(f"g5_gate_{model}")
    N, G, T, S = int(g["n"]), int(g["group"]), int(g["T"]), int(g["S"])
    x, lengths = make_curves(int(g["seed"]), N, T, 24)
    rnn = _model(model, g["sd"], 24, dropout=float(g["dropout"]))
    rnn.inference_precision = prec
    torch.manual_seed(777)
    xd = torch.from_numpy(x).to(DEV)
    out = []
    with torch.no_grad():
        for b0 in range(0, N, G):
            idx = np.arange(b0, b0 + G)
            rows = torch.cat([xd[: lengths[b], b] for b in idx])
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths[idx].astype(np.int64))[:-1]]) * 24).to(DEV)
            out.append(vr.predict_mc(rnn, (rows, off, lengths[idx]), S))       # (G, S, 2), one fresh seed per launch
    probs = torch.cat(out).double()
    mean, std = probs.mean(1).cpu().numpy(), probs.std(1, unbiased=True).cpu().numpy()
    res = sampled_gate(mean, std, g["mean"][0], g["std"][0], S, G, mean_ref2=g["mean"][1])
    print(model, prec, res)
    assert_sampled_gate(res)


@pytest.mark.parametrize("fast", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("model", ["vanilla", "bayesian"])
def test_inference_precision_through_the_module_api(model, fast):
    """`rnn.inference_precision = "bf16x3" | "f16x2"`: eval-mode forward, forward_mc and predict_mc run the
    split-precision kernels and agree with the reference fixture and with the exact-fp32 launch of the same draws to
    the parity bar; training is unaffected."""
    from supernnova_amd.validation.validate_rnn import predict_mc
    g = load_golden("g1_vanilla_mean" if model == "vanilla" else "g3_bayesian_mean")
    rnn = _model(model, g["sd"], 24)
    rnn.draws = "host"
    outs = {}
    for prec in ("f32", fast):
        rnn.inference_precision = prec
        with torch.no_grad():
            torch.manual_seed(2000)
            a = rnn(_packed(g)).cpu().numpy()
            torch.manual_seed(2000)
            b = rnn.forward_mc(_packed(g), 3).cpu().numpy()
            curves = [g["x"][: g["lengths"][i], i] for i in range(g["x"].shape[1])]
            torch.manual_seed(2000)
            c = predict_mc(rnn, curves, 3).cpu().numpy()
        outs[prec] = (a, b, c)
    _close(torch.from_numpy(outs[fast][0]), g["logits"][0] if model == "bayesian" else g["logits"])
    for u, v in zip(outs["f32"], outs[fast]):
        assert rel_err(softmax_np(u) if u is outs["f32"][0] else u, softmax_np(v) if v is outs[fast][0] else v) < PROB_RTOL
    np.testing.assert_allclose(outs[fast][1], outs[fast][2], rtol=0, atol=0)   # same launch, two entry points


@pytest.mark.parametrize("cell", ["gru", "rnn"])
@pytest.mark.parametrize("pool,kw,D", [("mean", {}, 24), ("standard", dict(rnn_output_option="standard", nb_classes=3), 26)])
def test_vanilla_module_gru_rnn(cell, pool, kw, D):
    """VanillaRNN with layer_type gru / rnn (vanilla_rnn.py:26-32): the reference's state dict loads key for key and
    the eval forward reproduces its logits; the training paths refuse with a clear error."""
    from supernnova_amd import _lib as L
    g = load_golden(f"g9_{cell}_{pool}")
    rnn = _model("vanilla", g["sd"], D, layer_type=cell, **kw)
    ref_cls = torch.nn.GRU if cell == "gru" else torch.nn.RNN
    ref = ref_cls(D, 32, num_layers=2, bidirectional=True)
    assert [(k, tuple(v.shape)) for k, v in rnn.rnn_layer.state_dict().items()] == \
        [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    with torch.no_grad():
        _close(rnn(_packed(g)), g["logits"])
        _close(rnn(torch.from_numpy(g["x"]).to(DEV)), g["logits_dense"])
        probs = rnn.forward_mc(_packed(g), 2).cpu().numpy()
    assert rel_err(probs[:, 0], g["probs"]) < PROB_RTOL and np.array_equal(probs[:, 0], probs[:, 1])
    # train mode: the GRU / tanh-RNN training kernels (round 2); with dropout 0 the logits are the eval-mode ones
    rnn.train()
    rnn.dropout = 0.0
    out = rnn(_packed(g))
    assert out.requires_grad
    _close(out, g["logits"])


def test_vanilla_train_mode_without_grad_still_applies_dropout():
    """dropout is decided by module.training alone (vanilla_rnn.py:26-33: nn.LSTM(dropout=p) and nn.Dropout
    act in train mode whatever the grad mode); torch.no_grad() only drops the tape.  Same seed -> same Philox masks ->
    the no-grad forward equals the grad-enabled one, and both differ from eval mode."""
    g = load_golden("g1_vanilla_mean")
    rnn = _model("vanilla", g["sd"], 24, dropout=0.3)
    x = _packed(g)
    with torch.no_grad():
        ev = rnn(x).cpu().numpy()
    rnn.train()
    torch.manual_seed(11)
    with_grad = rnn(x).detach().cpu().numpy()
    torch.manual_seed(11)
    with torch.no_grad():
        no_grad = rnn(x).cpu().numpy()
    # (equal to fp32 rounding, not bit for bit: a small batch trains on the half-tile kernels of snn_rec16.hip, whose
    #  16x16x4 MFMAs add the recurrent products in another order than the inference kernel's 32x32x2 ones)
    assert np.abs(with_grad - no_grad).max() < 2e-5
    assert np.abs(no_grad - ev).max() > 1e-3
