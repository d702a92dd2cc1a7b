#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  The reference's
model files are imported *by path* (they import nothing but torch; the package
itself is not pip-installed) and driven with the call sequence of
  - validation/validate_rnn.py:59-60,82-83   (forward + softmax)
  - utils/training_utils.py:484-525          (train_step)
to produce input/output vectors.  Random draws are pinned by capture:
  - Bayesian eps: Tensor.normal_ is wrapped to record every draw
    (bayesian_rnn.py:492-523, 608);
  - Variational / vanilla-train masks: replayed after re-seeding with the same
    `empty(shape).bernoulli_(1-p).div_(1-p)` sequence ATen uses, in the draw
    order pinned in SURVEY.md 8a, then VERIFIED here by pushing them through the
    C oracle and comparing with the reference logits.

Only data (inputs, weights, draws, outputs) is written -- no reference source.
"""
import importlib.util
import io
import contextlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/python/supernnova/training"
OUT = os.path.join(ROOT, "tests", "golden")


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, f"{REF}/{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


vanilla_rnn = _load("vanilla_rnn")
variational_rnn = _load("variational_rnn")
bayesian_rnn = _load("bayesian_rnn")


def settings(**kw):
    d = dict(layer_type="lstm", nb_classes=2, hidden_dim=32, num_layers=2, dropout=0.05,
             bidirectional=True, use_cuda=False, rnn_output_option="mean",
             pi=0.75, log_sigma1=-1.0, log_sigma2=-7.0, rho_scale_lower=4.0, rho_scale_upper=3.0,
             log_sigma1_output=-1.0, log_sigma2_output=-7.0,
             rho_scale_lower_output=4.0, rho_scale_upper_output=3.0)
    d.update(kw)
    return SimpleNamespace(**d)


def build(cls, D, st):
    with contextlib.redirect_stdout(io.StringIO()):
        return cls(D, st)


def make_batch(seed, T, B, D, lengths=None):
    g = np.random.default_rng(seed)
    if lengths is None:
        lengths = np.sort(g.integers(3, T + 1, size=B))[::-1].copy()
        lengths[0] = T
    lengths = np.asarray(lengths, dtype=np.int64)
    x = g.standard_normal((T, B, D)).astype(np.float32)
    for b in range(B):
        x[lengths[b]:, b] = 0
    return x, lengths


# ---------------------------------------------------------------- flat layouts
def flat_vanilla(sd, L, dirs):
    out = []
    for l in range(L):
        for d in range(dirs):
            s = "_reverse" if d else ""
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                out.append(sd[f"rnn_layer.{n}_l{l}{s}"].reshape(-1))
    out += [sd["output_layer.weight"].reshape(-1), sd["output_layer.bias"].reshape(-1)]
    return np.concatenate(out).astype(np.float32)


def flat_variational(sd, L, dirs):
    out = []
    for l in range(L):
        for d in range(dirs):
            s = "_reverse" if d else ""
            hh = "weight_hh_l0_raw" if d == 0 else "weight_hh_l0_reverse"
            for n in (f"weight_ih_l0{s}", hh, f"bias_ih_l0{s}", f"bias_hh_l0{s}"):
                out.append(sd[f"rnn_layer{l}.module.{n}"].reshape(-1))
    out += [sd["output_layer.weight"].reshape(-1), sd["output_layer.bias"].reshape(-1)]
    return np.concatenate(out).astype(np.float32)


def flat_bayes(sd, L, dirs, which):
    out = []
    for l in range(L):
        for d in range(dirs):
            s = "_reverse" if d else ""
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                out.append(sd[f"rnn_layer.{n}_{which}_l{l}{s}"].reshape(-1))
    out.append(sd[f"output_layer.{which}"].reshape(-1))
    return np.concatenate(out).astype(np.float32)


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


def ocfg(st, D):
    return oracle.make_cfg(D, st.hidden_dim, st.num_layers, st.bidirectional, st.nb_classes,
                           st.rnn_output_option)


def check(tag, a, b, tol=2e-6):
    err = float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
    print(f"  [{tag}] oracle vs reference max abs diff = {err:.3e}")
    assert err < tol, (tag, err)


def softmax(t):
    return torch.nn.functional.softmax(t, dim=-1).data.cpu().numpy()


def bern(shape, p):
    return torch.empty(shape).bernoulli_(1 - p).div_(1 - p).numpy().copy()


# ---------------------------------------------------------------- G0 init
def g0_init():
    """state_dicts right after a seeded construction: the modules of supernnova_amd must
    reproduce names, shapes, order AND values (same torch RNG calls in the same order)."""
    for model, cls in (("vanilla", vanilla_rnn.VanillaRNN), ("variational", variational_rnn.VariationalRNN),
                       ("bayesian", bayesian_rnn.BayesianRNN)):
        for tag, kw, D in (("mean", {}, 24), ("standard", dict(rnn_output_option="standard", nb_classes=3), 26)):
            torch.manual_seed(7)
            m = build(cls, D, settings(**kw))
            sd = sd_np(m)
            save(f"g0_init_{model}_{tag}", sd=sd, keys=np.array(list(sd.keys())))


# ---------------------------------------------------------------- G1 vanilla
def g1_vanilla():
    st = settings()
    D = 24
    m = build(vanilla_rnn.VanillaRNN, D, st)
    sd = torch.load("/root/reference/tests/onthefly_model/"
                    "vanilla_S_0_CLF_2_R_none_photometry_DF_1.0_N_global_lstm_32x2_0.05_128_True_mean.pt",
                    weights_only=True)
    m.load_state_dict(sd)
    m.eval()
    sdn = sd_np(m)
    w = flat_vanilla(sdn, 2, 2)
    cfg = ocfg(st, D)
    lens = [30, 28, 25, 25, 20, 18, 17, 15, 12, 10, 9, 8, 6, 5, 4, 3]
    x, lengths = make_batch(1, 30, 16, D, lens)
    with torch.no_grad():
        lo = m(pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths)))
        lo_dense = m(torch.from_numpy(x))           # un-packed path: runs over padding, /T
        x1, l1 = make_batch(2, 18, 1, D, [18])
        lo_b1 = m(pack_padded_sequence(torch.from_numpy(x1), torch.from_numpy(l1)))
    o = oracle.forward(cfg, x, lengths, w)
    check("vanilla packed", o["logits"], lo.numpy())
    od = oracle.forward(cfg, x, np.full(16, 30), w)
    check("vanilla dense", od["logits"], lo_dense.numpy())
    o1 = oracle.forward(cfg, x1, l1, w)
    check("vanilla B=1", o1["logits"], lo_b1.numpy())
    save("g1_vanilla_mean", sd=sdn, x=x, lengths=lengths, logits=lo.numpy(), probs=softmax(lo),
         logits_dense=lo_dense.numpy(), x_b1=x1, lengths_b1=l1, logits_b1=lo_b1.numpy(),
         probs_b1=softmax(lo_b1))

    # standard pooling, random weights, C=3, D=26 (redshift features)
    torch.manual_seed(11)
    st2 = settings(rnn_output_option="standard", nb_classes=3)
    D2 = 26
    m2 = build(vanilla_rnn.VanillaRNN, D2, st2)
    m2.eval()
    x2, l2 = make_batch(3, 21, 37, D2)
    with torch.no_grad():
        lo2 = m2(pack_padded_sequence(torch.from_numpy(x2), torch.from_numpy(l2)))
    sd2 = sd_np(m2)
    o2 = oracle.forward(ocfg(st2, D2), x2, l2, flat_vanilla(sd2, 2, 2))
    check("vanilla standard", o2["logits"], lo2.numpy())
    save("g1_vanilla_standard", sd=sd2, x=x2, lengths=l2, logits=lo2.numpy(), probs=softmax(lo2))


# ---------------------------------------------------------------- G2 variational
def var_masks(st, D, B, F):
    """Draw order pinned in SURVEY 8a: (1) input (1,B,D); per layer: W_hh row mask (4H,1)
    [+ locked (1,B,2H) between layers]; (4) output (B,F)."""
    p, H, L = st.dropout, st.hidden_dim, st.num_layers
    dirs = 2 if st.bidirectional else 1
    mask_in = bern((1, B, D), p)[0]
    whh, mid = [], []
    for l in range(L):
        whh.append(bern((4 * H, 1), p)[:, 0])
        if l != L - 1:
            mid.append(bern((1, B, dirs * H), p)[0])
    mask_out = bern((B, F), p)
    return mask_in, np.stack(whh), (np.stack(mid) if mid else None), mask_out


def g2_variational():
    for tag, kw, D, T, B, seed in (("mean", {}, 24, 30, 48, 21),
                                   ("standard", dict(rnn_output_option="standard", nb_classes=4), 26, 17, 33, 22)):
        st = settings(**kw)
        torch.manual_seed(seed)
        m = build(variational_rnn.VariationalRNN, D, st)
        m.eval()
        sdn = sd_np(m)
        cfg = ocfg(st, D)
        F = oracle.feat(cfg)
        w = flat_variational(sdn, 2, 2)
        x, lengths = make_batch(seed, T, B, D)
        packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
        S = 3
        all_masks, all_logits = [], []
        for s in range(S):
            torch.manual_seed(1000 + s)
            with torch.no_grad():
                lo = m(packed)
            torch.manual_seed(1000 + s)
            mk = var_masks(st, D, B, F)
            o = oracle.forward(cfg, x, lengths, w, *mk)
            check(f"variational {tag} s={s}", o["logits"], lo.numpy())
            all_masks.append(mk)
            all_logits.append(lo.numpy())
        with torch.no_grad():
            lo_mfe = m(packed, mean_field_inference=True)
            torch.manual_seed(77)
            lo_dense = m(torch.from_numpy(x))
        torch.manual_seed(77)
        mk_d = var_masks(st, D, B, F)
        check(f"variational {tag} mfe", oracle.forward(cfg, x, lengths, w)["logits"], lo_mfe.numpy())
        check(f"variational {tag} dense",
              oracle.forward(cfg, x, np.full(B, T), w, *mk_d)["logits"], lo_dense.numpy())
        logits = np.stack(all_logits)
        save(f"g2_variational_{tag}", sd=sdn, x=x, lengths=lengths,
             mask_in=np.stack([a[0] for a in all_masks]), mask_whh=np.stack([a[1] for a in all_masks]),
             mask_mid=np.stack([a[2] for a in all_masks]), mask_out=np.stack([a[3] for a in all_masks]),
             logits=logits, probs=softmax(torch.from_numpy(logits)),
             logits_mfe=lo_mfe.numpy(), logits_dense=lo_dense.numpy(),
             dense_mask_in=mk_d[0], dense_mask_whh=mk_d[1], dense_mask_mid=mk_d[2], dense_mask_out=mk_d[3],
             dropout=np.float32(st.dropout))


# ---------------------------------------------------------------- G3 bayesian
class EpsCapture:
    """Record every Tensor.normal_ draw (the reference's only RNG use in BayesianRNN.forward)."""

    def __enter__(self):
        self.draws = []
        self._orig = torch.Tensor.normal_
        cap = self

        def normal_(t, *a, **k):
            r = cap._orig(t, *a, **k)
            cap.draws.append(r.detach().numpy().reshape(-1).copy())
            return r

        torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *a):
        torch.Tensor.normal_ = self._orig

    def flat(self):
        return np.concatenate(self.draws).astype(np.float32)


def bayes_oracle(st, cfg, sdn, x, lengths, eps):
    mu = flat_bayes(sdn, st.num_layers, 2 if st.bidirectional else 1, "mu")
    rho = flat_bayes(sdn, st.num_layers, 2 if st.bidirectional else 1, "rho")
    w, sigma = oracle.bbb_sample(mu, rho, eps)
    kl = oracle.kl(cfg, w, mu, sigma, st.pi, st.log_sigma1, st.log_sigma2,
                   st.log_sigma1_output, st.log_sigma2_output)
    wfull = np.concatenate([w, sdn["output_layer.bias"].reshape(-1)])
    return oracle.forward(cfg, x, lengths, wfull), kl, w


def g3_bayesian():
    cases = (("mean", {}, 24, 30, 40, 31),
             ("standard", dict(rnn_output_option="standard", nb_classes=3), 26, 19, 35, 32),
             ("h64", dict(hidden_dim=64, nb_classes=5), 24, 12, 9, 33),
             ("h128_d78", dict(hidden_dim=128, nb_classes=14), 78, 60, 5, 34))
    for tag, kw, D, T, B, seed in cases:
        st = settings(**kw)
        torch.manual_seed(seed)
        S = {"h64": 2, "h128_d78": 1}.get(tag, 3)   # keep the big-H fixtures small
        m = build(bayesian_rnn.BayesianRNN, D, st)
        m.eval()
        # move rho/mu off their init a little so the test is not init-specific
        with torch.no_grad():
            m.output_layer.bias.uniform_(-0.1, 0.1)
            for n, p in m.named_parameters():
                if "mu" in n:
                    p.mul_(4.0)
        sdn = sd_np(m)
        cfg = ocfg(st, D)
        x, lengths = make_batch(seed, T, B, D)
        packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
        eps_all, logits_all, kl_all = [], [], []
        for s in range(S):
            torch.manual_seed(2000 + s)
            with EpsCapture() as cap, torch.no_grad():
                lo = m(packed)
            eps = cap.flat()
            o, kl, _ = bayes_oracle(st, cfg, sdn, x, lengths, eps)
            check(f"bayesian {tag} s={s}", o["logits"], lo.numpy(), tol=5e-6)
            klr = float(m.kl)
            print(f"  [bayesian {tag} s={s}] kl oracle {kl:.6f} ref {klr:.6f} rel {abs(kl - klr) / abs(klr):.2e}")
            assert abs(kl - klr) / abs(klr) < 2e-6
            eps_all.append(eps)
            logits_all.append(lo.numpy())
            kl_all.append(klr)
        with torch.no_grad():
            lo_mfe = m(packed, mean_field_inference=True)
            kl_mfe = float(m.kl)
            torch.manual_seed(2100)
            with EpsCapture() as cap:
                lo_dense = m(torch.from_numpy(x))
            eps_dense = cap.flat()
        o, kl, _ = bayes_oracle(st, cfg, sdn, x, lengths, None)
        check(f"bayesian {tag} mfe", o["logits"], lo_mfe.numpy(), tol=5e-6)
        assert abs(kl - kl_mfe) / abs(kl_mfe) < 2e-6, (kl, kl_mfe)
        o, _, _ = bayes_oracle(st, cfg, sdn, x, np.full(B, T), eps_dense)
        check(f"bayesian {tag} dense", o["logits"], lo_dense.numpy(), tol=5e-6)
        logits = np.stack(logits_all)
        save(f"g3_bayesian_{tag}", sd=sdn, x=x, lengths=lengths, eps=np.stack(eps_all),
             logits=logits, probs=softmax(torch.from_numpy(logits)), kl=np.array(kl_all, np.float64),
             logits_mfe=lo_mfe.numpy(), kl_mfe=np.float64(kl_mfe),
             logits_dense=lo_dense.numpy(), eps_dense=eps_dense)


# ---------------------------------------------------------------- G4 train step
def ref_train_step(rnn, packed, target, optimizer, batch_size, num_batches, is_bayes):
    """Call sequence of utils/training_utils.py:508-525."""
    criterion = torch.nn.CrossEntropyLoss()
    rnn.train()
    optimizer.zero_grad()
    output = rnn(packed)
    loss = criterion(output.squeeze(), target)
    if is_bayes:
        loss = loss + rnn.kl / (num_batches * batch_size)
    loss.backward()
    grads = {n: p.grad.detach().numpy().copy() for n, p in rnn.named_parameters()}
    optimizer.step()
    return float(loss), grads, output.detach().numpy().copy()


def g4_train():
    D, T, B = 24, 24, 32
    num_batches = 7
    for model in ("bayesian", "variational", "vanilla"):
        st = settings()
        torch.manual_seed(41)
        cls = dict(bayesian=bayesian_rnn.BayesianRNN, variational=variational_rnn.VariationalRNN,
                   vanilla=vanilla_rnn.VanillaRNN)[model]
        m = build(cls, D, st)
        sd0 = sd_np(m)
        x, lengths = make_batch(42, T, B, D)
        target = np.random.default_rng(43).integers(0, 2, size=B).astype(np.int64)
        packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
        wd = 1e-7 if model == "variational" else 0.0   # experiment_settings.py:40-43
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=wd)
        extra = {}
        steps = 2
        losses, grads_all, sds, outs = [], [], [], []
        for it in range(steps):
            torch.manual_seed(4000 + it)
            if model == "bayesian":
                with EpsCapture() as cap:
                    loss, grads, out = ref_train_step(m, packed, torch.from_numpy(target), opt, B, num_batches, True)
                extra.setdefault("eps", []).append(cap.flat())
            else:
                loss, grads, out = ref_train_step(m, packed, torch.from_numpy(target), opt, B, num_batches, False)
                torch.manual_seed(4000 + it)
                if model == "variational":
                    mk = var_masks(st, D, B, 64)
                    for k, v in zip(("mask_in", "mask_whh", "mask_mid", "mask_out"), mk):
                        extra.setdefault(k, []).append(v)
                else:
                    # torch.nn.LSTM(dropout=p) in train mode: element mask on the packed
                    # layer-0 output (sum(len), 2H), then nn.Dropout on the pooled (B, 2H)
                    n_tok = int(lengths.sum())
                    mid_packed = bern((n_tok, 64), st.dropout)
                    mask_out = bern((B, 64), st.dropout)
                    mid = np.zeros((1, T, B, 64), np.float32)
                    bs = packed.batch_sizes.numpy()
                    off = 0
                    for t in range(T):
                        mid[0, t, :bs[t]] = mid_packed[off:off + bs[t]]
                        off += bs[t]
                    extra.setdefault("mask_mid", []).append(mid)
                    extra.setdefault("mask_out", []).append(mask_out)
            losses.append(loss)
            grads_all.append(grads)
            outs.append(out)
            sds.append(sd_np(m))
        # verify the captured draws reproduce the train-mode logits of step 0 through the oracle
        cfg = ocfg(st, D)
        if model == "bayesian":
            o, _, _ = bayes_oracle(st, cfg, sd0, x, lengths, extra["eps"][0])
        elif model == "variational":
            o = oracle.forward(cfg, x, lengths, flat_variational(sd0, 2, 2), extra["mask_in"][0],
                               extra["mask_whh"][0], extra["mask_mid"][0], extra["mask_out"][0])
        else:
            o = oracle.forward(cfg, x, lengths, flat_vanilla(sd0, 2, 2), None, None,
                               extra["mask_mid"][0], extra["mask_out"][0])
        check(f"train {model} step0 logits", o["logits"], outs[0], tol=5e-6)
        arrs = dict(sd0=sd0, x=x, lengths=lengths, target=target,
                    num_batches=np.int64(num_batches), batch_size=np.int64(B),
                    lr=np.float64(1e-3), weight_decay=np.float64(wd), dropout=np.float32(st.dropout),
                    loss=np.array(losses, np.float64), logits=np.stack(outs))
        for it in range(steps):
            arrs[f"grads{it}"] = grads_all[it]
            arrs[f"sd{it + 1}"] = sds[it]
        for k, v in extra.items():
            arrs[k] = np.stack(v)
        save(f"g4_train_{model}", **arrs)


# ---------------------------------------------------------------- G13 train steps at the configs[4] shape
def _subsample(a):
    """Every 8th element of a large tensor (all of a small one): keeps a 1.2 M-parameter fixture at a few hundred KB."""
    a = np.asarray(a).reshape(-1)
    return a if a.size <= 4096 else a[::8]


def g13_train_elasticc():
    """BayesianRNN train step (utils/training_utils.py:508-525) at H=128, D=78 (6 filters + redshift), C=14 -- the shape
    BASELINE configs[4] trains at -- for a short batch and for a batch with 300-step curves.  The model is built from
    a seed (the package's modules reproduce the reference's seeded initial values: fixture g0), eps are the reference's
    own draws after torch.manual_seed(4000) (the package's host_eps makes the same calls in the same order), so the
    fixture holds inputs, loss, logits and -- per parameter tensor -- the gradient and the Adam-updated values at every
    8th element plus the tensor's sum and L2 norm."""
    D, C, H = 78, 14, 128
    for tag, T, lengths in (("t13", 13, None), ("t300", 300, [300, 300, 257, 120, 33, 3])):
        B = 20 if lengths is None else len(lengths)
        st = settings(hidden_dim=H, nb_classes=C)
        torch.manual_seed(131)
        m = build(bayesian_rnn.BayesianRNN, D, st)
        x, lens = make_batch(132, T, B, D, lengths)
        target = np.random.default_rng(133).integers(0, C, size=B).astype(np.int64)
        packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lens))
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        torch.manual_seed(4000)
        loss, grads, out = ref_train_step(m, packed, torch.from_numpy(target), opt, B, 7, True)
        arrs = dict(x=x, lengths=lens, target=target, num_batches=np.int64(7), batch_size=np.int64(B), lr=np.float64(1e-3),
                    loss=np.float64(loss), logits=out, kl=np.float64(float(m.kl)), names=np.array(list(grads.keys())))
        sd1 = sd_np(m)
        for i, (n, gr) in enumerate(grads.items()):
            arrs[f"g{i}"] = _subsample(gr)
            arrs[f"gstat{i}"] = np.array([gr.astype(np.float64).sum(), np.sqrt((gr.astype(np.float64) ** 2).sum()), np.abs(gr).max()])
            arrs[f"p{i}"] = _subsample(sd1[n])
        save(f"g13_train_h128_{tag}", **arrs)
        print(f"g13 {tag}: loss {loss:.6f}, {len(grads)} tensors")


# ---------------------------------------------------------------- G12 GRU / tanh-RNN train steps
def g12_train_cells():
    """VanillaRNN with layer_type gru / rnn (vanilla_rnn.py:26-32 -> torch.nn.GRU / torch.nn.RNN) through the reference's
    train step (utils/training_utils.py:508-525): loss, every gradient, parameters after each of two Adam steps.
    dropout = 0 (no draws to replay); one bidirectional mean-pooled and one unidirectional standard-pooled model each."""
    T, B = 21, 40
    for cell in ("gru", "rnn"):
        for tag, kw, D in (("mean", {}, 24), ("standard", dict(rnn_output_option="standard", bidirectional=False, nb_classes=3,
                                                              hidden_dim=64), 26)):
            st = settings(layer_type=cell, dropout=0.0, **kw)
            torch.manual_seed(121)
            m = build(vanilla_rnn.VanillaRNN, D, st)
            sd0 = sd_np(m)
            x, lengths = make_batch(122, T, B, D)
            target = np.random.default_rng(123).integers(0, st.nb_classes, size=B).astype(np.int64)
            packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0)
            arrs = dict(sd0=sd0, x=x, lengths=lengths, target=target, hidden=np.int64(st.hidden_dim),
                        nb_classes=np.int64(st.nb_classes), bidirectional=np.int64(st.bidirectional),
                        pool=np.array(st.rnn_output_option))
            losses, outs = [], []
            for it in range(2):
                loss, grads, out = ref_train_step(m, packed, torch.from_numpy(target), opt, B, 7, False)
                losses.append(loss)
                outs.append(out)
                arrs[f"grads{it}"] = grads
                arrs[f"sd{it + 1}"] = sd_np(m)
            arrs["loss"] = np.array(losses, np.float64)
            arrs["logits"] = np.stack(outs)
            save(f"g12_train_{cell}_{tag}", **arrs)


# ---------------------------------------------------------------- G5 MC statistics
def g5_mc_stats():
    D, T, B, S = 24, 40, 256, 50
    for model in ("bayesian", "variational"):
        st = settings()
        torch.manual_seed(51)
        cls = dict(bayesian=bayesian_rnn.BayesianRNN, variational=variational_rnn.VariationalRNN)[model]
        m = build(cls, D, st)
        if model == "variational":
            st.dropout = 0.05
        else:
            with torch.no_grad():      # widen the posterior so std is well resolved
                for n, p in m.named_parameters():
                    if "mu" in n:
                        p.mul_(6.0)
        m.eval()
        x, lengths = make_batch(52, T, B, D)
        packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
        torch.manual_seed(5000)
        # many more than 50 draws to pin the reference's predictive mean/std tightly
        n_ref = 2000
        probs = np.zeros((n_ref, B, 2), np.float32)
        with torch.no_grad():
            for s in range(n_ref):
                probs[s] = softmax(m(packed))
        save(f"g5_mcstats_{model}", sd=sd_np(m), x=x, lengths=lengths,
             mean=probs.mean(0).astype(np.float64), std=probs.std(0, ddof=1).astype(np.float64),
             n_ref=np.int64(n_ref), S=np.int64(S), dropout=np.float32(st.dropout))


# ---------------------------------------------------------------- G5b the sampled-mode gate as SURVEY 8d writes it
GATE_N, GATE_G, GATE_T, GATE_SEED = 10000, 5, 40, 521


def g5_gate(model):
    """10 000 curves, S = 50, in mini-batches of 5 curves (one weight draw / W_hh row mask is shared by a mini-batch,
    bayesian_rnn.py:492-496, variational_rnn.py:224-238, so curves of different batches are independent -- what the KS
    test on the standardised differences needs).  The reference runs 100 forwards per batch; run A = draws 0..49,
    run B = draws 50..99: per-curve mean / std (ddof 1) of both are stored, so the gate can also be evaluated
    reference-against-reference.  Inputs are regenerated in the test from the seed (tests/util.make_curves)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_curves
    D, S = 24, 50
    st = settings()
    torch.manual_seed(51)
    cls = dict(bayesian=bayesian_rnn.BayesianRNN, variational=variational_rnn.VariationalRNN)[model]
    m = build(cls, D, st)
    if model == "bayesian":
        with torch.no_grad():      # widen the posterior so the predictive std is well resolved (as g5_mcstats)
            for n, p in m.named_parameters():
                if "mu" in n:
                    p.mul_(6.0)
    m.eval()
    x, lengths = make_curves(GATE_SEED, GATE_N, GATE_T, D)
    mean = np.zeros((2, GATE_N, 2))
    std = np.zeros((2, GATE_N, 2))
    import time
    t0 = time.time()
    # resumable (an hour of CPU work): partial results in /tmp, each batch seeded by its own index
    ck = f"/tmp/g5_gate_{model}.npz"
    start = 0
    if os.path.exists(ck):
        z = np.load(ck)
        mean, std, start = z["mean"], z["std"], int(z["done"])
        print(f"  resuming {model} at curve {start}")
    with torch.no_grad():
        for b0 in range(start, GATE_N, GATE_G):
            if b0 % 250 == 0 and b0 > start:
                np.savez(ck, mean=mean, std=std, done=b0)
            torch.manual_seed(5001 + b0)
            idx = np.arange(b0, b0 + GATE_G)
            srt = idx[np.argsort(-lengths[idx], kind="stable")]
            lb = lengths[srt].astype(np.int64)
            packed = pack_padded_sequence(torch.from_numpy(x[: lb[0], srt]), torch.from_numpy(lb))
            pr = np.stack([softmax(m(packed)) for _ in range(2 * S)]).astype(np.float64)     # (100, G, 2)
            for r in range(2):
                mean[r, srt] = pr[r * S:(r + 1) * S].mean(0)
                std[r, srt] = pr[r * S:(r + 1) * S].std(0, ddof=1)
            if b0 % 500 == 0:
                print(f"  g5 gate {model}: {b0} curves, {time.time() - t0:.0f} s", flush=True)
    save(f"g5_gate_{model}", sd=sd_np(m), mean=mean.astype(np.float32), std=std.astype(np.float32),
         n=np.int64(GATE_N), group=np.int64(GATE_G), T=np.int64(GATE_T), seed=np.int64(GATE_SEED), S=np.int64(S),
         dropout=np.float32(st.dropout))


# ---------------------------------------------------------------- G10 get_evaluation_metrics / get_predictions
def _grab_nested(path, outer, inner, ns):
    """exec the function `inner` defined INSIDE the reference function `outer` at module level of `ns`: its free
    variables (the closure of the outer function) are then looked up in `ns`."""
    import ast
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == outer:
            for sub in ast.walk(node):
                if isinstance(sub, ast.FunctionDef) and sub.name == inner:
                    exec(compile(ast.Module([sub], []), path, "exec"), ns)
                    return
    raise KeyError(inner)


def g10_settings(model, feats, arr_norm):
    """The attributes get_data_batch / get_evaluation_metrics / _populate_predict_dict read from ExperimentSettings
    (utils/experiment_settings.py:200-246), for 4 filters, redshift none, global normalisation."""
    st = settings(model=model, nb_classes=2, batch_size=16, num_inference_samples=4, random_length=False,
                  random_redshift=False, redshift="none", norm="global",
                  pytorch_model_name=f"{model}_S_0_CLF_2", swag_scale=0.5, swag_no_lr_cov=False)
    st.training_features = list(feats)
    st.all_features = list(feats)
    st.idx_flux = [i for i, f in enumerate(feats) if "FLUXCAL_" in f]
    st.idx_fluxerr = [i for i, f in enumerate(feats) if "FLUXCALERR_" in f]
    st.idx_delta_time = [i for i, f in enumerate(feats) if "delta_time" in f]
    st.idx_features_to_normalize = st.idx_flux + st.idx_fluxerr + st.idx_delta_time
    st.d_feat_to_idx = {f: i for i, f in enumerate(feats)}
    st.arr_norm = arr_norm
    return st


def g10_data(n, seed, feats, arr_norm):
    """Synthetic test list in load_HDF5's tuple form (X_normed, target, SNID, X_all, X_ori): raw delta times (first 0,
    then 0.4..6 days), log-normal fluxes / errors, one filter-combination one-hot per step; X_normed is
    normalize_arr's formula on the 9 normalised columns."""
    g = np.random.default_rng(seed)
    F = 4
    data, peak = [], {}
    for i in range(n):
        L = int(g.integers(4, 31))
        X = np.zeros((L, len(feats)), np.float32)
        X[:, :F] = g.lognormal(3.0, 1.0, (L, F)) - 5.0
        X[:, F:2 * F] = g.lognormal(1.0, 0.5, (L, F))
        dt = g.uniform(0.4, 6.0, L)
        dt[0] = 0.0
        X[:, 2 * F] = dt
        X[np.arange(L), 2 * F + 1 + g.integers(0, 15, L)] = 1.0
        Xn = X.copy()
        cols = list(range(2 * F + 1))
        clipped = np.clip(X[:, cols], arr_norm[:, 0], np.inf)
        Xn[:, cols] = ((np.log(clipped - arr_norm[:, 0] + 1e-5) - arr_norm[:, 1]) / arr_norm[:, 2]).astype(np.float32)
        snid = f"{1000 + i}"
        data.append((Xn, int(g.integers(0, 2)), snid, X, X.copy()))
        t_end = float(np.cumsum(dt)[-1])
        # peak inside the curve, before its first step (every early slice empty) or far after it
        peak[snid] = [g.uniform(0, t_end), -40.0, t_end + 50.0][int(g.choice([0, 0, 0, 1, 2]))]
    return data, peak


def g10_predictions():
    """a10 / a12 / f2: the reference's own get_data_batch, eval_step, get_evaluation_metrics
    (utils/training_utils.py:374-481, 528-544, 609-707) and find_idx, construct_predict_dict, get_batch_predictions{,_MFE},
    dispatch_batch_predictions, flatten_to_dataframe, get_aggregated_dataframe and the nested _populate_predict_dict of
    get_predictions (validation/validate_rnn.py:19-40, 43-86, 112-172, 292-394), exec'd from its files (their modules
    import h5py / colorama, absent here), driving the reference's three model classes on a synthetic test list.
    Seeds are recorded: with rnn.draws = "host" the modules of supernnova_amd draw eps / masks with the same torch
    calls in the same order, so seeding alike replays every output."""
    import math
    import pandas as pd
    from sklearn import metrics
    R = "/root/reference/python/supernnova"
    feats = ([f"FLUXCAL_{f}" for f in "griz"] + [f"FLUXCALERR_{f}" for f in "griz"] + ["delta_time"] +
             ["g", "gi", "gir", "girz", "giz", "gr", "grz", "gz", "i", "ir", "irz", "iz", "r", "rz", "z"])
    arr_norm = np.array([[-10.0, 3.2, 0.9]] * 4 + [[0.0, 1.0, 0.5]] * 4 + [[0.0, 0.9, 0.8]], np.float64)
    tun = {"np": np, "torch": torch, "nn": torch.nn, "metrics": metrics}
    _grab(f"{R}/utils/training_utils.py", {"get_data_batch", "unnormalize_arr", "eval_step", "get_evaluation_metrics"}, tun)
    du = SimpleNamespace(OFFSETS_STR=["-7", "-2", "-1", "", "+1", "+2", "+30"], OFFSETS_VAL=[-7, -2, -1, 0, 1, 2, 30],
                         OOD_TYPES=["random", "reverse", "shuffle", "sin"])
    data, peak = g10_data(40, 91, feats, arr_norm)
    out = dict(features=np.array(feats), arr_norm=arr_norm, n=np.int64(len(data)),
               lengths=np.array([d[0].shape[0] for d in data], np.int64), x=np.concatenate([d[0] for d in data]),
               x_all=np.concatenate([d[3] for d in data]), target=np.array([d[1] for d in data], np.int64),
               snid=np.array([d[2] for d in data]), peak=np.array([peak[d[2]] for d in data], np.float64))
    for model, cls in (("vanilla", vanilla_rnn.VanillaRNN), ("variational", variational_rnn.VariationalRNN),
                       ("bayesian", bayesian_rnn.BayesianRNN)):
        st = g10_settings(model, feats, arr_norm)
        torch.manual_seed(61)
        rnn = build(cls, len(feats), st)
        if model == "bayesian":
            with torch.no_grad():
                for n_, p_ in rnn.named_parameters():
                    if "mu" in n_:
                        p_.mul_(6.0)
        rnn.eval()
        out[f"{model}/sd"] = sd_np(rnn)
        # ---- a10: get_evaluation_metrics, full list and a sampled subset
        np.random.seed(17)
        torch.manual_seed(1700)
        m_full = tun["get_evaluation_metrics"](st, data, rnn)
        np.random.seed(18)
        torch.manual_seed(1800)
        m_sub = tun["get_evaluation_metrics"](st, data, rnn, sample_size=20)
        for tag, m in (("full", m_full), ("sub", m_sub)):
            out[f"{model}/metrics_{tag}"] = np.array([m["AUC"], m["Acc"], m["loss"], m.get("KL", np.nan)], np.float64)
        # ---- a12 / f2: the MC and MFE passes of get_predictions
        ns = {"np": np, "torch": torch, "nn": torch.nn, "pd": pd, "math": math, "du": du,
              "tu": SimpleNamespace(get_data_batch=tun["get_data_batch"]), "tqdm": lambda it, **kw: it}
        _grab(f"{R}/validation/validate_rnn.py",
              {"find_idx", "construct_predict_dict", "flatten_to_dataframe", "get_aggregated_dataframe",
               "get_batch_predictions", "get_batch_predictions_MFE", "dispatch_batch_predictions"}, ns)
        _grab_nested(f"{R}/validation/validate_rnn.py", "get_predictions", "_populate_predict_dict", ns)
        if model == "vanilla":
            st.num_inference_samples = 1
        S = st.num_inference_samples
        num_elem = len(data)
        ns.update(settings=st, list_data_test=data,
                  list_batches=np.array_split(np.arange(num_elem), num_elem / min(num_elem, st.batch_size)),
                  df_SNinfo=pd.DataFrame({"SNID": [d[2] for d in data], "PEAKMJDNORM": [peak[d[2]] for d in data]}).set_index("SNID"))
        d_pred = ns["construct_predict_dict"](num_elem, S, st.nb_classes)
        d_mfe = ns["construct_predict_dict"](num_elem, 1, st.nb_classes, MFE=True)
        np.random.seed(19)
        torch.manual_seed(1900)
        ns["_populate_predict_dict"](rnn, d_pred, S)
        ns["_populate_predict_dict"](rnn, d_mfe, 1, _mode="MFE")
        for k, v in d_pred.items():
            out[f"{model}/pred/{k}"] = v
        for k, v in d_mfe.items():
            out[f"{model}/pred_mfe/{k}"] = v
        df = ns["flatten_to_dataframe"](d_pred)
        out[f"{model}/df_columns"] = np.array(list(df.columns))
        out[f"{model}/df_all_class1"] = df["all_class1"].values
        if model != "vanilla":
            agg = ns["get_aggregated_dataframe"](df)
            out[f"{model}/agg_columns"] = np.array(list(agg.columns))
            out[f"{model}/agg_index"] = np.array(agg.index.values)
            num = [c for c in agg.columns if c != "SNID"]
            out[f"{model}/agg_numeric_columns"] = np.array(num)
            out[f"{model}/agg_values"] = agg[num].values.astype(np.float64)
        nan_share = float(np.isnan(d_pred["PEAKMJD-7"][:, 0, 0]).mean())
        print(f"  g10 {model}: metrics full {m_full}, sub {m_sub}; PEAKMJD-7 empty for {nan_share:.0%} of curves")
    flat = {}
    for k, v in out.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
            continue
        flat[k] = np.asarray(v)
        if flat[k].dtype == object:
            flat[k] = flat[k].astype(str)
    path = os.path.join(OUT, "g10_predictions.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


# ---------------------------------------------------------------- G11 database loader
def g11_loader():
    """f3 (host logic): the reference's own load_HDF5 / fill_data_list / normalize_arr / unnormalize_arr
    (utils/training_utils.py:23-124, 127-317) and load_HDF5_SNinfo (utils/data_utils.py:427-455), exec'd from its
    files with `h5py.File` bound to a dict stand-in (tests/util.FakeH5) over a synthetic database of the datasets
    save_to_HDF5 writes.  Expected outputs only: the database is regenerated from its seed in the test."""
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synthetic_database, loader_settings, fake_h5
    R = "/root/reference/python/supernnova"
    db, feats = synthetic_database()
    quiet = SimpleNamespace(print_green=lambda *a, **k: None, print_red=lambda *a, **k: None)
    ns = {"np": np, "h5py": SimpleNamespace(File=lambda path, mode="r": fake_h5(db)), "lu": quiet,
          "tqdm": lambda it, **kw: it}
    _grab(f"{R}/utils/training_utils.py", {"normalize_arr", "unnormalize_arr", "fill_data_list", "load_HDF5"}, ns)
    dns = {"np": np, "pd": pd, "h5py": ns["h5py"]}
    _grab(f"{R}/utils/data_utils.py", {"load_HDF5_SNinfo"}, dns)
    out = {}
    for tag, kw in (("global", {}), ("zspe", dict(redshift="zspe")), ("cosmo", dict(norm="cosmo")),
                    ("quantile", dict(norm="cosmo_quantile")), ("none", dict(norm="none")), ("c7", dict(nb_classes=7))):
        st = loader_settings(feats, **kw)
        test = ns["load_HDF5"](st, test=True)
        out[f"{tag}/test_snid"] = np.array([t[2] for t in test])
        out[f"{tag}/test_target"] = np.array([t[1] for t in test], np.int64)
        out[f"{tag}/test_lengths"] = np.array([t[0].shape[0] for t in test], np.int64)
        for j, name in ((0, "x_normed"), (3, "x_all"), (4, "x_ori")):
            out[f"{tag}/test_{name}"] = np.concatenate([t[j] for t in test]).astype(np.float32)
        assert all(t[0].dtype == np.float32 for t in test)
        np.random.seed(5)
        train, val = ns["load_HDF5"](st)
        for nm, lst in (("train", train), ("val", val)):
            out[f"{tag}/{nm}_snid"] = np.array([t[2] for t in lst])
            out[f"{tag}/{nm}_target"] = np.array([t[1] for t in lst], np.int64)
            out[f"{tag}/{nm}_x_normed"] = np.concatenate([t[0] for t in lst]).astype(np.float32)
        out[f"{tag}/data_types_training"] = np.asarray(st.data_types_training).astype(str)
        print(f"  g11 {tag}: test {len(test)}, train {len(train)}, val {len(val)} curves")
    info = dns["load_HDF5_SNinfo"](loader_settings(feats))
    out["sninfo_columns"] = np.array(list(info.columns))
    out["sninfo_snid"] = info["SNID"].values.astype(str)
    out["sninfo_peak"] = info["PEAKMJDNORM"].values
    path = os.path.join(OUT, "g11_loader.npz")
    np.savez_compressed(path, **{k: (v.astype(str) if v.dtype == object else v) for k, v in out.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


# ---------------------------------------------------------------- G7 on-the-fly (broker) path
def _grab(path, names, ns):
    """exec the named top-level functions of a reference file (its module imports h5py/astropy/colorama,
    which this image lacks, so the module as a whole cannot be imported)."""
    import ast
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)


def onthefly_frames():
    """(name, DataFrame) inputs: the reference's example light curve as run_onthefly.load_lc_csv reads it
    (:41-84: sorted by MJD, SNID int -> str), and a synthetic multi-curve alert batch with the awkward cases
    (unsorted rows, repeated filters inside one 8-hour group, equal MJDs, a filter the model does not know,
    string SNIDs whose lexicographic order differs from the numeric one, a 1-row curve)."""
    import pandas as pd
    df = pd.read_csv("/root/reference/tests/onthefly_lc/example_lc.csv")
    df = df.sort_values(by=["MJD"])
    df["SNID"] = df["SNID"].astype(int).astype(str)
    frames = [("example", df[["SNID", "MJD", "FLUXCAL", "FLUXCALERR", "FLT"]].reset_index(drop=True))]
    g = np.random.default_rng(77)
    rows = []
    for snid, n in (("9", 25), ("10", 40), ("123", 7), ("7", 1), ("88", 60), ("1000", 33)):
        t = 57000.0 + np.cumsum(g.choice([0.0, 0.01, 0.1, 0.2, 0.34, 0.5, 1.0, 3.0, 7.5], size=n))
        for k in range(n):
            rows.append((snid, t[k], g.normal(50, 200), abs(g.normal(10, 5)) + 0.1,
                         g.choice(["g", "r", "i", "z", "Y"], p=[0.25, 0.25, 0.2, 0.2, 0.1])))
    rows.append(("88", 57000.0 - 5.0, -3000.0, 4.0, "g"))       # below the training minimum: clipped
    df2 = pd.DataFrame(rows, columns=["SNID", "MJD", "FLUXCAL", "FLUXCALERR", "FLT"])
    df2 = df2.iloc[g.permutation(len(df2))].reset_index(drop=True)
    frames.append(("synthetic", df2))
    return frames


def g7_onthefly():
    """validate_onthefly.classify_lcs (:123-236) on the bundled vanilla model.  format_data (:65-120),
    data_utils.compute_delta_time (:364-390), make_dataset.pivot_dataframe_single_from_df (:800-938),
    training_utils.normalize_arr (:23-86) and conf.get_norm_from_model (:722-738) are the reference's own
    code, exec'd from its files; the glue between them (:160-188 and get_data_batch's sort/pad/pack) is
    re-stated here because classify_lcs itself needs ExperimentSettings/h5py.  NOTE: with pandas >= 2
    get_dummies returns bool columns, `sel.values` becomes an object array and the reference's np.log raises
    TypeError (seen here with pandas 2.3.3 / numpy 2.2); the one-hot columns are cast to float32 -- the dtype
    the same code yields with uint8 dummies -- before `.values`."""
    import json
    from pathlib import Path
    import pandas as pd
    R = "/root/reference/python/supernnova"
    du = {"np": np, "pd": pd, "os": os}
    _grab(f"{R}/utils/data_utils.py", {"compute_delta_time", "load_fitfile"}, du)
    data_utils = SimpleNamespace(compute_delta_time=du["compute_delta_time"], load_fitfile=du["load_fitfile"])
    md = {"np": np, "pd": pd, "data_utils": data_utils}
    _grab(f"{R}/data/make_dataset.py", {"pivot_dataframe_single_from_df"}, md)
    tun = {"np": np}
    _grab(f"{R}/utils/training_utils.py", {"normalize_arr"}, tun)
    cf = {"Path": Path}
    _grab(f"{R}/conf.py", {"get_norm_from_model"}, cf)
    vo = {"np": np, "pd": pd, "data_utils": data_utils,
          "pivot_dataframe_single_from_df": md["pivot_dataframe_single_from_df"]}
    _grab(f"{R}/validation/validate_onthefly.py", {"format_data"}, vo)
    mdir = "/root/reference/tests/onthefly_model"
    mf = f"{mdir}/vanilla_S_0_CLF_2_R_none_photometry_DF_1.0_N_global_lstm_32x2_0.05_128_True_mean.pt"
    cli = json.load(open(f"{mdir}/cli_args.json"))
    cli.setdefault("sntype_var", "SNTYPE")
    cli.setdefault("additional_train_var", None)
    st = cf["get_norm_from_model"](mf, SimpleNamespace(**cli))
    st.idx_features_to_normalize = [i for i, f in enumerate(st.training_features)
                                    if f in st.training_features_to_normalize]
    st.idx_features = [i for i, f in enumerate(st.all_features) if f in st.training_features]
    rnn = build(vanilla_rnn.VanillaRNN, len(st.training_features), st)
    rnn.load_state_dict(torch.load(mf, map_location="cpu", weights_only=True))
    rnn.eval()
    out = {}
    import warnings
    for name, df in onthefly_frames():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fd = vo["format_data"](df.copy(), st)
        fd = fd.astype({c: np.float32 for c in fd.columns if fd[c].dtype == bool})
        ids, list_lcs = [], []
        for snid, sel in fd.groupby(level=0):
            X_all = sel.values
            assert X_all.dtype == np.float32
            X_clip = np.clip(X_all.copy()[:, st.idx_features_to_normalize], st.arr_norm[:, 0], np.inf)
            X_all[:, st.idx_features_to_normalize] = X_clip
            X_normed = tun["normalize_arr"](X_all.copy(), st)[:, st.idx_features]
            ids.append(snid)
            list_lcs.append(X_normed)
        lens = np.array([len(x) for x in list_lcs])
        idx_sort = np.argsort(lens)[::-1]
        T, B, D = lens.max(), len(lens), list_lcs[0].shape[1]
        X = np.zeros((T, B, D), np.float32)
        for j, k in enumerate(idx_sort):
            X[: lens[k], j] = list_lcs[k]
        packed = pack_padded_sequence(torch.from_numpy(X), lens[idx_sort].tolist())
        with torch.no_grad():
            probs = softmax(rnn(packed))[np.argsort(idx_sort)]
        assert list(fd.index.unique()) == ids
        out[name] = dict(snid=np.array(df["SNID"].values, dtype="U8"), mjd=df["MJD"].values.astype(np.float64),
                         flux=df["FLUXCAL"].values.astype(np.float64), err=df["FLUXCALERR"].values.astype(np.float64),
                         flt=np.array(df["FLT"].values, dtype="U2"),
                         fmt_index=np.array(fd.index.values, dtype="U8"), fmt_values=fd.values.astype(np.float32),
                         ids=np.array(ids, dtype="U8"), lengths=lens.astype(np.int64),
                         x_normed=np.concatenate(list_lcs, 0).astype(np.float32), probs=probs[:, None, :])
        print(f"  g7 {name}: {B} curves, rows {lens.tolist()}, probs[0] {probs[0]}")
    save("g7_onthefly", **out)


# ---------------------------------------------------------------- G9 GRU / RNN layer types
def flat_rnn_generic(sd, L, dirs):
    """Same key order as flat_vanilla (torch's _flat_weights); row counts follow the cell (3H for GRU, H for RNN)."""
    return flat_vanilla(sd, L, dirs)


def g9_layer_types():
    """VanillaRNN with layer_type gru / rnn (vanilla_rnn.py:26-32: torch.nn.GRU / torch.nn.RNN), both poolings,
    packed and dense input; the C oracle's GRU / tanh-RNN cells are checked against the reference here."""
    for cell, seed in (("gru", 31), ("rnn", 32)):
        for pool, C, D in (("mean", 2, 24), ("standard", 3, 26)):
            torch.manual_seed(seed + (pool == "standard"))
            st = settings(layer_type=cell, rnn_output_option=pool, nb_classes=C)
            m = build(vanilla_rnn.VanillaRNN, D, st)
            m.eval()
            x, lengths = make_batch(seed, 19, 37, D)
            with torch.no_grad():
                lo = m(pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths)))
                lo_dense = m(torch.from_numpy(x))
            sdn = sd_np(m)
            cfg = oracle.make_cfg(D, st.hidden_dim, st.num_layers, st.bidirectional, C, pool, cell=cell)
            w = flat_rnn_generic(sdn, 2, 2)
            check(f"{cell} {pool} packed", oracle.forward(cfg, x, lengths, w)["logits"], lo.numpy())
            check(f"{cell} {pool} dense", oracle.forward(cfg, x, np.full(37, 19), w)["logits"], lo_dense.numpy())
            save(f"g9_{cell}_{pool}", sd=sdn, x=x, lengths=lengths, logits=lo.numpy(), probs=softmax(lo),
                 logits_dense=lo_dense.numpy())


# ---------------------------------------------------------------- G8 SWAG
def g8_swag():
    """utils/swag_utils.py SwagModel around a VanillaRNN: 6 parameter snapshots (a random walk standing in for SGD
    iterates) through update_parameters, then sample(scale, cov) with a fixed seed, 3 samples each with and without
    the low-rank covariance term; logits of every sampled model and of the SWA mean.  The draws are captured by
    re-seeding and repeating the reference's calls (randn_like(var) then normal_() of K) -- verified below."""
    spec = importlib.util.spec_from_file_location("ref_swag_utils", "/root/reference/python/supernnova/utils/swag_utils.py")
    swag_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(swag_utils)
    torch.manual_seed(21)
    st, D = settings(), 24
    m = build(vanilla_rnn.VanillaRNN, D, st)
    m.eval()
    swag = swag_utils.SwagModel(m)
    snaps = []
    g = torch.Generator().manual_seed(5)
    for it in range(6):
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.02 * torch.randn(p.shape, generator=g))
        snaps.append(flat_vanilla(sd_np(m), 2, 2))
        swag.update_parameters(m)
    x, lengths = make_batch(8, 22, 19, D)
    packed = pack_padded_sequence(torch.from_numpy(x), torch.from_numpy(lengths))
    sd = {k: v.detach().cpu().numpy().copy() for k, v in swag.state_dict().items()}
    out = dict(sd=sd, snapshots=np.stack(snaps), x=x, lengths=lengths, scale=0.5)
    with torch.no_grad():
        out["logits_swa"] = swag(packed).numpy()
    for cov in (True, False):
        logits, flats = [], []
        for s in range(3):
            torch.manual_seed(3000 + s)
            sm = swag.sample(0.5, cov)
            sm.eval()
            with torch.no_grad():
                logits.append(sm(packed).numpy())
            flats.append(flat_vanilla(sd_np(sm), 2, 2))
        tag = "cov" if cov else "diag"
        out[f"logits_{tag}"] = np.stack(logits)
        out[f"flat_{tag}"] = np.stack(flats)
        o = oracle.forward(ocfg(st, D), x, lengths, flats[0])
        check(f"swag {tag} sample 0 vs oracle", o["logits"], logits[0])
    save("g8_swag", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    oracle.build()
    torch.set_num_threads(int(os.environ.get("SNN_GOLDEN_THREADS", "8")))   # tiny batches: 1 thread is fastest
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g7", "g8", "g9"]
    if "g0" in which:
        g0_init()
    if "g1" in which:
        g1_vanilla()
    if "g2" in which:
        g2_variational()
    if "g3" in which:
        g3_bayesian()
    if "g4" in which:
        g4_train()
    if "g5" in which:
        g5_mc_stats()
    for model in ("bayesian", "variational"):
        if f"g5gate_{model}" in which:
            g5_gate(model)
    if "g7" in which:
        g7_onthefly()
    if "g10" in which:
        g10_predictions()
    if "g11" in which:
        g11_loader()
    if "g12" in which:
        g12_train_cells()
    if "g13" in which:
        g13_train_elasticc()
    if "g8" in which:
        g8_swag()
    if "g9" in which:
        g9_layer_types()
