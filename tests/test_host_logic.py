"""CPU: host-side mirror of the reference interface -- module construction / state_dict parity,
host batching, plan building, error behaviour, and that libsnn_hip.so loads and exports every
symbol include/snn_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from util import ROOT, load_golden


def settings(**kw):
    d = dict(model="vanilla", layer_type="lstm", nb_classes=2, hidden_dim=32, num_layers=2, dropout=0.05,
             bidirectional=True, use_cuda=False, rnn_output_option="mean", pi=0.75, log_sigma1=-1.0,
             log_sigma2=-7.0, rho_scale_lower=4.0, rho_scale_upper=3.0, log_sigma1_output=-1.0,
             log_sigma2_output=-7.0, rho_scale_lower_output=4.0, rho_scale_upper_output=3.0,
             random_length=False, random_redshift=False, redshift="none", no_dump=True,
             learning_rate=1e-3, weight_decay=0.0)
    d.update(kw)
    return SimpleNamespace(**d)


def test_abi_exports_every_declared_symbol():
    from supernnova_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "snn_hip.h")).read()
    declared = set(re.findall(r"\b(snn_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"snn_cfg", "snn_prior", "snn_masks", "snn_status"}
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in snn_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.snn_version() >= 100


def test_sizes_and_errors_through_abi():
    from supernnova_amd import _lib, engine as E
    cfg = E.Config(24, 32, 2, True, 2, "mean")
    assert cfg.P == 40066 and cfg.F == 64          # 40 066 params at H=32, D=24, C=2
    assert cfg.PK >= cfg.P and cfg.supported()
    assert E.Config(26, 32, 2, True, 3, "standard").F == 128
    assert E.Config(78, 128, 2, True, 14, "mean").supported()      # two-pass kernel (config 5)
    assert E.Config(50, 128, 2, True, 14, "mean").supported()       # any width <= 78 (zero-padded k-steps)
    assert not E.Config(80, 128, 2, True, 14, "mean").supported()
    assert not E.Config(24, 48, 2, True, 2, "mean").supported()     # hidden sizes: 32, 64, 128
    with pytest.raises(_lib.SnnError):
        E.Config(24, 30, 2, True, 2, "mean").P     # H % 8 != 0
    with pytest.raises(_lib.SnnError):
        E.Plan(cfg, [5, 0, 3])                      # zero-length curve
    with pytest.raises(_lib.SnnError):
        E.pack_x(cfg, E.Plan(cfg, [3, 2]), torch.zeros(3, 2, 24))   # CPU tensor: no fallback


def test_plan_matches_reference_sort():
    """snn_plan_build == stable descending sort by length, tiles of 32 (cf. training_utils.py:449-453)."""
    from supernnova_amd import engine as E
    cfg = E.Config(24, 32)
    g = np.random.default_rng(0)
    lengths = g.integers(1, 41, size=1000).astype(np.int32)
    plan = E.Plan(cfg, lengths)
    n = lengths.size
    ref = np.argsort(-lengths, kind="stable")
    assert (plan.order[:n] == ref).all() and (plan.order[n:] == -1).all()
    assert (plan.len_sorted[:n] == lengths[ref]).all() and (plan.len_sorted[n:] == 0).all()
    assert (plan.tile_T == plan.len_sorted[::32]).all()
    assert (np.diff(plan.tile_off) == plan.tile_T.astype(np.int64) * 24 * 32).all()
    assert plan.max_T == lengths.max()


@pytest.mark.parametrize("model", ["vanilla", "variational", "bayesian"])
@pytest.mark.parametrize("tag,kw,D", [("mean", {}, 24), ("standard", dict(rnn_output_option="standard", nb_classes=3), 26)])
def test_init_and_state_dict_parity(model, tag, kw, D):
    """Seeded construction reproduces the reference's state_dict: names, order, shapes, values."""
    from supernnova_amd.utils import training_utils as tu
    g = load_golden(f"g0_init_{model}_{tag}")
    torch.manual_seed(7)
    rnn = tu.get_model(settings(model=model, **kw), D)
    sd = rnn.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    for k, v in sd.items():
        assert tuple(v.shape) == g["sd"][k].shape, k
        assert np.array_equal(v.numpy(), g["sd"][k]), k


def test_reference_checkpoint_loads():
    """The vanilla checkpoint the reference ships (tests/onthefly_model/*.pt, here as fixture data)."""
    from supernnova_amd.training.vanilla_rnn import VanillaRNN
    g = load_golden("g1_vanilla_mean")
    rnn = VanillaRNN(24, settings())
    missing = rnn.load_state_dict({k: torch.from_numpy(v) for k, v in g["sd"].items()})
    assert not missing.missing_keys and not missing.unexpected_keys


def test_module_errors():
    from supernnova_amd import _lib
    from supernnova_amd.utils import training_utils as tu
    with pytest.raises(ValueError):
        tu.get_model(settings(model="bayesian", layer_type="gru"), 24)
    with pytest.raises(ValueError):
        tu.get_model(settings(model="nope"), 24)
    rnn = tu.get_model(settings(model="bayesian"), 24)
    with pytest.raises(RuntimeError):
        rnn(torch.zeros(5, 3, 23))                  # wrong feature count (bayesian_rnn.py:291-296)
    with pytest.raises(_lib.SnnError):
        rnn(torch.zeros(5, 3, 24))                  # CPU input: no fallback


def _ref_get_data_batch(list_data, idxs):
    """Independent restatement of training_utils.py:449-481 for the plain case."""
    lens = [list_data[i][0].shape[0] for i in idxs]
    idx_sort = np.argsort(lens)[::-1]
    X = np.zeros((lens[idx_sort[0]], len(idxs), list_data[idxs[0]][0].shape[1]), np.float32)
    for j, s in enumerate(idx_sort):
        a = list_data[idxs[s]][0]
        X[: a.shape[0], j] = a
    return X, [list_data[idxs[s]][1] for s in idx_sort], [lens[s] for s in idx_sort], np.argsort(idx_sort)


def test_get_data_batch_contract():
    from supernnova_amd.utils import training_utils as tu
    g = np.random.default_rng(1)
    data = [(g.standard_normal((int(g.integers(3, 30)), 24)).astype(np.float32), int(g.integers(0, 2)), f"id{i}")
            for i in range(50)]
    idxs = np.arange(10, 40)
    st = settings()
    packed, X, target, rev = tu.get_data_batch(data, idxs, st)
    Xr, tr, lr, revr = _ref_get_data_batch(data, idxs)
    assert np.array_equal(X.numpy(), Xr) and target.tolist() == tr and (rev == revr).all()
    pad, lens = torch.nn.utils.rnn.pad_packed_sequence(packed)
    assert lens.tolist() == lr and np.array_equal(pad.numpy(), Xr)
    # truncation for early classification (validate_rnn.py:351-357)
    ml = [min(5, data[i][0].shape[0]) for i in idxs]
    _, X2, _, _ = tu.get_data_batch(data, idxs, st, max_lengths=ml)
    assert X2.shape[0] == 5
    # random_length consumes numpy's RNG like the reference: randint(3, L+1) per curve, in order
    st.random_length = True
    np.random.seed(3)
    _, X3, _, rev3 = tu.get_data_batch(data, idxs, st)
    np.random.seed(3)
    exp = [np.random.randint(3, data[i][0].shape[0] + 1) for i in idxs]
    assert X3.shape[0] == max(exp)
    # OOD reverse keeps lengths and reverses time
    st.random_length = False
    _, X4, _, rev4 = tu.get_data_batch(data, idxs, st, OOD="reverse")
    b = int(rev4[0])
    L0 = data[idxs[0]][0].shape[0]
    assert np.array_equal(X4[:L0, b].numpy(), data[idxs[0]][0][::-1])


def test_find_idx_and_predict_dict():
    from supernnova_amd.validation import validate_rnn as vr
    a = np.array([0.0, 1.0, 2.5, 7.0])
    assert [vr.find_idx(a, v) for v in (-1, 0.4, 0.6, 2.0, 6.9, 99)] == [0, 0, 1, 2, 3, 3]
    d = vr.construct_predict_dict(5, 3, 2)
    assert d["all"].shape == (5, 3, 2) and "PEAKMJD-7" in d and "all_sin" in d and d["target"].dtype == np.int64
    assert "PEAKMJD" not in vr.construct_predict_dict(5, 1, 2, MFE=True)


def test_plateau_loss_string_schedule_and_training_log(tmp_path):
    """Host policies of the epoch loops (utils/training_utils.py:710-769, 847-911; training/train_rnn.py:112-122)."""
    import json
    from supernnova_amd.utils import training_utils as tu
    from supernnova_amd.training import train_rnn

    opt = SimpleNamespace(param_groups=[{"lr": 1e-3}])
    p = tu.StopOnPlateau(reduce_lr_on_plateau=True)
    outs = [p.step(0.9, opt) for _ in range(10)]
    assert outs == [None] * 10 and opt.param_groups[0]["lr"] == 1e-3          # window not full yet
    assert p.step(0.9, opt) is None and abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12   # 1st reduction
    assert p.step(0.9, opt) is None                                             # 2nd
    assert p.step(0.9, opt) is True and abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-14   # 3rd: stop
    q = tu.StopOnPlateau()
    vals = [0.5 + 0.01 * i for i in range(11)]
    assert [q.step(v) for v in vals][-1] is False                               # still improving
    assert [q.step(0.6) for _ in range(10)][-1] is True                         # flat, no LR policy: stop

    s = tu.get_loss_string({"AUC": None, "Acc": 0.91234, "loss": 0.3}, {"AUC": 0.95, "Acc": 0.9, "loss": 0.25})
    assert s == "AUC/Acc/loss [T]: NA/0.912/0.3 [V]: 0.95/0.9/0.25"

    ph = [5, 10, 15]
    sched = [train_rnn.one_cycle_sched(e, 1e-4, 1e-3, ph) for e in (0, 5, 10, 15)]
    np.testing.assert_allclose(sched, [1e-4, 1e-3, 1e-4, 1e-6], rtol=1e-12)

    st = settings(rnn_dir=str(tmp_path), pytorch_model_name="m1", data_types_training=np.array(["Ia", "nonIa"]))
    tu.save_training_results(st, {"AUC": [0.8, 0.9], "Acc": [0.7, 0.75], "loss": [0.5, 0.4]}, 12.5)
    st.pytorch_model_name = "m2"
    st.nb_classes = 3
    tu.save_training_results(st, {"AUC": [None, None], "Acc": [0.5, 0.6], "loss": [0.9, 1.0]}, 1.0)
    log = json.load(open(tmp_path / "training_log.json"))
    assert log["m1"]["AUC"] == 0.9 and log["m1"]["Acc"] == 0.75 and log["m1"]["loss"] == 0.4
    assert log["m1"]["training_time"] == 12.5 and log["m2"]["AUC"] == -1 and log["m2"]["loss"] == 0.9

    with pytest.raises(FileNotFoundError):   # no data lists: the loop reads <processed_dir>/database.h5 like the reference
        train_rnn.train(settings(processed_dir=str(tmp_path / "nowhere"), rnn_dir=str(tmp_path), source_data="photometry",
                                 training_features_to_normalize=[], arr_norm=np.zeros((0, 3))))


def test_prediction_dataframes_match_the_reference_layout():
    """flatten_to_dataframe / get_aggregated_dataframe (validate_rnn.py:147-172): column names, row order, and the
    device-side aggregate() against pandas' per-SNID median / std (ddof = 1)."""
    from supernnova_amd.validation import validate_rnn as vr
    g = np.random.default_rng(0)
    N, S, C = 7, 6, 2
    d = vr.construct_predict_dict(N, S, C)
    p = g.random((N, S, 1)).astype(np.float32)
    d["all"][:] = np.concatenate([p, 1 - p], -1)
    d["target"][:] = g.integers(0, 2, N)[:, None]
    d["SNID"][:] = np.array([str(100 + i) for i in range(N)])[:, None]
    df = vr.flatten_to_dataframe(d)
    assert len(df) == N * S and {"all_class0", "all_class1", "target", "SNID", "PEAKMJD-2_class0",
                                 "all_sin_class1"} <= set(df.columns)
    assert np.array_equal(df["all_class1"].values.reshape(N, S), d["all"][..., 1])
    agg = vr.get_aggregated_dataframe(df)
    assert list(agg["SNID"]) == sorted(str(100 + i) for i in range(N)) and "all_class0_std" in agg.columns
    med, std = vr.aggregate(torch.from_numpy(d["all"]))
    order = np.argsort([str(100 + i) for i in range(N)])
    np.testing.assert_allclose(agg["all_class0_median"].values, med.numpy()[order, 0], rtol=1e-6)
    np.testing.assert_allclose(agg["all_class0_std"].values, std.numpy()[order, 0], rtol=1e-5)
    assert np.array_equal(agg["target"].values, d["target"][order, 0])


def test_unpack_packed_sequence_equals_pad_packed_sequence():
    """The modules read a PackedSequence in place (engine.PackedRows -> snn_pack_x_steps): the lengths are those of
    torch's pad_packed_sequence (vanilla_rnn.py:75-76), and row t of curve n addressed as the kernel addresses it
    (step_off[t] + column offset) is pad_packed_sequence's [t, n]: sorted and enforce_sorted=False inputs, single curve,
    single step."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    from supernnova_amd.modules import unpack_packed_sequence
    g = torch.Generator().manual_seed(0)
    D = 6
    for trial, (B, T) in enumerate([(1, 1), (1, 9), (7, 1), (33, 40), (128, 40), (5, 3), (64, 17), (31, 29), (2000, 40), (2001, 40)]):
        lens = torch.randint(1, T + 1, (B,), generator=g)
        x = torch.randn(T, B, D, generator=g)
        srt = trial % 2 == 0
        if srt:
            lens = lens.sort(descending=True).values
        pk = pack_padded_sequence(x, lens, enforce_sorted=srt)
        ref, ref_lens = pad_packed_sequence(pk)
        rows, got_lens = unpack_packed_sequence(pk)
        assert got_lens.dtype == np.int32 and np.array_equal(ref_lens.numpy(), got_lens)
        assert rows.data.dtype == torch.float32 and rows.n_curves == B and (rows.col_off is None) == srt
        flat = rows.data.reshape(-1).numpy()
        step_off = rows.step_off.numpy()
        col = np.arange(B) * D if rows.col_off is None else rows.col_off.numpy()
        got = np.zeros_like(ref.numpy())
        for n in range(B):
            for t in range(int(got_lens[n])):
                o = step_off[t] + col[n]
                got[t, n] = flat[o:o + D]
        assert np.array_equal(got, ref.numpy())
