"""GPU: the epoch loops of training/train_rnn.py around the HIP train step -- what the reference's own
tests/test_rnn.py checks (the run completes and leaves its files), plus that training actually learns a
separable synthetic problem and that get_evaluation_metrics agrees with metrics computed from a plain forward."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from test_host_logic import settings

pytestmark = pytest.mark.gpu


def _dataset(n, seed, D=24):
    g = np.random.default_rng(seed)
    data = []
    for i in range(n):
        L = int(g.integers(5, 21))
        y = int(g.integers(0, 2))
        X = g.standard_normal((L, D)).astype(np.float32)
        X[:, :4] += 1.2 * (2 * y - 1)                # class-dependent flux level
        data.append((X, y, f"id{i}"))
    return data


def _settings(tmp_path, model, **kw):
    feats = [f"f{i}" for i in range(4)]
    return settings(model=model, use_cuda=True, batch_size=64, nb_epoch=4, monitor_interval=1, random_length=True,
                    rnn_dir=str(tmp_path), pytorch_model_name=f"{model}_test", learning_rate=5e-3,
                    training_features_to_normalize=feats, arr_norm=np.arange(12, dtype=np.float64).reshape(4, 3),
                    data_types_training=np.array(["Ia", "nonIa"]), cyclic=False, cyclic_phases=[1, 2, 3], **kw)


@pytest.mark.parametrize("model", ["vanilla", "variational", "bayesian"])
def test_train_loop_learns_and_leaves_the_reference_files(tmp_path, model):
    from supernnova_amd.training import train_rnn
    from supernnova_amd.utils import training_utils as tu
    np.random.seed(0)
    torch.manual_seed(0)
    train, val = _dataset(512, 1), _dataset(128, 2)
    st = _settings(tmp_path, model)
    if model == "bayesian":      # KL / (num_batches * batch_size) dominates the loss on a 512-curve set: slower start
        st.nb_epoch = 16
    rnn = train_rnn.train(st, train, val)
    log = json.load(open(tmp_path / "training_log.json"))[st.pytorch_model_name]
    assert log["Acc"] > 0.9 and log["AUC"] > 0.95 and log["training_time"] > 0
    assert os.path.isfile(tmp_path / f"{st.pytorch_model_name}.pt") and os.path.isfile(tmp_path / "data_norm.json")
    assert json.load(open(tmp_path / "data_norm.json"))["f1"] == {"min": 3.0, "mean": 4.0, "std": 5.0}
    # the checkpoint is a plain state dict with the reference's keys, loadable into a fresh module
    sd = torch.load(tmp_path / f"{st.pytorch_model_name}.pt", weights_only=True)
    fresh = tu.get_model(st, 24).cuda()
    fresh.load_state_dict(sd)
    # get_evaluation_metrics vs metrics from one plain deterministic forward (vanilla: eval mode has no randomness)
    if model == "vanilla":
        m = tu.get_evaluation_metrics(st, val, fresh)
        st.random_length = False
        packed, _, target, rev = tu.get_data_batch(val, np.arange(len(val)), st)
        with torch.no_grad():
            pr = torch.softmax(tu.eval_step(fresh, packed), 1).cpu().numpy()
        acc = float((pr.argmax(1) == target.cpu().numpy()).mean())
        assert abs(m["Acc"] - acc) < 1e-12 and m["AUC"] > 0.95
        nll = -np.log(pr[np.arange(len(val)), target.cpu().numpy()]).mean()
        assert abs(m["loss"] - nll) < 1e-5
    elif model == "bayesian":
        assert "KL" in tu.get_evaluation_metrics(st, val, rnn)


def test_lr_range_test_and_one_cycle(tmp_path):
    from supernnova_amd.training import train_rnn
    np.random.seed(1)
    torch.manual_seed(1)
    train, val = _dataset(640, 3), _dataset(128, 4)
    st = _settings(tmp_path, "vanilla", )
    lr = train_rnn.get_lr(st, train, val)
    assert 1e-8 <= lr <= 10.0
    st.cyclic = True
    train_rnn.train(st, train, val)
    assert os.path.isfile(tmp_path / "training_log.json")


@pytest.mark.parametrize("model", ["vanilla", "bayesian"])
def test_get_predictions_writes_the_reference_pickles(tmp_path, model):
    """Train briefly, then validate_rnn.get_predictions: PRED_<model>.pickle with one row per (curve, sample) and the
    reference's columns; the aggregated pickle for the Bayesian model; accuracy of the per-SNID median prediction."""
    import pandas as pd
    from supernnova_amd.training import train_rnn
    from supernnova_amd.validation import validate_rnn as vr
    np.random.seed(3)
    torch.manual_seed(3)
    train, val, test = _dataset(512, 5), _dataset(128, 6), _dataset(96, 7)
    st = _settings(tmp_path, model, num_inference_samples=7, device="cuda", dump_dir=str(tmp_path))
    st.nb_epoch = 16 if model == "bayesian" else 4
    train_rnn.train(st, train, val)
    np.random.seed(4)
    files = vr.get_predictions(st, list_data=test)
    assert files == [str(tmp_path / f"PRED_{st.pytorch_model_name}.pickle")]
    df = pd.read_pickle(files[0])
    S = 1 if model == "vanilla" else 7
    assert len(df) == len(test) * S
    assert {"all_class0", "all_class1", "all_reverse_class0", "all_shuffle_class1", "target", "SNID"} <= set(df.columns)
    np.testing.assert_allclose(df["all_class0"] + df["all_class1"], 1.0, atol=1e-6)
    med = df.groupby("SNID")[["all_class0", "all_class1", "target"]].median()
    assert (med[["all_class0", "all_class1"]].values.argmax(1) == med["target"].values).mean() > 0.9
    agg_file = tmp_path / f"PRED_{st.pytorch_model_name}_aggregated.pickle"
    assert agg_file.exists() == (model == "bayesian")
    if model == "bayesian":
        agg = pd.read_pickle(agg_file)
        assert len(agg) == len(test) and (agg["all_class1_std"] > 0).all()


def test_train_swag_collects_statistics_and_predicts(tmp_path):
    """train_swag: statistics from swag_start_epoch on, SWAG-draw metrics once three models are averaged, the saved
    state dict loads into a fresh SwagModel, and the MC loop classifies the test set from its draws."""
    from supernnova_amd.training import train_rnn
    from supernnova_amd.utils import training_utils as tu
    from supernnova_amd.validation import validate_rnn as vr
    np.random.seed(8)
    torch.manual_seed(8)
    train, val, test = _dataset(512, 9), _dataset(128, 10), _dataset(96, 11)
    st = _settings(tmp_path, "vanilla", swag=True, swag_start_epoch=2, swag_scale=0.5, swag_no_lr_cov=False)
    st.nb_epoch = 6
    rnn, swag = train_rnn.train(st, train, val)
    assert int(swag.n_averaged) == 4 and swag.dev_0.shape[1] == 3
    assert swag.last_metrics["Acc"] > 0.85
    sd = torch.load(tmp_path / f"{st.pytorch_model_name}_swag.pt", weights_only=True)
    st.model = "swag"
    fresh = tu.get_model(st, 24)
    for i in range(len(list(fresh.module.parameters()))):
        setattr(fresh, f"dev_{i}", torch.zeros_like(sd[f"dev_{i}"]))
    fresh.load_state_dict(sd)
    fresh.cuda()
    with torch.no_grad():
        probs = vr.predict_mc(fresh, [d[0] for d in test], 10).cpu().numpy()      # (N, 10, C), one launch
    target = np.array([d[1] for d in test])
    assert (np.median(probs, 1).argmax(1) == target).mean() > 0.85 and probs.std(1).max() > 0


def test_fused_sgd_equals_torch_sgd():
    """snn_sgd_step = torch.optim.SGD(momentum 0.9, weight_decay 1e-4) (train_rnn.py:541-546) over three steps."""
    from supernnova_amd import engine as E
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(10_001, generator=g)
    grads = [torch.randn(10_001, generator=g) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.SGD([ref], lr=0.05, momentum=0.9, weight_decay=1e-4)
    p, buf = p0.cuda(), torch.zeros(10_001, device="cuda")
    for step, gr in enumerate(grads, start=1):
        ref.grad = gr.double()
        opt.step()
        E.sgd_step(p, gr.cuda(), buf, step, 0.05, 0.9, 1e-4)
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=0)


def test_train_swag_finetune_from_a_database_file(tmp_path):
    """train(settings) with swag_pretrained_path set and NO lists: the loop reads <processed_dir>/database.h5 through
    load_HDF5 (the package's reader; file written by tests/h5writer.py), loads the pre-trained state dict, runs the 30
    SGD epochs of train_swag_finetune, collects one SWAG snapshot per epoch and leaves the reference's files."""
    from h5writer import write_database
    from test_h5reader import _file_db
    from util import loader_settings
    from supernnova_amd.training import train_rnn
    from supernnova_amd.utils import training_utils as tu
    out, _, feats = _file_db()
    write_database(str(tmp_path / "database.h5"), out)
    ls = loader_settings(feats)
    st = _settings(tmp_path, "vanilla", swag_scale=0.5, swag_no_lr_cov=False)
    st.__dict__.update({k: v for k, v in ls.__dict__.items() if k not in ("arr_norm",)})
    st.arr_norm = ls.arr_norm
    st.training_features_to_normalize = ls.training_features_to_normalize
    st.processed_dir, st.batch_size, st.learning_rate, st.swag = str(tmp_path), 16, 1e-2, False
    st.random_length = False             # the synthetic database holds curves of 1-2 epochs (random prefixes need >= 3)
    np.random.seed(12)
    torch.manual_seed(12)
    pre = tu.get_model(st, len(st.training_features))
    torch.save(pre.state_dict(), tmp_path / "pretrained.pt")
    st.swag_pretrained_path = str(tmp_path / "pretrained.pt")
    rnn, swag = train_rnn.train(st)
    assert st.nb_epoch == 30 and st.swag_start_epoch == 0 and int(swag.n_averaged) == 30
    log = json.load(open(tmp_path / "training_log.json"))[st.pytorch_model_name]
    assert np.isfinite(log["loss"]) and set(swag.last_metrics) >= {"loss", "AUC", "Acc"}
    moved = max(float((a.detach().cpu() - b.detach()).abs().max()) for a, b in zip(rnn.parameters(), pre.parameters()))
    assert moved > 1e-3
    for f in (f"{st.pytorch_model_name}.pt", f"{st.pytorch_model_name}_swag.pt", "data_norm.json"):
        assert os.path.isfile(tmp_path / f), f
