"""f3, host logic: load_HDF5 / fill_data_list / load_HDF5_SNinfo of supernnova_amd against the reference's own functions
(fixture g11_loader: exec'd from utils/training_utils.py:23-317 and utils/data_utils.py:427-455 with `h5py.File` bound to
a dict stand-in over the same synthetic database).  Every tuple is bit-identical: X_normed, X_all (clipped), X_ori,
targets, SNIDs, the order of the shuffled train / validation lists under the same numpy seed, for every normalisation."""
import numpy as np
import pytest

from util import load_golden, synthetic_database, loader_settings, fake_h5


@pytest.fixture()
def database():
    from supernnova_amd.utils import data_utils as du
    db, feats = synthetic_database()
    du._opener = lambda path: fake_h5(db)
    yield db, feats
    du._opener = None


@pytest.mark.parametrize("tag,kw", [("global", {}), ("zspe", dict(redshift="zspe")), ("cosmo", dict(norm="cosmo")),
                                    ("quantile", dict(norm="cosmo_quantile")), ("none", dict(norm="none")),
                                    ("c7", dict(nb_classes=7))])
def test_load_HDF5_equals_reference(database, tag, kw):
    from supernnova_amd.utils import training_utils as tu
    db, feats = database
    g = load_golden("g11_loader")[tag]
    st = loader_settings(feats, **kw)
    test = tu.load_HDF5(st, test=True)
    assert [t[2] for t in test] == [str(s) for s in g["test_snid"]]
    assert np.array_equal([t[1] for t in test], g["test_target"])
    assert np.array_equal([t[0].shape[0] for t in test], g["test_lengths"])
    for j, name in ((0, "x_normed"), (3, "x_all"), (4, "x_ori")):
        got = np.concatenate([t[j] for t in test])
        assert got.dtype == np.float32 and np.array_equal(got, g[f"test_{name}"]), name
    # the ragged triple the GPU path consumes is the same data, no copy per curve
    assert test.rows.base is not None or test.rows.flags.owndata
    assert np.array_equal(test.rows, g["test_x_normed"]) and np.array_equal(test.lengths, g["test_lengths"])
    assert np.array_equal(test.curve_off, np.concatenate([[0], np.cumsum(g["test_lengths"])[:-1]]) * test.rows.shape[1])
    assert all(t[0].base is not None for t in test)          # views of `rows`
    np.random.seed(5)
    train, val = tu.load_HDF5(st)
    for nm, lst in (("train", train), ("val", val)):
        assert [t[2] for t in lst] == [str(s) for s in g[f"{nm}_snid"]]
        assert np.array_equal([t[1] for t in lst], g[f"{nm}_target"])
        assert np.array_equal(np.concatenate([t[0] for t in lst]), g[f"{nm}_x_normed"])
        assert all(len(t) == 3 for t in lst)
    assert list(st.data_types_training) == [str(s) for s in g["data_types_training"]]


def test_load_HDF5_SNinfo_equals_reference(database):
    from supernnova_amd.utils import data_utils as du
    _, feats = database
    g = load_golden("g11_loader")
    info = du.load_HDF5_SNinfo(loader_settings(feats))
    assert list(info.columns) == [str(c) for c in g["sninfo_columns"]]
    assert list(info["SNID"].values) == [str(s) for s in g["sninfo_snid"]]
    assert np.array_equal(info["PEAKMJDNORM"].values, g["sninfo_peak"])


def test_load_HDF5_rejects_a_database_without_the_training_features(database):
    from supernnova_amd.utils import training_utils as tu
    _, feats = database
    st = loader_settings(feats)
    st.training_features_to_normalize = st.training_features_to_normalize + ["FLUXCAL_Y"]
    with pytest.raises(ValueError):
        tu.load_HDF5(st, test=True)
