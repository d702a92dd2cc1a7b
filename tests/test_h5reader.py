"""f3, binary format: supernnova_amd/utils/h5reader.py (the database reader for images without h5py) against files
written by tests/h5writer.py.  PARITY UNPINNED for the format itself: reader and writer are both written from the HDF5
File Format Specification (no h5py / libhdf5 exists here to produce or check a file); what IS pinned is everything
above the bytes -- the file read back through tu.load_HDF5 gives the tuples of the reference's own load_HDF5
(fixture g11_loader).  CPU only (the vlen gather is host code of libsnn_hip.so)."""
import numpy as np
import pytest

from h5writer import VLen, write_database
from util import load_golden, synthetic_database, loader_settings


def _file_db():
    db, feats = synthetic_database()
    out = {}
    for k, (a, attrs) in db.items():
        if k == "data":
            out[k] = (VLen(list(a), base=np.float32), attrs)
        elif a.dtype == object:
            out[k] = (VLen(list(a), string=True), attrs)
        else:
            out[k] = (a, attrs)
    for f in ("FLUXCAL_g", "delta_time"):
        for j, stat in enumerate(("min", "mean", "std")):
            out[f"normalizations/{f}/{stat}"] = (np.float64(j + 0.25), {})
    for pre in ("FLUXCAL", "FLUXCALERR"):
        for j, stat in enumerate(("min", "mean", "std")):
            out[f"normalizations_global/{pre}/{stat}"] = (np.float64(-10.0 * j + 0.5), {})
    return out, db, feats


@pytest.mark.parametrize("variant", [dict(), dict(leaf_k=4), dict(new_style=True), dict(data_layout="chunked"),
                                     dict(data_layout="compact")])
def test_reader_returns_what_was_written(tmp_path, variant):
    from supernnova_amd.utils import h5reader
    out, db, feats = _file_db()
    path = str(tmp_path / "database.h5")
    write_database(path, out, **variant)
    with h5reader.File(path) as hf:
        assert sorted(hf.keys()) == sorted({k.split("/")[0] for k in out})
        assert "data" in hf and "nope" not in hf and "normalizations/FLUXCAL_g/min" in hf
        with pytest.raises(KeyError):
            hf["nope"]
        assert int(hf["data"].attrs["n_features"]) == len(feats)
        assert [s.decode() for s in hf["features"][:]] == feats
        assert [s.decode() for s in hf["SNID"][:]] == [s.decode() for s in db["SNID"][0]]
        data = hf["data"][:]
        assert data.dtype == object and len(data) == len(db["data"][0])
        for got, want in zip(data, db["data"][0]):
            assert got.dtype == np.float32 and np.array_equal(got, want)
        assert np.array_equal(hf["data"][3], db["data"][0][3])
        flat, lens = hf["data"].read_ragged()
        assert np.array_equal(flat, np.concatenate(list(db["data"][0]))) and np.array_equal(lens, [a.size for a in db["data"][0]])
        for k, (a, _) in db.items():
            if a.dtype != object:
                got = hf[k][:]
                assert got.dtype == a.dtype and np.array_equal(got, a), k
        assert float(hf["normalizations/delta_time/std"][()]) == 2.25
        assert float(hf["normalizations_global"]["FLUXCALERR/mean"][()]) == -9.5
        assert sorted(hf["normalizations"].keys()) == ["FLUXCAL_g", "delta_time"]


@pytest.mark.parametrize("variant", [dict(), dict(new_style=True)])
def test_load_HDF5_from_a_file_equals_the_reference_lists(tmp_path, variant):
    """End to end without a stand-in: database.h5 on disk -> h5reader -> tu.load_HDF5 / du.load_HDF5_SNinfo."""
    from supernnova_amd.utils import data_utils as du, training_utils as tu
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed: open_database uses it, not the native reader")
    except ImportError:
        pass
    out, db, feats = _file_db()
    write_database(str(tmp_path / "database.h5"), out, **variant)
    g = load_golden("g11_loader")
    for tag, kw in (("global", {}), ("c7", dict(nb_classes=7))):
        st = loader_settings(feats, **kw)
        st.processed_dir = str(tmp_path)
        test = tu.load_HDF5(st, test=True)
        assert [t[2] for t in test] == [str(s) for s in g[tag]["test_snid"]]
        assert np.array_equal(np.concatenate([t[0] for t in test]), g[tag]["test_x_normed"])
        assert np.array_equal(np.concatenate([t[4] for t in test]), g[tag]["test_x_ori"])
        np.random.seed(5)
        train, val = tu.load_HDF5(st)
        assert [t[2] for t in train] == [str(s) for s in g[tag]["train_snid"]]
        assert np.array_equal(np.concatenate([t[0] for t in val]), g[tag]["val_x_normed"])
    info = du.load_HDF5_SNinfo(st)
    assert list(info.columns) == [str(c) for c in g["sninfo_columns"]]
    assert np.array_equal(info["PEAKMJDNORM"].values, g["sninfo_peak"])


def test_reader_rejects_what_it_does_not_cover(tmp_path):
    from supernnova_amd.utils import h5reader
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(ValueError):
        h5reader.File(str(p))
    with pytest.raises(ValueError):
        h5reader.File(str(p), "w")
