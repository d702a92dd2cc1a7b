"""Hot-path subset of the reference's database access:
  load_HDF5_SNinfo   utils/data_utils.py:427-455
  OFFSETS / OOD constants :10-19
and the database opener: `open_database(path)` returns an h5py.File when h5py is importable (the reference's own
dependency) and the package's native reader otherwise (utils/h5reader.py: a reader of exactly the subset of HDF5
that save_to_HDF5, utils/data_utils.py:486-745, writes).  Both expose the mapping interface load_HDF5 uses:
hf[name][:], hf[name][i], hf[name].attrs[key], hf.keys(), `name in hf`.
"""
import numpy as np

OOD_TYPES = ["random", "reverse", "shuffle", "sin"]
OFFSETS_STR = ["-7", "-2", "-1", "", "+1", "+2", "+30"]
OFFSETS_VAL = [0 if v == "" else int(v.replace("+", "")) for v in OFFSETS_STR]

_opener = None        # tests install a stand-in here: opener(path) -> context manager yielding the mapping


def open_database(path):
    if _opener is not None:
        return _opener(path)
    try:
        import h5py
    except ImportError:
        from . import h5reader
        return h5reader.File(path)
    return h5py.File(path, "r")


def _as_str(a):
    a = np.asarray(a)
    if a.dtype.kind == "O":
        return np.array([v.decode("utf8") if isinstance(v, bytes) else str(v) for v in a.ravel()]).reshape(a.shape)
    if a.dtype.kind == "S":
        return np.char.decode(a, "utf8")
    return a.astype(str)


def load_HDF5_SNinfo(settings):
    """Per-light-curve information of the database as a DataFrame: SNID, the class variable, SALT parameters, every
    SIM_*, dataset_* and PEAK* column (utils/data_utils.py:427-455)."""
    import pandas as pd
    d = {}
    with open_database(f"{settings.processed_dir}/database.h5") as hf:
        keys = list(hf.keys())
        cols = ["SNID", settings.sntype_var, "mB", "c", "x1"]
        cols += [c for c in keys if "SIM_" in c] + [c for c in keys if "dataset_" in c] + [c for c in keys if "PEAK" in c]
        for key in cols:
            d[key] = hf[key][:]
    d["SNID"] = _as_str(d["SNID"])
    return pd.DataFrame(d)
