"""CPU: the host-side C of libsnn_hip.so under AddressSanitizer + UBSan, and hostile database files through the
package's HDF5 reader.

`make -C supernnova_amd/csrc asan` compiles the three translation units without device code (snn_plan.hip,
snn_featurize.hip, snn_h5.hip) as plain C++ with -fsanitize=address,undefined and links them with tests/host_fuzz.cpp
into supernnova_amd/host_fuzz_asan (built by __graft_entry__.build(); built here when missing: host compiler only, a few
seconds).  The driver hands the library exact-size heap blocks and hostile inputs; a sanitizer report or a crash fails
the test.  The C oracle gets the same treatment through a sanitized build of its own (`make -C oracle asan`).
The reader tests feed truncated and bit-flipped copies of a valid database through h5reader + the native vlen gather of
the regular library and expect CorruptFile / SnnError / KeyError / NotImplementedError -- never another exception type,
never a crash (reference format: utils/data_utils.py:691-733)."""
import os
import subprocess

import numpy as np
import pytest

from h5writer import write_database
from test_h5reader import _file_db
from util import ROOT

FUZZ = os.path.join(ROOT, "supernnova_amd", "host_fuzz_asan")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "supernnova_amd", "csrc"), "asan"])
    return FUZZ


def _run(*args):
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([_build(), *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_plan_build_under_sanitizers():
    assert "plan ok" in _run("plan")


def test_featurizer_under_sanitizers():
    assert "featurize ok" in _run("featurize")


def _database(tmp_path, **variant):
    from supernnova_amd.utils import h5reader
    out, db, _ = _file_db()
    path = str(tmp_path / "database.h5")
    write_database(path, out, **variant)
    with h5reader.File(path) as hf:
        desc = np.ascontiguousarray(hf["data"]._descriptors()).copy()
    return path, desc, db


@pytest.mark.parametrize("variant", [dict(), dict(new_style=True)])
def test_vlen_gather_under_sanitizers(tmp_path, variant):
    path, desc, _ = _database(tmp_path, **variant)
    (tmp_path / "desc.bin").write_bytes(desc.tobytes())
    out = _run("h5", path, str(tmp_path / "desc.bin"), "1500")
    assert "h5 ok" in out
    n_ok, n_refused = (int(w) for w in out.split() if w.isdigit())
    assert n_refused > 500 and n_ok > 50          # both outcomes are exercised


def test_oracle_under_sanitizers():
    """oracle/snn_oracle.c (test infrastructure) built with -fsanitize=address,undefined and run on a small forward +
    KL in a child process with the sanitizer runtime preloaded in front of whatever this environment already preloads."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = ("import numpy as np, oracle\n"
            "oracle.use_library('libsnn_oracle_asan.so')\n"
            "g = np.random.default_rng(0)\n"
            "for (D, H, L, bi, C, pool, N, T) in [(24, 32, 2, True, 2, 'mean', 5, 7), (26, 16, 3, False, 3, 'standard', 1, 1), (78, 64, 1, True, 14, 'mean', 3, 9)]:\n"
            "    cfg = oracle.make_cfg(D, H, L, bi, C, pool)\n"
            "    P = oracle.param_count(cfg)\n"
            "    w = g.standard_normal(P).astype(np.float32) * 0.1\n"
            "    lengths = g.integers(1, T + 1, N).astype(np.int32); lengths[0] = T\n"
            "    x = g.standard_normal((T, N, D)).astype(np.float32)\n"
            "    o = oracle.forward(cfg, x, lengths, w)\n"
            "    assert np.isfinite(o['probs']).all()\n"
            "    ws, sigma = oracle.bbb_sample(w[:P - C], np.full(P - C, -3, np.float32), g.standard_normal(P - C).astype(np.float32))\n"
            "    assert np.isfinite(oracle.kl(cfg, ws, w[:P - C], sigma))\n"
            "print('oracle ok')\n")
    pre = ":".join(p for p in (libasan, os.environ.get("LD_PRELOAD", "")) if p)
    env = dict(os.environ, LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(["python", "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "oracle ok" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        (r.returncode, r.stdout[-1000:], r.stderr[-4000:])


_ALLOWED = None


def _allowed():
    global _ALLOWED
    if _ALLOWED is None:
        from supernnova_amd import _lib
        from supernnova_amd.utils import h5reader
        _ALLOWED = (h5reader.CorruptFile, _lib.SnnError, KeyError, NotImplementedError)
    return _ALLOWED


def _read_everything(path):
    """What load_HDF5 / load_HDF5_SNinfo touch: every dataset of every group, the vlen light curves through both read
    paths."""
    from supernnova_amd.utils import h5reader

    def walk(group, depth=0):
        for k in group.keys():
            node = group[k]
            if isinstance(node, h5reader.Group):
                if depth < 4:
                    walk(node, depth + 1)
            else:
                node[()]
                dict(node.attrs)

    with h5reader.File(path) as hf:
        walk(hf)
        hf["data"].read_ragged()


@pytest.mark.parametrize("variant", [dict(), dict(new_style=True), dict(data_layout="chunked")])
def test_reader_refuses_damaged_files_with_one_exception_type(tmp_path, variant):
    path, _, db = _database(tmp_path, **variant)
    _read_everything(path)                                   # the intact file reads
    good = open(path, "rb").read()
    g = np.random.default_rng(11)
    bad = str(tmp_path / "bad.h5")
    outcomes = {"ok": 0, "refused": 0}
    cases = []
    for cut in sorted(set(int(c) for c in g.integers(0, len(good), 60)) | {0, 1, 7, 8, 9, 100, len(good) - 1}):
        cases.append(good[:cut])                             # truncations
    for _ in range(260):                                     # bit flips and byte smashes, weighted towards the metadata at the front
        b = bytearray(good)
        for _k in range(int(g.integers(1, 6))):
            at = int(g.integers(0, min(len(b), 6000))) if g.random() < 0.6 else int(g.integers(0, len(b)))
            if g.random() < 0.5:
                b[at] ^= 1 << int(g.integers(0, 8))
            else:
                b[at] = int(g.integers(0, 256))
        cases.append(bytes(b))
    for _ in range(60):                                      # 8-byte fields replaced by values near 2^64 / the file size
        b = bytearray(good)
        at = int(g.integers(0, len(b) // 8)) * 8
        v = [2 ** 64 - 1, 2 ** 64 - 16, 2 ** 63, len(b), len(b) - 1, len(b) + 1, 0][int(g.integers(0, 7))]
        b[at:at + 8] = v.to_bytes(8, "little")
        cases.append(bytes(b))
    for blob in cases:
        with open(bad, "wb") as f:
            f.write(blob)
        try:
            _read_everything(bad)
            outcomes["ok"] += 1
        except _allowed():
            outcomes["refused"] += 1
        except OSError:                                      # an empty file cannot be mapped
            assert len(blob) == 0
            outcomes["refused"] += 1
    assert outcomes["refused"] > 60 and outcomes["ok"] > 20, outcomes


def test_vlen_records_without_a_heap_object_are_empty_in_both_read_paths(tmp_path):
    """a record with length > 0 and heap address 0 or all-ones (HDF5 'undefined') is an empty element for the
    native gather exactly as for the Python path -- and an address within 16 bytes of 2^64 is refused, not followed."""
    import ctypes
    from supernnova_amd import _lib
    from supernnova_amd.utils import h5reader
    path, desc, db = _database(tmp_path)
    want = list(db["data"][0])
    with h5reader.File(path) as hf:
        for addr in (0, 2 ** 64 - 1):
            d = desc.copy()
            d[2, 4:12] = np.frombuffer(int(addr).to_bytes(8, "little"), np.uint8)
            flat, lens = hf._vlen_gather(d, np.dtype("<f4"))
            assert lens[2] == 0 and lens.sum() == flat.size
            assert np.array_equal(flat, np.concatenate([w for i, w in enumerate(want) if i != 2]))
        lib = _lib.load()
        arr = np.frombuffer(hf.buf, dtype=np.uint8)
        out = np.empty(sum(w.size for w in want), np.float32)
        for addr in (2 ** 64 - 2, 2 ** 64 - 15, 2 ** 64 - 16, 2 ** 63, arr.size - 15, arr.size):
            d = desc.copy()
            d[5, 4:12] = np.frombuffer(int(addr).to_bytes(8, "little"), np.uint8)
            rc = lib.snn_h5_vlen_gather(arr.ctypes.data_as(ctypes.c_void_p), arr.size, d.ctypes.data_as(ctypes.c_void_p),
                                        d.shape[0], 4, out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
            assert rc == -1, addr
            with pytest.raises(_allowed()):
                hf._vlen_gather(d, np.dtype("<f4"))
        del arr
