"""Reader of the subset of HDF5 that the reference's `save_to_HDF5` writes (utils/data_utils.py:486-745), for images
without h5py.  PARITY UNPINNED: no h5py and no HDF5 file exist in this build environment, so the parser
is written from the published HDF5 File Format Specification (version 3.0) and tested against files produced by
tests/h5writer.py, which is written from the same specification; it has never read a file libhdf5 wrote.  When h5py is
importable `utils.data_utils.open_database` uses h5py and this module is not involved.

Covered (what libhdf5 >= 1.8 writes for that call sequence with default properties, plus the "latest format" forms):
  superblock versions 0-3; object headers version 1 (with continuation blocks) and version 2 ("OHDR"/"OCHK");
  groups as symbol tables (B-tree v1 + local heap, "TREE"/"SNOD"/"HEAP") or as compact link messages;
  datasets with contiguous, compact or chunked (B-tree v1, no filters) layout;
  datatypes: fixed-point, IEEE float, fixed-length string, variable-length sequence and variable-length string
  (global heap collections, "GCOL"); attributes (message versions 1-3) of those types.
Not covered (raises NotImplementedError): dense link storage (fractal heaps), filters / compression, compound and
other datatype classes, external / virtual storage.

Interface = the part of h5py.File that tu.load_HDF5 / du.load_HDF5_SNinfo use: `with File(path) as hf`, `hf[name]`
(nested "a/b/c"), `name in hf`, `hf.keys()`, `ds[:]`, `ds[i]`, `ds.shape`, `ds.dtype`, `ds.attrs[key]`.
Variable-length float32 datasets -- the light curves, `hf["data"]` -- are gathered by the native helper
`snn_h5_vlen_gather` (libsnn_hip.so, host code): one pass for the lengths, one copy pass into a single (sum(len),)
float32 buffer; `ds.read_ragged()` returns (flat values, lengths) without building one array per curve.
"""
import mmap
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class CorruptFile(ValueError):
    """The file's own fields contradict each other or point outside the file (truncated or damaged database.h5)."""


def _guard(fn):
    """A damaged file shows up in the parser as struct.error / IndexError / OverflowError / RecursionError / ... from
    wherever a field was followed; the public entry points turn all of them into CorruptFile (a ValueError), so callers
    handle ONE exception type.  KeyError (a name that does not exist), NotImplementedError (a valid feature outside the
    covered subset) and the native gather's SnnError pass through."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            return fn(*a, **kw)
        except (KeyError, NotImplementedError, CorruptFile):
            raise
        except (struct.error, IndexError, OverflowError, RecursionError, MemoryError, UnicodeDecodeError, TypeError,
                ValueError, ZeroDivisionError) as e:
            raise CorruptFile(f"damaged or truncated HDF5 file: {type(e).__name__}: {e}") from e
    return wrapped


_MAX_BLOCKS = 4096          # object-header continuation blocks followed per object (a cycle would never end)


class _Type:
    """Parsed datatype message."""

    def __init__(self, buf, off):
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
        self.cls, self.version, self.size = cv & 0x0F, cv >> 4, size
        self.bits = b0 | (b1 << 8) | (b2 << 16)
        self.base = None
        p = off + 8
        if self.cls == 0:                                   # fixed point
            self.np = np.dtype(("<" if not (b0 & 1) else ">") + ("i" if b0 & 8 else "u") + str(size))
        elif self.cls == 1:                                 # floating point (IEEE assumed)
            self.np = np.dtype(("<" if not (b0 & 1) else ">") + "f" + str(size))
        elif self.cls == 3:                                 # fixed-length string
            self.np = np.dtype(f"S{size}")
        elif self.cls == 9:                                 # variable length: 16-byte (length, heap address, index) records
            self.base = _Type(buf, p)
            self.vlen_string = (b0 & 0x0F) == 1
            self.np = np.dtype(object)
        else:
            raise NotImplementedError(f"HDF5 datatype class {self.cls} is outside the subset this reader covers")


def _dataspace(buf, off):
    ver, rank, flags = struct.unpack_from("<BBB", buf, off)
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if buf[off + 3] == 2:                               # null dataspace
            return None
        p = off + 4
    else:
        raise NotImplementedError(f"dataspace message version {ver}")
    return tuple(struct.unpack_from(f"<{rank}Q", buf, p)) if rank else ()


class _Object:
    """Header messages of one object: list of (type, flags, offset, size)."""

    def __init__(self, f, addr):
        self.f, self.addr, self.msgs = f, addr, []
        buf = f.buf
        if buf[addr:addr + 4] == b"OHDR":
            self._v2(addr)
        else:
            self._v1(addr)

    def _v1(self, addr):
        buf = self.f.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", buf, addr)
        if ver != 1:
            raise ValueError(f"object header at {addr}: version {ver}")
        blocks, seen = [(addr + 16, hsize)], 0
        while blocks and len(self.msgs) < nmsg:
            seen += 1
            if seen > _MAX_BLOCKS:
                raise CorruptFile(f"object header at {addr}: continuation blocks do not end")
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(self.msgs) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, p)
                body = p + 8
                if mtype == 0x10:
                    blocks.append(struct.unpack_from("<QQ", buf, body))
                self.msgs.append((mtype, mflags, body, msize))
                p = body + msize

    def _v2(self, addr):
        buf = self.f.buf
        ver, flags = struct.unpack_from("<BB", buf, addr + 4)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size0 = int.from_bytes(buf[p:p + w], "little")
        p += w
        blocks, seen = [(p, size0)], 0
        track = bool(flags & 0x04)
        while blocks:
            seen += 1
            if seen > _MAX_BLOCKS:
                raise CorruptFile(f"object header at {addr}: continuation blocks do not end")
            p, n = blocks.pop(0)
            end = p + n
            while p + 4 + (2 if track else 0) <= end:
                mtype, msize, mflags = struct.unpack_from("<BHB", buf, p)
                body = p + 4 + (2 if track else 0)
                if body + msize > end:
                    break
                if mtype == 0x10:
                    a, ln = struct.unpack_from("<QQ", buf, body)
                    blocks.append((a + 4, ln - 8))           # skip "OCHK", drop the checksum
                if mtype != 0 or msize:
                    self.msgs.append((mtype, mflags, body, msize))
                p = body + msize

    def find(self, mtype):
        return [(o, n) for t, _, o, n in self.msgs if t == mtype]


class AttributeManager(dict):
    pass


class Dataset:
    def __init__(self, f, obj, name):
        self.f, self.name = f, name
        buf = f.buf
        (o, _), = obj.find(0x03)
        self.type = _Type(buf, o)
        (o, _), = obj.find(0x01)
        self.shape = _dataspace(buf, o)
        self.dtype = self.type.np
        (o, n), = obj.find(0x08)
        ver, cls = struct.unpack_from("<BB", buf, o)
        if ver != 3:
            raise NotImplementedError(f"data layout message version {ver}")
        self.layout = cls
        if cls == 0:
            size, = struct.unpack_from("<H", buf, o + 2)
            self.where = (o + 4, size)
        elif cls == 1:
            self.where = struct.unpack_from("<QQ", buf, o + 2)
        elif cls == 2:
            nd, = struct.unpack_from("<B", buf, o + 2)
            self.chunk_btree, = struct.unpack_from("<Q", buf, o + 3)
            self.chunk_dims = struct.unpack_from(f"<{nd}I", buf, o + 11)       # last entry = element size
        else:
            raise NotImplementedError(f"data layout class {cls}")
        if obj.find(0x0B):
            raise NotImplementedError(f"{name}: filtered (compressed) datasets are outside the subset this reader covers")
        self.attrs = AttributeManager()
        for o, n in obj.find(0x0C):
            k, v = f._attribute(o)
            self.attrs[k] = v

    def __len__(self):
        return self.shape[0] if self.shape else 1

    # -- raw bytes of the whole dataset
    def _raw(self):
        buf = self.f.buf
        nbytes = int(np.prod([int(d) for d in self.shape], dtype=object)) * self.type.size if self.shape is not None else 0
        # contiguous / compact data lies inside the file; chunked data may have unwritten chunks, but a shape that asks
        # for far more than the file could ever hold is a damaged dataspace message, not a dataset
        if nbytes > (len(buf) if self.layout in (0, 1) else 64 * len(buf) + (1 << 20)):
            raise CorruptFile(f"{self.name}: shape {self.shape} x {self.type.size} bytes exceeds what the file can hold")
        if self.layout in (0, 1):
            addr, size = self.where
            if addr == UNDEF or nbytes == 0:                 # never written: fill value 0
                return bytes(nbytes)
            if addr + nbytes > len(buf):
                raise CorruptFile(f"{self.name}: data at {addr} + {nbytes} bytes runs past the end of the file")
            return buf[addr:addr + nbytes]
        out = bytearray(nbytes)
        self._chunks(self.chunk_btree, out)
        return bytes(out)

    def _chunks(self, addr, out):
        buf = self.f.buf
        if addr == UNDEF:
            return
        if buf[addr:addr + 4] != b"TREE":
            raise ValueError("chunk index: B-tree signature missing")
        ntype, level, used = struct.unpack_from("<BBH", buf, addr + 4)
        nd = len(self.chunk_dims)
        p = addr + 24
        key = 8 + 8 * nd
        esz = self.chunk_dims[-1]
        cdims, shape = self.chunk_dims[:-1], self.shape
        for _ in range(used):
            csize, _mask = struct.unpack_from("<II", buf, p)
            offs = struct.unpack_from(f"<{nd}Q", buf, p + 8)[:-1]
            child, = struct.unpack_from("<Q", buf, p + key)
            p += key + 8
            if level > 0:
                self._chunks(child, out)
                continue
            chunk = np.frombuffer(buf[child:child + csize], dtype=np.uint8).reshape(cdims + (esz,))
            full = np.frombuffer(out, dtype=np.uint8).reshape(tuple(shape) + (esz,))
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            full[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

    def _descriptors(self):
        raw = np.frombuffer(self._raw(), dtype=np.uint8)
        n = raw.size // 16
        return raw.reshape(n, 16)

    @_guard
    def read_ragged(self):
        """Variable-length numeric dataset -> (flat values, lengths int64), one buffer for all elements."""
        if self.type.cls != 9 or self.type.vlen_string:
            raise TypeError(f"{self.name} is not a variable-length numeric dataset")
        desc = np.ascontiguousarray(self._descriptors())
        return self.f._vlen_gather(desc, self.type.base.np)

    def _read(self):
        if self.shape is None:
            return None
        t = self.type
        if t.cls != 9:
            return np.frombuffer(self._raw(), dtype=t.np).reshape(self.shape).copy()
        if t.vlen_string:
            desc = self._descriptors()
            out = np.empty(desc.shape[0], dtype=object)
            for i, d in enumerate(desc):
                out[i] = self.f._heap_object(d)
            return out.reshape(self.shape)
        flat, lens = self.read_ragged()
        out = np.empty(lens.size, dtype=object)
        off = np.concatenate([[0], np.cumsum(lens)])
        for i in range(lens.size):
            out[i] = flat[off[i]:off[i + 1]]
        return out.reshape(self.shape)

    @_guard
    def __getitem__(self, idx):
        # one parse per Dataset object (hf["data"][i] in a loop would otherwise re-read and re-gather the whole dataset
        # per index); callers get copies of what they index, as from h5py, except for the object arrays' elements
        a = self.__dict__.get("_read_cache")
        if a is None:
            a = self._read()
            self.__dict__["_read_cache"] = a
        if isinstance(a, np.ndarray) and a.dtype != object and not (isinstance(idx, tuple) and idx == ()):
            out = a[idx]
            return out.copy() if isinstance(out, np.ndarray) else out
        if isinstance(a, np.ndarray) and a.dtype == object and isinstance(idx, slice):
            out = a[idx].copy()
            for i in range(out.size):
                if isinstance(out.flat[i], np.ndarray):
                    out.flat[i] = out.flat[i].copy()      # the reference clips the rows it reads in place
            return out
        if isinstance(idx, tuple) and idx == ():
            return a[()] if a.shape == () else a
        return a[idx]


class Group:
    def __init__(self, f, obj, name):
        self.f, self.name = f, name
        self.links = {}
        buf = f.buf
        st = obj.find(0x11)
        if st:
            btree, heap = struct.unpack_from("<QQ", buf, st[0][0])
            if buf[heap:heap + 4] != b"HEAP":
                raise ValueError("local heap signature missing")
            self._heap_data, = struct.unpack_from("<Q", buf, heap + 24)
            self._walk(btree)
        for o, n in obj.find(0x06):                          # compact link messages (new-style groups)
            ver, flags = struct.unpack_from("<BB", buf, o)
            p = o + 2
            ltype = 0
            if flags & 0x08:
                ltype = buf[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            w = 1 << (flags & 3)
            ln = int.from_bytes(buf[p:p + w], "little")
            p += w
            lname = bytes(buf[p:p + ln]).decode("utf8")
            p += ln
            if ltype == 0:
                self.links[lname], = struct.unpack_from("<Q", buf, p)
        if obj.find(0x02) and not self.links and not st:
            info = obj.find(0x02)[0][0]
            flags = buf[info + 1]
            p = info + 2 + (8 if flags & 1 else 0)
            fheap, = struct.unpack_from("<Q", buf, p)
            if fheap != UNDEF:
                raise NotImplementedError(f"group {name}: dense link storage (fractal heap) is outside the subset this reader covers")

    def _walk(self, addr):
        buf = self.f.buf
        if addr == UNDEF:
            return
        sig = bytes(buf[addr:addr + 4])
        if sig == b"TREE":
            _, level, used = struct.unpack_from("<BBH", buf, addr + 4)
            p = addr + 24 + 8                                # skip key 0
            for _ in range(used):
                child, = struct.unpack_from("<Q", buf, p)
                self._walk(child)
                p += 16
        elif sig == b"SNOD":
            n, = struct.unpack_from("<H", buf, addr + 6)
            p = addr + 8
            for _ in range(n):
                name_off, ohdr = struct.unpack_from("<QQ", buf, p)
                s = self._heap_data + name_off
                e = self.f.buf.find(b"\x00", s)
                self.links[bytes(buf[s:e]).decode("utf8")] = ohdr
                p += 40
        else:
            raise ValueError(f"group B-tree: unexpected signature {sig!r}")

    def keys(self):
        return list(self.links.keys())

    def __iter__(self):
        return iter(self.links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    @_guard
    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split("/") if p]:
            if not isinstance(node, Group) or part not in node.links:
                raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")
            node = node.f._open(node.links[part], part)
        return node


class File(Group):
    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("supernnova_amd.utils.h5reader opens files read-only")
        self._fh = open(path, "rb")
        try:
            self._parse(path)
        except BaseException:
            self.close()
            raise

    @_guard
    def _parse(self, path):
        self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self._cache, self._gcol = {}, {}
        base = 0
        while self.buf[base:base + 8] != b"\x89HDF\r\n\x1a\n":
            base = 512 if base == 0 else base * 2
            if base >= len(self.buf):
                raise ValueError(f"{path}: not an HDF5 file (no superblock signature)")
        ver = self.buf[base + 8]
        if ver in (0, 1):
            so, sl = self.buf[base + 13], self.buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            if (so, sl) != (8, 8):
                raise NotImplementedError("only 8-byte offsets and lengths")
            root_ohdr, = struct.unpack_from("<Q", self.buf, p + 32 + 8)       # root symbol table entry: header address
        elif ver in (2, 3):
            if self.buf[base + 9] != 8:
                raise NotImplementedError("only 8-byte offsets and lengths")
            root_ohdr, = struct.unpack_from("<Q", self.buf, base + 12 + 24)
        else:
            raise NotImplementedError(f"superblock version {ver}")
        Group.__init__(self, self, _Object(self, root_ohdr), "/")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def close(self):
        buf, self.buf = getattr(self, "buf", None), None
        if buf is not None:
            try:
                buf.close()
            except BufferError:       # a numpy view of the mapping is still alive (it keeps the pages; the file closes below)
                pass
        self._fh.close()

    def _open(self, addr, name):
        node = self._cache.get(addr)
        if node is None:
            obj = _Object(self, addr)
            node = Dataset(self, obj, name) if obj.find(0x08) else Group(self, obj, name)
            self._cache[addr] = node
        return node

    def _attribute(self, o):
        buf = self.buf
        ver, _, nsz, tsz, ssz = struct.unpack_from("<BBHHH", buf, o)
        p = o + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = bytes(buf[p:p + nsz]).split(b"\x00")[0].decode("utf8")
        p += pad(nsz)
        t = _Type(buf, p)
        p += pad(tsz)
        shape = _dataspace(buf, p)
        p += pad(ssz)
        if t.cls == 9:
            n = int(np.prod(shape, dtype=np.int64)) if shape else 1
            vals = [self._heap_object(np.frombuffer(buf[p + 16 * i:p + 16 * i + 16], dtype=np.uint8)) for i in range(n)]
            return name, (vals[0] if not shape else np.array(vals, dtype=object).reshape(shape))
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        a = np.frombuffer(buf[p:p + n * t.size], dtype=t.np).copy()
        return name, (a[0] if not shape else a.reshape(shape))

    # -- global heap
    def _collection(self, addr):
        tab = self._gcol.get(addr)
        if tab is None:
            buf = self.buf
            if addr > len(buf) - 16 or buf[addr:addr + 4] != b"GCOL":
                raise ValueError(f"global heap collection at {addr}: signature missing")
            size, = struct.unpack_from("<Q", buf, addr + 8)
            tab, p = {}, addr + 16
            while p + 16 <= addr + size:
                idx, _, _, osz = struct.unpack_from("<HHIQ", buf, p)
                if idx == 0:
                    break
                tab[idx] = (p + 16, osz)
                p += 16 + ((osz + 7) & ~7)
            self._gcol[addr] = tab
        return tab

    def _heap_object(self, d, elem_size=1):
        """The `ln` elements a 16-byte vlen record {length, collection address, object index} points to.  The heap
        object may be longer than ln * elem_size (the heap pads objects to 8 bytes): only the record's length counts."""
        ln, addr, idx = struct.unpack("<IQI", bytes(d))
        if ln == 0 or addr in (0, UNDEF):
            return b""
        try:
            off, size = self._collection(addr)[idx]
        except KeyError:
            raise ValueError(f"vlen element points to heap object {idx}, which collection {addr} does not hold") from None
        if off + size > len(self.buf):
            raise ValueError(f"heap object {idx} of collection {addr} runs past the end of the file")
        return bytes(self.buf[off:off + min(ln * elem_size, size)])

    def _vlen_gather(self, desc, base):
        """(n, 16) descriptor records -> (flat values of dtype `base`, lengths).  float32 goes through the native
        helper; other base types through the Python path."""
        lens = desc[:, :4].copy().view("<u4").ravel().astype(np.int64)
        addrs = desc[:, 4:12].copy().view("<u8").ravel()
        lens[(addrs == 0) | (addrs == UNDEF)] = 0          # "no heap object": an empty element whatever its length field says
        total = int(lens.sum())
        if total * np.dtype(base).itemsize > len(self.buf):
            raise CorruptFile(f"variable-length dataset claims {total} values, more than the file holds")
        flat = np.empty(total, dtype=base)
        if base == np.dtype("<f4") and total:
            try:
                from .. import _lib
                import ctypes
                lib = _lib.load()
                arr = np.frombuffer(self.buf, dtype=np.uint8)
                rc = lib.snn_h5_vlen_gather(arr.ctypes.data_as(ctypes.c_void_p), arr.size,
                                            desc.ctypes.data_as(ctypes.c_void_p), desc.shape[0], 4,
                                            flat.ctypes.data_as(ctypes.c_void_p), flat.nbytes)
                del arr
                _lib.check(rc, "snn_h5_vlen_gather")
                return flat, lens
            except (ImportError, OSError, AttributeError):
                # no usable libsnn_hip.so on this host (not built, libamdhip64 unresolvable on a CPU-only machine, a stale
                # build without the symbol): the gather below does the same in Python.  A malformed FILE is not this
                # case: the native gather reports it as SnnError, which propagates.
                pass
        off = 0
        isz = np.dtype(base).itemsize
        for d, n in zip(desc, lens):
            if n:
                b = self._heap_object(d, isz)
                if len(b) != n * isz:
                    raise ValueError(f"vlen element of {n} values points to a heap object of {len(b)} bytes")
                flat[off:off + n] = np.frombuffer(b, dtype=base, count=n)
                off += n
        return flat, lens
