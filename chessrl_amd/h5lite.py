"""Minimal pure-Python HDF5 reader / writer -- just what Keras ``.h5`` weight files need.

The reference stores its tower as Keras HDF5 weights (``model.save_weights`` /
``load_weights``, /root/reference/src/chessrl/model.py:77-81; files ``model-<v>.h5``,
selfplay.py:33-56) and h5py is not in this image, so the subset of the HDF5 file format that
h5py's defaults produce is read and written here directly (HDF5 File Format Specification v1/v2
structures: superblock v0/v1, version-1 object headers with continuation blocks, old-style groups
= symbol-table message + v1 B-tree + local heap + symbol nodes, dataspace v1/v2, fixed-point /
IEEE-float / fixed-length-string datatypes, layout v3 contiguous / compact, attribute messages
v1-v3).  Anything else (new-style groups, chunked or filtered datasets, variable-length data)
raises ``H5Error`` -- never a silent wrong answer.  Pinned in tests/test_keras_h5.py against a
file written by the real HDF5 1.10.6 library and, the other way round, by feeding files written
here to that library's ``h5dump``.

``read(path) -> Group`` (dict-like tree of ``Group`` / ``numpy.ndarray`` with ``.attrs``),
``write(path, tree)`` with ``tree = Group(...)``.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


class Group(dict):
    """name -> Group | Dataset, plus ``attrs`` (name -> ndarray / bytes / scalar)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.attrs = {}


class Dataset(np.ndarray):
    """ndarray with ``attrs``."""

    def __new__(cls, arr, attrs=None):
        obj = np.asarray(arr).view(cls)
        obj.attrs = dict(attrs or {})
        return obj

    def __array_finalize__(self, obj):
        self.attrs = getattr(obj, "attrs", {})


# ================================================================================== reader

class _Reader(object):
    def __init__(self, buf):
        self.b = buf
        base = 0
        while self.b[base:base + 8] != SIGNATURE:               # superblock at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.b):
                raise H5Error("not an HDF5 file (no signature)")
        ver = self.b[base + 8]
        if ver > 1:
            raise H5Error("superblock version %d (new-style file) is not supported" % ver)
        self.O, self.L = self.b[base + 13], self.b[base + 14]
        if (self.O, self.L) != (8, 8):
            raise H5Error("only 8-byte offsets/lengths are supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base = self.u64(p)
        p += 4 * 8                                               # base, free-space, EOF, driver info
        self.root_header = self.u64(p + 8)                       # root symbol-table entry

    def u16(self, p):
        return struct.unpack_from("<H", self.b, p)[0]

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def u64(self, p):
        return struct.unpack_from("<Q", self.b, p)[0]

    # ---- object headers ---------------------------------------------------------------------
    def messages(self, addr):
        addr += self.base
        if self.b[addr:addr + 4] == b"OHDR":
            raise H5Error("version-2 object headers (new-style file) are not supported")
        if self.b[addr] != 1:
            raise H5Error("object header version %d at %d" % (self.b[addr], addr))
        nmsg, size = self.u16(addr + 2), self.u32(addr + 8)
        out, blocks = [], [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u16(p), self.u16(p + 2), self.b[p + 4]
                body = p + 8
                if flags & 0x02:
                    raise H5Error("shared object-header messages are not supported")
                if mtype == 0x0010:
                    blocks.append((self.base + self.u64(body), self.u64(body + 8)))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    def datatype(self, p):
        """-> (numpy dtype or None, encoded size)."""
        cls, ver = self.b[p] & 0x0F, self.b[p] >> 4
        bits = self.b[p + 1] | (self.b[p + 2] << 8) | (self.b[p + 3] << 16)
        size = self.u32(p + 4)
        if ver not in (1, 2, 3):
            raise H5Error("datatype version %d" % ver)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits & 0x08 else "u", size)), 8 + 4
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error("float of %d bytes" % size)
            return np.dtype("%sf%d" % (order, size)), 8 + 12
        if cls == 3:
            return np.dtype("S%d" % size), 8
        return None, None                                        # vlen, compound, ...: not needed

    def dataspace(self, p):
        ver, rank, flags = self.b[p], self.b[p + 1], self.b[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.b[p + 3] == 2:
                return None                                      # null dataspace
            q = p + 4
        else:
            raise H5Error("dataspace version %d" % ver)
        return tuple(self.u64(q + 8 * i) for i in range(rank))

    def attribute(self, p):
        ver = self.b[p]
        nsz, tsz, ssz = self.u16(p + 2), self.u16(p + 4), self.u16(p + 6)
        if ver == 1:
            def pad(n):
                return (n + 7) & ~7
            q = p + 8
        elif ver in (2, 3):
            if self.b[p + 1] & 0x03:
                raise H5Error("shared attribute datatype/dataspace")

            def pad(n):
                return n
            q = p + 8 + (1 if ver == 3 else 0)
        else:
            raise H5Error("attribute version %d" % ver)
        name = bytes(self.b[q:q + nsz]).split(b"\0")[0].decode("utf8")
        q += pad(nsz)
        dt, _ = self.datatype(q)
        q += pad(tsz)
        shape = self.dataspace(q)
        q += pad(ssz)
        if dt is None or shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        val = np.frombuffer(self.b, dtype=dt, count=n, offset=q).reshape(shape).copy()
        return name, (val[()] if shape == () else val)

    # ---- groups and datasets ----------------------------------------------------------------
    def heap_string(self, heap, off):
        h = self.base + heap
        if self.b[h:h + 4] != b"HEAP":
            raise H5Error("bad local heap at %d" % heap)
        data = self.base + self.u64(h + 24)
        end = data + off
        while self.b[end] != 0:
            end += 1
        return bytes(self.b[data + off:end]).decode("utf8")

    def symbols(self, btree, heap):
        """[(name, object header address)] of an old-style group."""
        out, stack = [], [btree]
        while stack:
            t = self.base + stack.pop()
            if self.b[t:t + 4] != b"TREE" or self.b[t + 4] != 0:
                raise H5Error("bad group B-tree node at %d" % t)
            level, n = self.b[t + 5], self.u16(t + 6)
            p = t + 24 + 8                                       # skip siblings and key 0
            kids = [self.u64(p + 16 * i) for i in range(n)]
            if level > 0:
                stack.extend(reversed(kids))
                continue
            for k in kids:
                s = self.base + k
                if self.b[s:s + 4] != b"SNOD":
                    raise H5Error("bad symbol node at %d" % k)
                for i in range(self.u16(s + 6)):
                    e = s + 8 + 40 * i
                    out.append((self.heap_string(heap, self.u64(e)), self.u64(e + 8)))
        return out

    def load(self, addr):
        msgs = self.messages(addr)
        attrs = {}
        stab = layout = dtype = shape = None
        for mtype, p, n in msgs:
            if mtype == 0x000C:
                k, v = self.attribute(p)
                attrs[k] = v
            elif mtype == 0x0011:
                stab = (self.u64(p), self.u64(p + 8))
            elif mtype == 0x0008:
                layout = p
            elif mtype == 0x0003:
                dtype = self.datatype(p)[0]
            elif mtype == 0x0001:
                shape = self.dataspace(p)
            elif mtype in (0x0002, 0x0006, 0x000A):
                raise H5Error("new-style group (link messages) is not supported")
            elif mtype == 0x000B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if stab is not None:
            g = Group()
            g.attrs = attrs
            for name, child in self.symbols(*stab):
                g[name] = self.load(child)
            return g
        if layout is None or dtype is None or shape is None:
            raise H5Error("object at %d is neither an old-style group nor a readable dataset" % addr)
        if self.b[layout] != 3:
            raise H5Error("data layout version %d" % self.b[layout])
        cls = self.b[layout + 1]
        n = int(np.prod(shape)) if shape else 1
        if cls == 1:
            a = self.u64(layout + 2)
            if a == UNDEF:
                arr = np.zeros(shape, dtype)
            else:
                arr = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.base + a).reshape(shape).copy()
        elif cls == 0:
            arr = np.frombuffer(self.b, dtype=dtype, count=n, offset=layout + 4).reshape(shape).copy()
        else:
            raise H5Error("chunked datasets are not supported")
        return Dataset(arr, attrs)


def read(path):
    with open(path, "rb") as f:
        buf = f.read()
    r = _Reader(memoryview(buf))
    return r.load(r.root_header)


# ================================================================================== writer

class _Writer(object):
    LEAF_K, NODE_K = 4, 16                                       # library defaults

    def __init__(self):
        self.b = bytearray(96)                                   # superblock v0 is 96 bytes

    def alloc(self, n):
        while len(self.b) % 8:
            self.b.append(0)
        a = len(self.b)
        self.b.extend(b"\0" * n)
        return a

    def put(self, a, data):
        self.b[a:a + len(data)] = data

    # ---- message encodings ------------------------------------------------------------------
    @staticmethod
    def enc_datatype(dt):
        dt = np.dtype(dt)
        if dt.kind == "f":
            if dt.byteorder == ">":
                raise H5Error("big-endian data")
            exp, man, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[dt.itemsize]
            bits = dt.itemsize * 8
            # class 1 v1; bit field: LE, pad 0, mantissa normalisation = implied msb (2 << 4),
            # sign bit location in byte 1
            return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize, 0, bits,
                               man, exp, 0, man, bias)
        if dt.kind in "iu":
            return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize, 0,
                               dt.itemsize * 8)
        if dt.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(1, dt.itemsize))     # null-padded ASCII
        raise H5Error("cannot encode dtype %s" % dt)

    @staticmethod
    def enc_dataspace(shape):
        if shape == ():
            return struct.pack("<BBBB4x", 1, 0, 0, 0)
        return (struct.pack("<BBBB4x", 1, len(shape), 1, 0) +
                b"".join(struct.pack("<Q", int(s)) for s in shape) * 2)            # dims, max dims

    def enc_attribute(self, name, value):
        if isinstance(value, str):
            value = value.encode("utf8")
        if isinstance(value, bytes):
            value = np.array(value, dtype="S%d" % max(1, len(value)))
        value = np.asarray(value)
        if value.dtype.kind == "U":
            value = np.char.encode(value, "utf8")
        if value.dtype.kind == "S" and value.dtype.itemsize == 0:
            value = value.astype("S1")
        if value.dtype.kind == "f" and value.size == 0:
            value = value.astype("S1")                           # Keras' empty weight_names
        nm = name.encode("utf8") + b"\0"
        dt, ds = self.enc_datatype(value.dtype), self.enc_dataspace(value.shape)

        def pad(x):
            return x + b"\0" * (-len(x) % 8)
        return (struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + pad(nm) + pad(dt) + pad(ds) +
                np.array(value, order="C").tobytes())

    def object_header(self, msgs):
        body = b""
        for mtype, data in msgs:
            data = data + b"\0" * (-len(data) % 8)
            body += struct.pack("<HHB3x", mtype, len(data), 0) + data
        a = self.alloc(16 + len(body))
        self.put(a, struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)
        return a

    # ---- objects ----------------------------------------------------------------------------
    def dataset(self, arr, attrs):
        arr = np.array(arr, order="C")                      # (ascontiguousarray would turn 0-d into 1-d)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.tobytes()
        if raw:
            da = self.alloc(len(raw))
            self.put(da, raw)
        else:
            da = UNDEF
        msgs = [(0x0001, self.enc_dataspace(arr.shape)), (0x0003, self.enc_datatype(arr.dtype)),
                (0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),     # fill value v2: late alloc, undefined
                (0x0008, struct.pack("<BBQQ", 3, 1, da, len(raw)))]
        msgs += [(0x000C, self.enc_attribute(k, v)) for k, v in attrs.items()]
        return self.object_header(msgs)

    def group(self, g):
        """-> (object header address, B-tree address, heap address)."""
        entries = []
        for name in sorted(g, key=lambda s: s.encode("utf8")):
            child = g[name]
            if isinstance(child, dict):
                oh, bt, hp = self.group(child)
                entries.append((name, oh, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((name, self.dataset(child, getattr(child, "attrs", {})), 0, b"\0" * 16))
        # local heap: offset 0 = empty string (key 0 of the B-tree), then the names, 8-aligned
        heap, offs = bytearray(b"\0" * 8), []
        for name, _, _, _ in entries:
            offs.append(len(heap))
            nm = name.encode("utf8") + b"\0"
            heap += nm + b"\0" * (-len(nm) % 8)
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)                        # one free block: next = 1 (none), size
        hdata = self.alloc(len(heap))
        self.put(hdata, heap)
        hp = self.alloc(32)
        self.put(hp, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, hdata))
        # symbol nodes of at most 2*LEAF_K entries, in name order
        per = 2 * self.LEAF_K
        chunks = [list(range(i, min(i + per, len(entries)))) for i in range(0, len(entries), per)]
        if len(chunks) > 2 * self.NODE_K:
            raise H5Error("group with more than %d members" % (per * 2 * self.NODE_K))
        snods = []
        for ch in chunks:
            s = self.alloc(8 + 40 * per)
            data = b"SNOD" + struct.pack("<BBH", 1, 0, len(ch))
            for i in ch:
                name, oh, cache, scratch = entries[i]
                data += struct.pack("<QQII", offs[i], oh, cache, 0) + scratch
            self.put(s, data)
            snods.append(s)
        bt = self.alloc(24 + (2 * self.NODE_K) * 16 + 8)
        data = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for ch, s in zip(chunks, snods):
            data += struct.pack("<QQ", s, offs[ch[-1]])          # child, key = last (largest) name
        self.put(bt, data)
        msgs = [(0x0011, struct.pack("<QQ", bt, hp))]
        msgs += [(0x000C, self.enc_attribute(k, v)) for k, v in getattr(g, "attrs", {}).items()]
        return self.object_header(msgs), bt, hp

    def finish(self, root):
        oh, bt, hp = root
        eof = len(self.b)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.NODE_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, oh, 1, 0) + struct.pack("<QQ", bt, hp)
        assert len(sb) == 96
        self.put(0, sb)
        return bytes(self.b)


def write(path, tree):
    w = _Writer()
    data = w.finish(w.group(tree))
    with open(path, "wb") as f:
        f.write(data)
