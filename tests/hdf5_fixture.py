"""Test fixture writer: MATLAB 7.3 (HDF5) files put together byte by byte from the HDF5 File Format Specification -- the layout
``save -v7.3`` produces (512-byte user block, version-0 superblock, old-style groups, version-1 object headers, contiguous / compact /
chunked + deflate (+ shuffle) datasets, object references into ``/#refs#``, the ``MATLAB_class`` attribute).  No HDF5 library exists
in this image, so ``mri_super_resolution_amd/mat73io.py`` (the reader) and this writer are two restatements of the same document;
neither is pinned against a MATLAB-written file (DESIGN.md says so).  Test infrastructure only."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
BASE = 512


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * ((-len(b)) % 8)


class Writer:
    stored_base = BASE          # the superblock's "base address" field; 0 = the other form libhdf5 accepts behind a user block

    def __init__(self):
        head = b"MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Mon Oct  5 00:00:00 2026 HDF5 schema 1.00 ."
        self.buf = bytearray(head.ljust(116) + b"\0" * 8 + struct.pack("<H", 0x0200) + b"IM")
        self.buf += b"\0" * (BASE - len(self.buf))
        self.buf += b"\0" * 96                                   # superblock + root symbol table entry, filled in by finish()
        self.refs = {}                                           # name in /#refs# -> header address

    # ---- allocation (addresses relative to the base address) ----
    def put(self, data: bytes, align: int = 8) -> int:
        self.buf += b"\0" * ((-len(self.buf)) % align)
        rel = len(self.buf) - BASE
        self.buf += data
        return rel

    # ---- messages ----
    @staticmethod
    def msg(mtype: int, body: bytes) -> bytes:
        body = _pad8(body)
        return struct.pack("<HHB3x", mtype, len(body), 0) + body

    @staticmethod
    def dataspace(dims, with_max=False) -> bytes:
        b = struct.pack("<BBB5x", 1, len(dims), 1 if with_max else 0) + struct.pack(f"<{len(dims)}Q", *dims)
        if with_max:
            b += struct.pack(f"<{len(dims)}Q", *dims)
        return b

    @staticmethod
    def datatype(dt) -> bytes:
        if dt == "ref":
            return struct.pack("<B3BI", 0x17, 0, 0, 0, 8)
        if isinstance(dt, tuple):                                # ("str", n)
            return struct.pack("<B3BI", 0x13, 0, 0, 0, dt[1])
        dt = np.dtype(dt)
        if dt.kind == "f":
            exp, man = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
            bits0 = 0x20                                          # little endian, implied-1 mantissa normalisation
            return struct.pack("<B3BI", 0x11, bits0, dt.itemsize * 8 - 1, 0, dt.itemsize) + \
                struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man, exp, 0, man, (1 << (exp - 1)) - 1)
        signed = 0x08 if dt.kind == "i" else 0
        return struct.pack("<B3BI", 0x10, signed, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)

    def attribute(self, name: str, dt, dims, data: bytes) -> bytes:
        nm = name.encode() + b"\0"
        t, s = self.datatype(dt), self.dataspace(dims)
        return struct.pack("<BxHHH", 1, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + data

    def class_attr(self, klass: str) -> bytes:
        return self.msg(0x000C, self.attribute("MATLAB_class", ("str", len(klass)), (), klass.encode()))

    def header(self, messages, split=False) -> int:
        """Version-1 object header; split=True moves the last message into a continuation block (as HDF5 does when a header grows)."""
        if split and len(messages) > 1:
            tail = messages[-1]
            tail_at = self.put(tail)
            cont = self.msg(0x0010, struct.pack("<QQ", tail_at, len(tail)))
            body = b"".join(messages[:-1]) + cont
            n = len(messages) + 1
        else:
            body, n = b"".join(messages), len(messages)
        return self.put(struct.pack("<BxHII4x", 1, n, 1, len(body)) + body)

    # ---- datasets (arr in MATLAB's shape; stored with reversed dimensions) ----
    def dataset(self, arr, klass=None, layout="contiguous", chunk=None, shuffle=False, extra=(), two_level=False, split=False,
                with_max=False) -> int:
        arr = np.asarray(arr)
        is_ref = arr.dtype == np.dtype("<u8") and klass == "cell"
        stored = np.ascontiguousarray(arr.T)                    # HDF5 sees the reversed shape, C order
        dims = stored.shape
        raw = stored.tobytes()
        es = stored.dtype.itemsize
        msgs = [self.msg(0x0001, self.dataspace(dims, with_max)), self.msg(0x0003, self.datatype("ref" if is_ref else stored.dtype))]
        if layout == "compact":
            msgs.append(self.msg(0x0008, struct.pack("<BBH", 3, 0, len(raw)) + raw))
        elif layout == "contiguous":
            at = self.put(raw) if raw else UNDEF
            msgs.append(self.msg(0x0008, struct.pack("<BBQQ", 3, 1, at, len(raw))))
        else:
            chunk = tuple(chunk)
            filt = struct.pack("<BB6x", 1, 2 if shuffle else 1)
            if shuffle:
                filt += struct.pack("<HHHH", 2, 8, 1, 1) + b"shuffle\0" + struct.pack("<II", es, 0)
            filt += struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\0" + struct.pack("<II", 3, 0)
            msgs.append(self.msg(0x000B, filt))
            keys = []
            for idx in np.ndindex(*[-(-d // c) for d, c in zip(dims, chunk)]):
                off = tuple(i * c for i, c in zip(idx, chunk))
                block = np.zeros(chunk, stored.dtype)
                src = tuple(slice(o, min(o + c, d)) for o, c, d in zip(off, chunk, dims))
                block[tuple(slice(0, s.stop - s.start) for s in src)] = stored[src]
                data = block.tobytes()
                if shuffle:
                    data = np.frombuffer(data, np.uint8).reshape(-1, es).T.tobytes()
                data = zlib.compress(data, 3)
                keys.append((len(data), off, self.put(data)))
            ndim = len(dims) + 1

            def node(level, entries, last_off):
                body = struct.pack("<4sBBHQQ", b"TREE", 1, level, len(entries), UNDEF, UNDEF)
                for nbytes, off, child in entries:
                    body += struct.pack("<II", nbytes, 0) + struct.pack(f"<{ndim}Q", *off, 0) + struct.pack("<Q", child)
                body += struct.pack("<II", 0, 0) + struct.pack(f"<{ndim}Q", *last_off, 0)
                return self.put(body)

            end = tuple(dims)
            if two_level and len(keys) > 2:
                half = len(keys) // 2
                kids = [(keys[0][0], keys[0][1], node(0, keys[:half], keys[half][1])),
                        (keys[half][0], keys[half][1], node(0, keys[half:], end))]
                root = node(1, kids, end)
            else:
                root = node(0, keys, end)
            msgs.append(self.msg(0x0008, struct.pack("<BBBQ", 3, 2, ndim, root) + struct.pack(f"<{ndim}I", *chunk, es)))
        if klass:
            msgs.append(self.class_attr(klass))
        msgs += list(extra)
        return self.header(msgs, split=split)

    def value(self, v) -> int:
        """Header address of a MATLAB value: numeric array, str (char), bool array (logical), list / object array (cell), dict (struct)."""
        if isinstance(v, dict):
            return self.group({k: self.value(x) for k, x in v.items()}, klass="struct")
        if isinstance(v, str):
            codes = np.array([[ord(c) for c in v]], dtype="<u2")
            return self.dataset(codes, "char")
        if isinstance(v, (list, tuple)) or (isinstance(v, np.ndarray) and v.dtype == object):
            cells = np.empty((1, len(v)), dtype=object) if not isinstance(v, np.ndarray) else v
            if not isinstance(v, np.ndarray):
                for i, x in enumerate(v):
                    cells[0, i] = x
            refs = np.zeros(cells.shape, dtype="<u8")
            for idx in np.ndindex(*cells.shape):
                addr = self.value(cells[idx])
                self.refs[f"r{len(self.refs)}"] = addr
                refs[idx] = addr
            return self.dataset(refs, "cell")
        a = np.asarray(v)
        if a.ndim < 2:
            a = a.reshape(1, -1)
        if a.size == 0:
            dims = np.asarray(a.shape, dtype="<u8").reshape(1, -1)
            empty = self.msg(0x000C, self.attribute("MATLAB_empty", "<u1", (), b"\x01"))
            return self.dataset(dims, {"float64": "double", "float32": "single"}.get(a.dtype.name, a.dtype.name), extra=(empty,))
        if a.dtype == np.bool_:
            return self.dataset(a.astype("<u1"), "logical")
        klass = {"float64": "double", "float32": "single"}.get(a.dtype.name, a.dtype.name)
        return self.dataset(a.astype(a.dtype.newbyteorder("<")), klass)

    # ---- groups ----
    def group(self, members: dict, klass=None, leaf_k=4):
        names = sorted(members)
        heap_data = bytearray(b"\0" * 8)
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\0")
        seg = self.put(bytes(heap_data))
        heap = self.put(struct.pack("<4sB3xQQQ", b"HEAP", 0, len(heap_data), UNDEF, seg))
        snods, keys = [], [0]
        for i in range(0, max(len(names), 1), 2 * leaf_k):
            part = names[i:i + 2 * leaf_k]
            body = struct.pack("<4sBxH", b"SNOD", 1, len(part))
            for n in part:
                body += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
            body += b"\0" * (40 * (2 * leaf_k - len(part)))
            snods.append(self.put(body))
            keys.append(offs[part[-1]] if part else 0)
        tree = struct.pack("<4sBBHQQ", b"TREE", 0, 0, len(snods), UNDEF, UNDEF) + struct.pack("<Q", keys[0])
        for child, key in zip(snods, keys[1:]):
            tree += struct.pack("<QQ", child, key)
        btree = self.put(tree)
        msgs = [self.msg(0x0011, struct.pack("<QQ", btree, heap))]
        if klass:
            msgs.append(self.class_attr(klass))
        hdr = self.header(msgs)
        self._last_group = (btree, heap)
        return hdr

    def finish(self, variables: dict) -> bytes:
        members = {k: (v if isinstance(v, int) else self.value(v)) for k, v in variables.items()}
        if self.refs:
            members["#refs#"] = self.group(dict(self.refs))
        root = self.group(members)
        btree, heap = self._last_group
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, 4, 16, 0)
        sb += struct.pack("<QQQQ", self.stored_base, UNDEF, len(self.buf) - BASE, UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
        self.buf[BASE:BASE + len(sb)] = sb
        return bytes(self.buf)


def write_mat73(path, variables: dict, writer: Writer = None):
    w = writer or Writer()
    data = w.finish(variables)
    with open(path, "wb") as fh:
        fh.write(data)
