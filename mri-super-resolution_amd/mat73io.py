"""MATLAB 7.3 ``.mat`` files: the HDF5 subset MATLAB's ``save -v7.3`` writes, read without h5py.

The reference falls back to ``mat73.loadmat`` when ``scipy.io.loadmat`` refuses a file (``superresDWI.py:40-43``,
``superresHybrid.py:36-39``, ``inrDWI.py:33-36``) -- the ``hybrid_raw`` ``master.mat`` volumes of its primary driver are large
enough to need it.  Neither h5py nor mat73 exists in this image, so this module restates the file format itself (HDF5 File Format
Specification, version-0 superblock era, which is what MATLAB -- libver "earliest" -- emits):

* 512-byte user block (the ``MATLAB 7.3 MAT-file`` text header), superblock version 0 / 1, base address;
* old-style groups: symbol-table message -> version-1 B-tree (``TREE``) -> symbol-table nodes (``SNOD``) + local heap (``HEAP``);
* version-1 object headers with continuation blocks; messages: dataspace (v1 / v2), datatype (fixed point, floating point, string,
  object reference), data layout v3 (compact, contiguous, chunked with a version-1 chunk B-tree), filter pipeline (v1 / v2: deflate,
  shuffle), attribute (v1 - v3);
* MATLAB's conventions on top: dimensions stored reversed (column-major data seen as a C-order array of the reversed shape),
  the ``MATLAB_class`` attribute (numeric classes, ``logical``, ``char`` as UTF-16 code units, ``cell`` as a dataset of object
  references into ``/#refs#``, ``struct`` as a group), ``MATLAB_empty``.

What comes back mirrors ``matio.loadmat`` (MAT-5): numeric arrays in MATLAB's shape, cells as object arrays of that shape (so
``data['hybrid_raw'][b][te]`` reads as in the reference, where mat73 hands out nested lists), structs as dicts, strings as ``str``.
Not read: version-2 object headers / fractal heaps (written by newer libver settings, not by MATLAB), compound types (complex
numbers), variable-length types, sparse matrices, function handles -- each raises ``MatFormatError`` naming the construct.

PARITY NOTE: there is no MATLAB-written 7.3 file in the reference tree and no HDF5 library in the image; the tests read files put
together by ``tests/hdf5_fixture.py`` from the same specification (chunk B-trees, filters, references, user block), i.e. reader and
fixture writer are two restatements of one document -- unpinned against MATLAB itself, and said so in DESIGN.md.
Pure host-side I/O; nothing here touches the device.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

try:
    from .matio import MatFormatError
except ImportError:                               # loaded by path next to a matio.py that was loaded by path (see matio._mat73_module)
    class MatFormatError(ValueError):
        pass

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, buf: bytes, path: str):
        self.buf, self.path = buf, path
        at = 0
        while at < len(buf) and buf[at:at + 8] != _SIG:      # the superblock sits at 0, 512, 1024, ... (MATLAB: 512)
            at = 512 if at == 0 else at * 2
        if at >= len(buf):
            raise MatFormatError(f"{path}: no HDF5 superblock (not a MATLAB 7.3 file)")
        ver = buf[at + 8]
        if ver not in (0, 1):
            raise MatFormatError(f"{path}: HDF5 superblock version {ver} (MATLAB writes 0; newer layouts are not read)")
        self.o, self.l = buf[at + 13], buf[at + 14]          # size of offsets / lengths
        if (self.o, self.l) != (8, 8):
            raise MatFormatError(f"{path}: {self.o}-byte offsets / {self.l}-byte lengths are not supported")
        p = at + 24 + (4 if ver == 1 else 0)
        _stored_base, _free, _eof, _drv = struct.unpack_from("<4Q", buf, p)
        # libhdf5 (H5F__super_read) takes the place where it FOUND the superblock as the base of every address and overrides the
        # stored field when the two differ (a 512-byte user block in front of a file whose stored base address is 0)
        self.base = at
        self.root = self._symbol_entry(p + 32)

    # -- primitives ------------------------------------------------------------------------------------------------
    def abs(self, rel: int) -> int:
        return self.base + rel

    def _symbol_entry(self, p: int) -> Dict[str, int]:
        name_off, header, cache = struct.unpack_from("<QQI", self.buf, p)
        e = {"name_off": name_off, "header": header, "cache": cache}
        if cache == 1:
            e["btree"], e["heap"] = struct.unpack_from("<QQ", self.buf, p + 24)
        return e

    def _heap_data(self, addr: int) -> int:
        p = self.abs(addr)
        if self.buf[p:p + 4] != b"HEAP":
            raise MatFormatError(f"{self.path}: local heap signature missing at {p}")
        return self.abs(struct.unpack_from("<Q", self.buf, p + 24)[0])

    def _cstr(self, p: int) -> str:
        return self.buf[p:self.buf.index(b"\0", p)].decode("utf-8")

    # -- object headers --------------------------------------------------------------------------------------------------
    def messages(self, header_addr: int) -> List[Tuple[int, int, int]]:
        """[(type, offset of the message body, body size)] of the version-1 object header at ``header_addr``."""
        p = self.abs(header_addr)
        if self.buf[p:p + 4] == b"OHDR":
            raise MatFormatError(f"{self.path}: version-2 object header (not written by MATLAB; not read)")
        if self.buf[p] != 1:
            raise MatFormatError(f"{self.path}: object header version {self.buf[p]} at {p}")
        nmsg, = struct.unpack_from("<H", self.buf, p + 2)
        size, = struct.unpack_from("<I", self.buf, p + 8)
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize = struct.unpack_from("<HH", self.buf, q)
                body = q + 8
                if mtype == 0x0010:                      # continuation: more messages elsewhere
                    off, length = struct.unpack_from("<QQ", self.buf, body)
                    blocks.append((self.abs(off), length))
                out.append((mtype, body, msize))
                q = body + msize
        return out

    # -- groups ------------------------------------------------------------------------------------------------------------
    def group_members(self, btree: int, heap: int) -> Dict[str, int]:
        """{name: object header address} of an old-style group."""
        names = self._heap_data(heap)
        out: Dict[str, int] = {}

        def walk(addr):
            p = self.abs(addr)
            sig = self.buf[p:p + 4]
            if sig == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", self.buf, p + 4)
                if ntype != 0:
                    raise MatFormatError(f"{self.path}: chunk B-tree where a group B-tree was expected")
                q = p + 24
                for i in range(used):                    # key_i, child_i, ..., key_used
                    child, = struct.unpack_from("<Q", self.buf, q + 8 + 16 * i)
                    walk(child)
            elif sig == b"SNOD":
                n, = struct.unpack_from("<H", self.buf, p + 6)
                for i in range(n):
                    e = self._symbol_entry(p + 8 + 40 * i)
                    out[self._cstr(names + e["name_off"])] = e["header"]
            else:
                raise MatFormatError(f"{self.path}: unexpected node {sig!r} in a group B-tree")

        walk(btree)
        return out

    # -- message decoders --------------------------------------------------------------------------------------------------
    def dataspace(self, p: int) -> Tuple[int, ...]:
        ver, rank, flags = self.buf[p], self.buf[p + 1], self.buf[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.buf[p + 3] == 2:                      # null dataspace
                return (0,)
            q = p + 4
        else:
            raise MatFormatError(f"{self.path}: dataspace message version {ver}")
        return tuple(struct.unpack_from(f"<{rank}Q", self.buf, q)) if rank else ()

    def datatype(self, p: int):
        """('num', numpy dtype) | ('ref', 8) | ('str', size)"""
        cls, ver = self.buf[p] & 0x0F, self.buf[p] >> 4
        bits0 = self.buf[p + 1]
        size, = struct.unpack_from("<I", self.buf, p + 4)
        if ver not in (1, 2, 3):
            raise MatFormatError(f"{self.path}: datatype message version {ver}")
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return "num", np.dtype(f"{order}{'i' if bits0 & 8 else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise MatFormatError(f"{self.path}: {size}-byte floating-point type")
            return "num", np.dtype(f"{order}f{size}")
        if cls == 3:
            return "str", size
        if cls == 7:
            if bits0 & 0x0F:
                raise MatFormatError(f"{self.path}: dataset region references are not read")
            return "ref", size
        names = {2: "time", 4: "bit field", 5: "opaque", 6: "compound (complex numbers?)", 8: "enumeration", 9: "variable-length",
                 10: "array"}
        raise MatFormatError(f"{self.path}: {names.get(cls, f'class {cls}')} datatypes are not read")

    def filters(self, p: int) -> List[Tuple[int, Tuple[int, ...]]]:
        ver, n = self.buf[p], self.buf[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid, = struct.unpack_from("<H", self.buf, q)
            if ver == 1 or fid >= 256:
                nlen, = struct.unpack_from("<H", self.buf, q + 2)
                q += 4
            else:
                nlen = 0
                q += 2
            _flags, nvals = struct.unpack_from("<HH", self.buf, q)
            q += 4
            q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            vals = struct.unpack_from(f"<{nvals}I", self.buf, q)
            q += 4 * nvals + (4 if ver == 1 and nvals % 2 else 0)
            out.append((fid, vals))
        return out

    # -- raw data ----------------------------------------------------------------------------------------------------------
    def read_raw(self, msgs, itemsize: int, dims: Tuple[int, ...]) -> bytes:
        """The dataset's elements as one C-order byte string (HDF5 dimension order)."""
        layout = next(((b, s) for t, b, s in msgs if t == 0x0008), None)
        if layout is None:
            raise MatFormatError(f"{self.path}: dataset without a data layout message")
        p = layout[0]
        if self.buf[p] != 3:
            raise MatFormatError(f"{self.path}: data layout message version {self.buf[p]} (HDF5 1.8 writes 3)")
        total = int(np.prod(dims, dtype=np.int64)) * itemsize if dims else itemsize
        kind = self.buf[p + 1]
        if kind == 0:                                     # compact: the data sit in the header
            n, = struct.unpack_from("<H", self.buf, p + 2)
            return bytes(self.buf[p + 4:p + 4 + n])
        if kind == 1:
            addr, _n = struct.unpack_from("<QQ", self.buf, p + 2)
            if addr == _UNDEF:
                return b"\0" * total
            return bytes(self.buf[self.abs(addr):self.abs(addr) + total])
        if kind != 2:
            raise MatFormatError(f"{self.path}: data layout class {kind}")
        ndim = self.buf[p + 2]                            # rank + 1 (the last "dimension" is the element size)
        btree, = struct.unpack_from("<Q", self.buf, p + 3)
        chunk = struct.unpack_from(f"<{ndim}I", self.buf, p + 11)[:-1]
        pipeline = next((self.filters(b) for t, b, s in msgs if t == 0x000B), [])
        out = bytearray(total)
        view = np.frombuffer(out, dtype=np.uint8).reshape(tuple(dims) + (itemsize,))
        if btree == _UNDEF:
            return bytes(out)

        def put(offsets, raw):
            block = np.frombuffer(raw, dtype=np.uint8, count=int(np.prod(chunk)) * itemsize).reshape(tuple(chunk) + (itemsize,))
            dst = tuple(slice(o, min(o + c, d)) for o, c, d in zip(offsets, chunk, dims))
            src = tuple(slice(0, s.stop - s.start) for s in dst)
            view[dst] = block[src]

        def walk(addr):
            q = self.abs(addr)
            if self.buf[q:q + 4] != b"TREE" or self.buf[q + 4] != 1:
                raise MatFormatError(f"{self.path}: chunk B-tree node expected at {q}")
            level, used = struct.unpack_from("<BH", self.buf, q + 5)
            keysize = 8 + 8 * ndim
            q += 24
            for i in range(used):
                nbytes, mask = struct.unpack_from("<II", self.buf, q)
                offsets = struct.unpack_from(f"<{ndim}Q", self.buf, q + 8)[:-1]
                child, = struct.unpack_from("<Q", self.buf, q + keysize)
                q += keysize + 8
                if level > 0:
                    walk(child)
                    continue
                raw = bytes(self.buf[self.abs(child):self.abs(child) + nbytes])
                for k in reversed(range(len(pipeline))):  # undo the pipeline last filter first
                    if mask & (1 << k):
                        continue
                    fid, vals = pipeline[k]
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:                        # shuffle: byte planes -> elements
                        es = vals[0] if vals else itemsize
                        n = len(raw) // es
                        raw = np.frombuffer(raw, np.uint8, n * es).reshape(es, n).T.tobytes() + raw[n * es:]
                    else:
                        raise MatFormatError(f"{self.path}: HDF5 filter {fid} is not read (deflate and shuffle are)")
                put(offsets, raw)

        walk(btree)
        return bytes(out)

    def attributes(self, msgs) -> Dict[str, object]:
        out = {}
        for t, p, _s in msgs:
            if t != 0x000C:
                continue
            ver = self.buf[p]
            nsz, tsz, ssz = struct.unpack_from("<HHH", self.buf, p + 2)
            q = p + (9 if ver == 3 else 8)
            pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
            name = self.buf[q:q + nsz].split(b"\0")[0].decode("utf-8")
            q += pad(nsz)
            try:
                kind, info = self.datatype(q)
            except MatFormatError:
                continue                                   # an attribute of a type this reader does not know: not needed
            dims = self.dataspace(q + pad(tsz))
            data = q + pad(tsz) + pad(ssz)
            count = int(np.prod(dims, dtype=np.int64)) if dims else 1
            if kind == "str":
                out[name] = self.buf[data:data + info].split(b"\0")[0].decode("utf-8")
            elif kind == "num":
                out[name] = np.frombuffer(self.buf, dtype=info, count=count, offset=data).copy()
        return out


def _convert(f: _File, header: int, depth: int = 0):
    if depth > 64:
        raise MatFormatError(f"{f.path}: reference chain too deep")
    msgs = f.messages(header)
    attrs = f.attributes(msgs)
    klass = attrs.get("MATLAB_class", "")
    sym = next((b for t, b, s in msgs if t == 0x0011), None)
    if sym is not None:                                    # a group: struct (or the file's root)
        btree, heap = struct.unpack_from("<QQ", f.buf, sym)
        return {k: _convert(f, h, depth + 1) for k, h in f.group_members(btree, heap).items() if not k.startswith("#")}
    space = next((b for t, b, s in msgs if t == 0x0001), None)
    dtype = next((b for t, b, s in msgs if t == 0x0003), None)
    if space is None or dtype is None:
        raise MatFormatError(f"{f.path}: object at {header} is neither a group nor a dataset")
    dims = f.dataspace(space)
    kind, info = f.datatype(dtype)
    if klass in ("function_handle", "sparse") or "MATLAB_sparse" in attrs:
        raise MatFormatError(f"{f.path}: MATLAB class '{klass or 'sparse'}' is not read")
    if "MATLAB_empty" in attrs:                            # the dataset holds the DIMENSIONS of an empty array
        shape = tuple(int(v) for v in np.frombuffer(f.read_raw(msgs, info.itemsize, dims), dtype=info))
        if klass == "char":
            return ""
        np_class = {"cell": object, "logical": bool, "double": np.float64, "single": np.float32, "struct": object, "": np.float64}
        return np.empty(shape, dtype=np_class.get(klass, klass))
    if kind == "ref":
        refs = np.frombuffer(f.read_raw(msgs, 8, dims), dtype="<u8")
        cells = np.empty(len(refs), dtype=object)
        for i, r in enumerate(refs):
            cells[i] = _convert(f, int(r), depth + 1)
        return cells.reshape(dims).T if dims else cells    # reversed dimensions -> MATLAB's shape
    if kind != "num":
        raise MatFormatError(f"{f.path}: string datasets are not read")
    arr = np.frombuffer(f.read_raw(msgs, info.itemsize, dims), dtype=info).reshape(dims).T
    arr = np.ascontiguousarray(arr.astype(info.newbyteorder("="), copy=False))
    if klass == "char":
        return "".join(chr(int(c)) for c in arr.reshape(-1, order="F"))
    if klass == "logical":
        return arr.astype(bool)
    return arr


def is_mat73(head: bytes) -> bool:
    return head[:10] == b"MATLAB 7.3" or head[:8] == _SIG


def loadmat73(path: str, buf: bytes = None) -> Dict[str, object]:
    """``{name: value}`` of a MATLAB 7.3 (HDF5) file: what ``mat73.loadmat`` gives the reference (``superresDWI.py:43``), with cells
    as object arrays (``value[b][te]`` works as there) instead of nested lists.  ``buf``: the file's bytes when the caller has
    already read them (``matio.loadmat`` has)."""
    if buf is None:
        with open(path, "rb") as fh:
            buf = fh.read()
    try:
        return _load(buf, path)
    except (struct.error, IndexError, KeyError, zlib.error, UnicodeDecodeError, OverflowError) as e:
        # a truncated or damaged file walks off a structure: say what it is instead of leaking the parser's exception
        raise MatFormatError(f"{path}: damaged or truncated MATLAB 7.3 file ({type(e).__name__}: {e})") from None


def _load(buf: bytes, path: str) -> Dict[str, object]:
    f = _File(buf, path)
    if f.root.get("cache") == 1:
        members = f.group_members(f.root["btree"], f.root["heap"])
    else:                                                  # the root's symbol table lives in its object header
        sym = next((b for t, b, s in f.messages(f.root["header"]) if t == 0x0011), None)
        if sym is None:
            raise MatFormatError(f"{path}: root group without a symbol table (new-style groups are not read)")
        members = f.group_members(*struct.unpack_from("<QQ", buf, sym))
    return {k: _convert(f, h) for k, h in members.items() if not k.startswith("#")}
