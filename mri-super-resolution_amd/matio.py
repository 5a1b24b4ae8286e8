"""MATLAB level-5 ``.mat`` files -- the container the reference's drivers read (``sio.loadmat``, superresDWI.py:41,
nn_mri.py:49-56) and write (``sio.savemat``, automate_INR.py:111).

Reader: numeric arrays (any numeric class, real), cell arrays (``hybrid_raw`` is a 4 x 4 cell of volumes) and
compressed elements (``miCOMPRESSED``, zlib).  Writer: numeric N-D arrays and (nested) lists / object arrays as cells,
uncompressed, little-endian, column-major as MATLAB stores them.  Level-7.3 (HDF5) files -- the reference falls back to
``mat73`` for those (superresDWI.py:42-43) -- are handed to ``mat73io.loadmat73`` (the HDF5 subset MATLAB writes, read without h5py).
Pure host-side I/O; nothing here touches the device.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict

import numpy as np

_MI = {1: "i1", 2: "u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 7: "<f4", 9: "<f8", 12: "<i8", 13: "<u8"}
_MI_OF = {"int8": 1, "uint8": 2, "int16": 3, "uint16": 4, "int32": 5, "uint32": 6, "float32": 7, "float64": 9,
          "int64": 12, "uint64": 13}
_MX_OF = {"float64": 6, "float32": 7, "int8": 8, "uint8": 9, "int16": 10, "uint16": 11, "int32": 12, "uint32": 13,
          "int64": 14, "uint64": 15}
_MX_DTYPE = {v: k for k, v in _MX_OF.items()}
miMATRIX, miCOMPRESSED, miUTF8, mxCELL, mxCHAR = 14, 15, 16, 1, 4


class MatFormatError(ValueError):
    pass


def _read_tag(buf, pos):
    word, = struct.unpack_from("<I", buf, pos)
    if word >> 16:                                   # small data element: type and byte count share the first word
        return word & 0xFFFF, word >> 16, pos + 4, pos + 8
    nbytes, = struct.unpack_from("<I", buf, pos + 4)
    return word, nbytes, pos + 8, pos + 8 + (nbytes + 7) // 8 * 8


def _read_numeric(buf, pos):
    mi, nbytes, data, nxt = _read_tag(buf, pos)
    if mi not in _MI:
        raise MatFormatError(f"unsupported data element type {mi}")
    return np.frombuffer(buf, dtype=_MI[mi], count=nbytes // np.dtype(_MI[mi]).itemsize, offset=data), nxt


def _read_matrix(buf, pos, end):
    if pos >= end:
        return None, np.zeros((0, 0))
    flags, pos = _read_numeric(buf, pos)
    klass = int(flags[0]) & 0xFF
    if int(flags[0]) & 0x0800:
        raise MatFormatError("complex arrays are not supported")
    dims, pos = _read_numeric(buf, pos)
    dims = tuple(int(d) for d in dims)
    name_raw, pos = _read_numeric(buf, pos)
    name = name_raw.tobytes().decode("ascii")
    if klass == mxCELL:
        cells = np.empty(int(np.prod(dims)), dtype=object)
        for i in range(cells.size):
            mi, nbytes, data, nxt = _read_tag(buf, pos)
            if mi != miMATRIX:
                raise MatFormatError("cell element is not a matrix")
            _, cells[i] = _read_matrix(buf, data, data + nbytes)
            pos = nxt
        return name, cells.reshape(dims, order="F")
    if klass == mxCHAR:
        mi, nbytes, data, _ = _read_tag(buf, pos)
        raw = bytes(buf[data:data + nbytes])
        text = raw.decode("utf-8") if mi in (miUTF8, 1, 2) else raw.decode("utf-16-le" if mi in (3, 4, 17) else "utf-32-le")
        return name, np.array(text)
    if klass not in _MX_DTYPE:
        raise MatFormatError(f"unsupported array class {klass} for variable '{name}'")
    real, pos = _read_numeric(buf, pos)
    arr = real.astype(_MX_DTYPE[klass], copy=False) if real.dtype != np.dtype(_MX_DTYPE[klass]) else real
    return name, np.array(arr.reshape(dims, order="F"))


def _mat73_module():
    try:
        from . import mat73io
        return mat73io
    except ImportError:                           # matio.py imported by path (tests do): load the sibling the same way
        import importlib.util
        import os
        import sys
        name = "_inr_mat73io"
        if name not in sys.modules:
            spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(os.path.abspath(__file__)), "mat73io.py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
        return sys.modules[name]


def loadmat(path: str) -> Dict[str, np.ndarray]:
    """``{name: ndarray}`` of a MAT-5 file (numeric arrays, cells as object arrays, strings)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:8] == b"\x89HDF\r\n\x1a\n" or buf[:10] == b"MATLAB 7.3":
        mod = _mat73_module()                     # the reference's fallback: `except NotImplementedError: mat73.loadmat(...)`
        try:
            return mod.loadmat73(path, buf)
        except mod.MatFormatError as e:           # (this file loaded on its own, outside the package: another class object)
            if isinstance(e, MatFormatError):
                raise
            raise MatFormatError(str(e)) from None
    if len(buf) < 128 or buf[126:128] != b"IM":
        raise MatFormatError(f"{path}: not a little-endian MAT-5 file")
    out, pos = {}, 128
    while pos + 8 <= len(buf):
        mi, nbytes, data, nxt = _read_tag(buf, pos)
        if mi == miCOMPRESSED:
            inner = zlib.decompress(buf[data:data + nbytes])
            mi2, nb2, d2, _ = _read_tag(inner, 0)
            if mi2 == miMATRIX:
                name, arr = _read_matrix(inner, d2, d2 + nb2)
                out[name] = arr
            nxt = data + nbytes                      # compressed elements are not padded
        elif mi == miMATRIX:
            name, arr = _read_matrix(buf, data, data + nbytes)
            out[name] = arr
        pos = nxt
    return out


def _tag(mi, payload: bytes) -> bytes:
    pad = (-len(payload)) % 8
    return struct.pack("<II", mi, len(payload)) + payload + b"\0" * pad


def _matrix(name: str, value) -> bytes:
    if isinstance(value, (list, tuple)) or (isinstance(value, np.ndarray) and value.dtype == object):
        cells = np.empty(len(value), dtype=object) if not isinstance(value, np.ndarray) else value
        if not isinstance(value, np.ndarray):
            for i, v in enumerate(value):
                cells[i] = v
            cells = cells.reshape(1, -1)
        dims = cells.shape if cells.ndim >= 2 else (1, cells.size)
        body = _tag(6, struct.pack("<II", mxCELL, 0)) + _tag(5, np.asarray(dims, "<i4").tobytes()) + \
            _tag(1, name.encode("ascii"))
        for v in cells.reshape(-1, order="F"):
            body += _matrix("", v)
        return _tag(miMATRIX, body)
    arr = np.asarray(value)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.name not in _MX_OF:
        raise MatFormatError(f"cannot store dtype {arr.dtype} (variable '{name}')")
    if arr.ndim < 2:
        arr = arr.reshape(1, -1)
    body = _tag(6, struct.pack("<II", _MX_OF[arr.dtype.name], 0)) + _tag(5, np.asarray(arr.shape, "<i4").tobytes()) + \
        _tag(1, name.encode("ascii")) + \
        _tag(_MI_OF[arr.dtype.name], np.asarray(arr, arr.dtype.newbyteorder("<")).tobytes(order="F"))
    return _tag(miMATRIX, body)


def savemat(path: str, variables: Dict[str, object]) -> None:
    """Writes ``variables`` (numeric arrays; lists / object arrays become cell arrays) as an uncompressed MAT-5 file."""
    header = b"MATLAB 5.0 MAT-file, written by mri-super-resolution_amd".ljust(116) + b"\0" * 8 + struct.pack("<H", 0x0100) + b"IM"
    with open(path, "wb") as fh:
        fh.write(header)
        for name, value in variables.items():
            fh.write(_matrix(name, value))
