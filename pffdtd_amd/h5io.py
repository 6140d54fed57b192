"""Minimal HDF5 dataset I/O for the file contract, via csrc/pf_h5.c + the system libhdf5.

h5py is not available in this image; the reference reads/writes these files with h5py
(python/fdtd/sim_fdtd.py:50-137,688-697) and the HDF5 C API (c_cuda/fdtd_data.h:746-860,928-980).
"""
import ctypes
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

F64, F32, I64, I8, BOOL, U8, I32 = range(7)
_CODE_TO_NP = {F64: np.float64, F32: np.float32, I64: np.int64, I8: np.int8, BOOL: np.int8, U8: np.uint8,
               I32: np.int32}


def _lib():
    global _LIB
    if _LIB is None:
        p = _HERE / "libpf_h5.so"
        if not p.exists():
            raise RuntimeError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(needs libhdf5)")
        L = ctypes.CDLL(str(p))
        L.pf_h5_last_error.restype = ctypes.c_char_p
        L.pf_h5_info.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int)]
        L.pf_h5_exists.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.pf_h5_read.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
        L.pf_h5_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.pf_h5_list.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64]
        _LIB = L
    return _LIB


def available():
    try:
        _lib()
        return True
    except (RuntimeError, OSError):
        return False


def _err():
    return _lib().pf_h5_last_error().decode()


def exists(path, name):
    return bool(_lib().pf_h5_exists(os.fsencode(str(path)), name.encode()))


def list_datasets(path):
    """Names of the datasets in the file's root group (sorted by name)."""
    buf = ctypes.create_string_buffer(1 << 16)
    n = _lib().pf_h5_list(os.fsencode(str(path)), buf, len(buf))
    if n < 0:
        raise IOError(_err())
    return [s for s in buf.value.decode().split("\n") if s]


def read(path, name, dtype=None):
    """Read dataset `name`; rank-0 datasets come back as numpy scalars (like h5py's ds[()])."""
    L = _lib()
    nd = ctypes.c_int()
    dims = (ctypes.c_int64 * 8)()
    cls = ctypes.c_int()
    size = ctypes.c_int()
    if L.pf_h5_info(os.fsencode(str(path)), name.encode(), nd, dims, cls, size) != 0:
        raise KeyError(_err())
    shape = tuple(int(dims[i]) for i in range(nd.value))
    if dtype is None:
        if cls.value == 1:
            code = F64
        elif cls.value == 2:
            code = BOOL
        elif cls.value == 0:
            code = I8 if size.value == 1 else I64
        else:
            raise TypeError(f"{path}::{name}: unsupported HDF5 class")
    else:
        code = dtype
    arr = np.empty(shape, dtype=_CODE_TO_NP[code])
    if L.pf_h5_read(os.fsencode(str(path)), name.encode(), code, arr.ctypes.data_as(ctypes.c_void_p),
                    int(arr.size)) != 0:
        raise IOError(_err())
    if code == BOOL and dtype is None:
        arr = arr.astype(np.bool_)
    return arr[()] if arr.ndim == 0 else arr


def write(path, name, data, code=None, append=True, gzip=0):
    """Write `data` as dataset `name`. bool arrays are stored as h5py-compatible enums."""
    a = np.asarray(data)
    if code is None:
        if a.dtype == np.bool_:
            code, a = BOOL, a.astype(np.int8)
        elif a.dtype == np.float64:
            code = F64
        elif a.dtype == np.float32:
            code = F32
        elif a.dtype == np.int64:
            code = I64
        elif a.dtype == np.int8:
            code = I8
        elif a.dtype == np.uint8:
            code = U8
        elif a.dtype == np.int32:
            code = I32
        elif np.issubdtype(a.dtype, np.integer):
            code, a = I64, a.astype(np.int64)
        elif np.issubdtype(a.dtype, np.floating):
            code, a = F64, a.astype(np.float64)
        else:
            raise TypeError(f"unsupported dtype {a.dtype}")
    shape = a.shape  # np.ascontiguousarray would promote rank-0 to rank-1; the reference needs rank-0 scalars
    a = np.ascontiguousarray(a, dtype=_CODE_TO_NP[code]).reshape(shape)
    dims = (ctypes.c_int64 * 8)(*shape)
    if _lib().pf_h5_write(os.fsencode(str(path)), name.encode(), code, a.ndim, dims,
                          a.ctypes.data_as(ctypes.c_void_p), 1 if append else 0, int(gzip)) != 0:
        raise IOError(_err())
