"""HDF5 files through the system's ``libhdf5`` (ctypes), with the slice of h5py's API that ManiSkill's trajectory code uses.

The reference stores trajectories as HDF5 via h5py (mani_skill/utils/wrappers/record.py:271,574-700: one ``traj_<id>`` group per episode,
``create_group(track_order=True)``, ``create_dataset(name, data=, dtype=, compression="gzip", compression_opts=5)``;
mani_skill/trajectory/dataset.py:16-41, merge_trajectory.py:30-60, replay_trajectory.py:397: ``File``, ``Group``, ``Dataset``, ``keys()``,
``len()``, ``group[name]``, ``group[new] = group[old]``, ``del group[name]``).  h5py is a wheel this image does not have, but the HDF5 C library
itself is there (``/opt/conda/lib/libhdf5.so``, 1.10): this module binds the few dozen entry points those calls need, so the files written
here ARE HDF5 -- ``h5dump`` / any h5py / the reference's own tools read them, and files recorded by the reference (bool as h5py's
FALSE/TRUE enum, gzip chunks, creation-ordered groups) are read here.

Objects keep (file, path) and open HDF5 ids only for the duration of a call: nothing leaks when a file is closed with datasets still
referenced.  Not covered (raises): variable-length / compound types, region references, partial I/O (a dataset is read whole, then indexed).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
from typing import Optional

import numpy as np

_CANDIDATES = [os.environ.get("MSK_HDF5_LIB"), "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
               "/usr/lib/x86_64-linux-gnu/libhdf5.so", ctypes.util.find_library("hdf5"), ctypes.util.find_library("hdf5_serial")]
_lib = None
_err: Optional[str] = None

hid_t, herr_t, hsize_t = C.c_int64, C.c_int, C.c_uint64
H5P_DEFAULT, H5S_ALL = 0, 0
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5I_FILE, H5I_GROUP, H5I_DATASET = 1, 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5_INDEX_NAME, H5_INDEX_CRT_ORDER, H5_ITER_INC = 0, 1, 0


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint), ("_pad", C.c_uint * 4)]


def _load():
    global _lib, _err
    if _lib is not None or _err is not None:
        return _lib
    for path in _CANDIDATES:
        if not path:
            continue
        try:
            lib = C.CDLL(path)
        except OSError as e:
            _err = str(e)
            continue
        sig = {
            "H5open": (herr_t, []), "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, C.c_int]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Gclose": (herr_t, [hid_t]), "H5Gget_info": (herr_t, [hid_t, C.POINTER(_GInfo)]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]), "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
            "H5Lcreate_hard": (herr_t, [hid_t, C.c_char_p, hid_t, C.c_char_p, hid_t, hid_t]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]),
            "H5Pset_link_creation_order": (herr_t, [hid_t, C.c_uint]), "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pget_nfilters": (C.c_int, [hid_t]),
            "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (herr_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_create_plist": (hid_t, [hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_sign": (C.c_int, [hid_t]), "H5Tget_order": (C.c_int, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tenum_create": (hid_t, [hid_t]), "H5Tenum_insert": (herr_t, [hid_t, C.c_char_p, C.c_void_p]), "H5Tget_nmembers": (C.c_int, [hid_t]),
            "H5Tget_super": (hid_t, [hid_t]), "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aclose": (herr_t, [hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Adelete": (herr_t, [hid_t, C.c_char_p]), "H5Aget_num_attrs": (C.c_int, [hid_t]),
            "H5Aget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            if lib.H5open() < 0:
                raise OSError("H5open failed")
            va, vb, vc = C.c_uint(), C.c_uint(), C.c_uint()
            lib.H5get_libversion(C.byref(va), C.byref(vb), C.byref(vc))
            if (va.value, vb.value) < (1, 10):      # hid_t is 64-bit from 1.10 on (int in 1.8): the ids read from the H5T_* / H5P_* globals would be garbage
                raise OSError(f"libhdf5 {va.value}.{vb.value}.{vc.value} is older than 1.10 (32-bit hid_t): not supported")
        except (AttributeError, OSError) as e:
            _err = f"{path}: {e}"
            continue
        lib.H5Eset_auto2(0, None, None)     # errors come back as Python exceptions, not as stack dumps on stderr
        _lib, _err = lib, None
        return _lib
    if _err is None:
        _err = "libhdf5 not found (set MSK_HDF5_LIB)"
    return None


def available() -> bool:
    """True when libhdf5 could be loaded: only then does this module write or read anything."""
    return _load() is not None


def _need():
    lib = _load()
    if lib is None:
        raise ImportError(f"maniskill_amd.hdf5 needs the HDF5 C library: {_err}")
    return lib


def _g(name) -> int:
    return hid_t.in_dll(_need(), name).value


class _Version:
    @property
    def hdf5_version(self):
        lib = _need()
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        lib.H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
        return f"{a.value}.{b.value}.{c.value}"

    version = "maniskill_amd.hdf5 (ctypes over libhdf5)"


version = _Version()
__version__ = "0.1+libhdf5"

# numpy dtype <-> HDF5 native type ------------------------------------------------------------------------------------------------------
_NATIVE = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
           "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g",
           "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g"}


def _bool_type():
    """h5py's mapping of numpy bool: an enum over int8 with members FALSE = 0, TRUE = 1"""
    lib = _need()
    t = lib.H5Tenum_create(_g("H5T_NATIVE_INT8_g"))
    for name, v in ((b"FALSE", 0), (b"TRUE", 1)):
        val = C.c_int8(v)
        lib.H5Tenum_insert(t, name, C.byref(val))
    return t


def _h5_type(dt: np.dtype):
    """(type id, must_close) for a numpy dtype"""
    lib = _need()
    dt = np.dtype(dt)
    if dt == np.bool_:
        return _bool_type(), True
    if dt.kind == "U":      # numpy's UCS-4 would be written as raw bytes and end at the first NUL: callers encode first (_as_bytes)
        raise TypeError("maniskill_amd.hdf5: unicode arrays are written as UTF-8 bytes: np.char.encode(a, 'utf-8') first")
    if dt.kind == "S":      # fixed-length strings (attributes): ASCII / UTF-8 bytes, null padded
        t = lib.H5Tcopy(_g("H5T_C_S1_g"))
        lib.H5Tset_size(t, max(int(dt.itemsize), 1))
        return t, True
    key = dt.kind + str(dt.itemsize)
    if key not in _NATIVE or dt.byteorder == ">":
        raise TypeError(f"maniskill_amd.hdf5: dtype {dt} is not supported (f4 f8 i1..i8 u1..u8 bool)")
    return _g(_NATIVE[key]), False


def _np_type(tid) -> np.dtype:
    lib = _need()
    cls, size = lib.H5Tget_class(tid), int(lib.H5Tget_size(tid))
    if cls == H5T_FLOAT and size in (4, 8):
        return np.dtype(f"f{size}")
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("i" if lib.H5Tget_sign(tid) == 1 else "u") + str(size))
    if cls == H5T_ENUM:
        sup = lib.H5Tget_super(tid)
        ok = lib.H5Tget_size(sup) == 1 and lib.H5Tget_nmembers(tid) == 2
        lib.H5Tclose(sup)
        if ok:
            return np.dtype(bool)
    if cls == H5T_STRING and lib.H5Tis_variable_str(tid) <= 0:
        return np.dtype(f"S{size}")
    raise TypeError(f"maniskill_amd.hdf5: HDF5 type class {cls} of size {size} is not supported")


def _c(x) -> np.ndarray:
    """C-contiguous array that keeps 0-d (np.ascontiguousarray would make scalars 1-d)"""
    a = np.asarray(x)
    if a.dtype.kind == "U":      # unicode arrays go to the file as UTF-8 bytes (numpy's UCS-4 buffer would end at the first NUL when read as a C string)
        a = np.char.encode(a, "utf-8")
    return a if a.flags.c_contiguous else np.array(a, order="C")


def _check(code, what):
    if code < 0:
        raise OSError(f"HDF5: {what} failed")
    return code


class AttributeManager:
    """``obj.attrs``: numeric scalars / arrays and strings."""

    def __init__(self, node):
        self._node = node

    def _names(self):
        lib = _need()
        with self._node._open() as oid:
            n = lib.H5Aget_num_attrs(oid)
            out = []
            for i in range(max(n, 0)):
                ln = lib.H5Aget_name_by_idx(oid, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
                buf = C.create_string_buffer(ln + 1)
                lib.H5Aget_name_by_idx(oid, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, ln + 1, H5P_DEFAULT)
                out.append(buf.value.decode())
            return out

    def keys(self): return self._names()
    def __iter__(self): return iter(self._names())
    def __len__(self): return len(self._names())
    def items(self): return [(k, self[k]) for k in self._names()]

    def __contains__(self, name):
        with self._node._open() as oid:
            return _need().H5Aexists(oid, name.encode()) > 0

    def __setitem__(self, name, value):
        lib = _need()
        if isinstance(value, str):
            value = np.array(value.encode("utf-8"))
        arr = _c(value)
        tid, own = _h5_type(arr.dtype)
        if arr.ndim == 0:
            sid = lib.H5Screate(0)
        else:
            dims = (hsize_t * arr.ndim)(*arr.shape)
            sid = lib.H5Screate_simple(arr.ndim, dims, None)
        with self._node._open() as oid:
            if lib.H5Aexists(oid, name.encode()) > 0:
                lib.H5Adelete(oid, name.encode())
            aid = _check(lib.H5Acreate2(oid, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT), f"create attribute {name}")
            try:
                _check(lib.H5Awrite(aid, tid, arr.ctypes.data_as(C.c_void_p)), f"write attribute {name}")
            finally:
                lib.H5Aclose(aid); lib.H5Sclose(sid)
                if own:
                    lib.H5Tclose(tid)

    def __getitem__(self, name):
        lib = _need()
        with self._node._open() as oid:
            aid = lib.H5Aopen(oid, name.encode(), H5P_DEFAULT)
            if aid < 0:
                raise KeyError(name)
            tid, sid = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
            try:
                dt = _np_type(tid)
                nd = lib.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sid, dims, None)
                out = np.empty(tuple(int(d) for d in dims[:nd]), dtype=np.int8 if dt == np.bool_ else dt)
                _check(lib.H5Aread(aid, tid, out.ctypes.data_as(C.c_void_p)), f"read attribute {name}")
            finally:
                lib.H5Tclose(tid); lib.H5Sclose(sid); lib.H5Aclose(aid)
        if dt == np.bool_:
            out = out.astype(bool)
        if out.dtype.kind == "S":
            return out.item().decode("utf-8") if out.ndim == 0 else out
        return out.item() if out.ndim == 0 else out

    def get(self, name, default=None):
        return self[name] if name in self else default


class _Opened:
    def __init__(self, node):
        self.node = node

    def __enter__(self):
        lib = _need()
        f = self.node.file
        if f._fid is None:
            raise ValueError("the file is closed")
        self.oid = lib.H5Oopen(f._fid, self.node.name.encode(), H5P_DEFAULT)
        if self.oid < 0:
            raise KeyError(f"no object {self.node.name!r} in {f.filename}")
        return self.oid

    def __exit__(self, *exc):
        _need().H5Oclose(self.oid)


class _Node:
    def __init__(self, file, path):
        self._file, self._path = file, path

    file = property(lambda self: self._file)
    name = property(lambda self: self._path)
    attrs = property(lambda self: AttributeManager(self))

    @property
    def parent(self):
        return Group(self._file, self._path.rsplit("/", 1)[0] or "/")

    def _open(self):
        return _Opened(self)

    def __eq__(self, other):
        return isinstance(other, _Node) and other._file is self._file and other._path == self._path

    def __hash__(self):
        return hash((id(self._file), self._path))


class Dataset(_Node):
    def _meta(self):
        lib = _need()
        with self._open() as did:
            tid, sid = lib.H5Dget_type(did), lib.H5Dget_space(did)
            try:
                nd = lib.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sid, dims, None)
                return tuple(int(d) for d in dims[:nd]), _np_type(tid)
            finally:
                lib.H5Tclose(tid); lib.H5Sclose(sid)

    shape = property(lambda self: self._meta()[0])
    dtype = property(lambda self: self._meta()[1])
    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))

    @property
    def compression(self):
        lib = _need()
        with self._open() as did:
            pl = lib.H5Dget_create_plist(did)
            n = lib.H5Pget_nfilters(pl)
            lib.H5Pclose(pl)
        return "gzip" if n > 0 else None

    def _read(self) -> np.ndarray:
        lib = _need()
        with self._open() as did:
            tid, sid = lib.H5Dget_type(did), lib.H5Dget_space(did)
            try:
                dt = _np_type(tid)
                nd = lib.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sid, dims, None)
                shape = tuple(int(d) for d in dims[:nd])
                if dt == np.bool_ or dt.kind == "S":
                    out = np.empty(shape, dtype=np.int8 if dt == np.bool_ else dt)
                    mem, own = tid, False                       # read in the file's own type (an int8 enum / fixed string)
                else:
                    out = np.empty(shape, dtype=dt)
                    mem, own = _h5_type(dt)
                if out.size:
                    _check(lib.H5Dread(did, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), f"read {self.name}")
                if own:
                    lib.H5Tclose(mem)
            finally:
                lib.H5Tclose(tid); lib.H5Sclose(sid)
        return out.astype(bool) if dt == np.bool_ else out

    def __getitem__(self, key):
        a = self._read()
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return a if a.ndim else a[()]
        return a[key]

    def __setitem__(self, key, value):
        a = self._read()
        a[key] = value
        self._write(a)

    def _write(self, arr):
        lib = _need()
        arr = _c(arr)
        with self._open() as did:
            tid = lib.H5Dget_type(did)
            try:
                if arr.dtype == np.bool_:
                    arr = arr.astype(np.int8)
                    mem, own = tid, False
                else:
                    mem, own = _h5_type(arr.dtype)
                _check(lib.H5Dwrite(did, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)), f"write {self.name}")
                if own:
                    lib.H5Tclose(mem)
            finally:
                lib.H5Tclose(tid)

    def __array__(self, dtype=None, copy=None):
        a = self._read()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self._read())

    def asstr(self):
        return np.char.decode(self._read(), "utf-8")

    def __repr__(self):
        shape, dt = self._meta()
        return f'<HDF5 dataset "{self.name.rsplit("/", 1)[-1]}": shape {shape}, type "{dt.str}">'


class Group(_Node):
    def _abs(self, name: str) -> str:
        if name.startswith("/"):
            return name.rstrip("/") or "/"
        base = self._path.rstrip("/")
        return f"{base}/{name}".rstrip("/")

    def _writable(self):
        if self._file.mode == "r":
            raise OSError(f"{self._file.filename} is open read-only")

    def _exists(self, path: str) -> bool:
        lib, fid = _need(), self._file._fid
        if fid is None:
            raise ValueError("the file is closed")
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if lib.H5Lexists(fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __contains__(self, name):
        return self._exists(self._abs(name))

    def _wrap(self, path):
        lib = _need()
        oid = lib.H5Oopen(self._file._fid, path.encode(), H5P_DEFAULT)
        if oid < 0:
            raise KeyError(f"no object {path!r} in {self._file.filename}")
        kind = lib.H5Iget_type(oid)
        lib.H5Oclose(oid)
        return Dataset(self._file, path) if kind == H5I_DATASET else Group(self._file, path)

    def __getitem__(self, name):
        path = self._abs(name)
        if path != "/" and not self._exists(path):
            raise KeyError(f"Unable to open object (object {name!r} doesn't exist)")
        return self._wrap(path)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        lib = _need()
        with self._open() as gid:
            info = _GInfo()
            _check(lib.H5Gget_info(gid, C.byref(info)), "H5Gget_info")
            names = []
            for order in (H5_INDEX_CRT_ORDER, H5_INDEX_NAME):     # creation order where the group tracks it (track_order=True), else by name
                names = []
                for i in range(int(info.nlinks)):
                    ln = lib.H5Lget_name_by_idx(gid, b".", order, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
                    if ln < 0:
                        names = None
                        break
                    buf = C.create_string_buffer(ln + 1)
                    lib.H5Lget_name_by_idx(gid, b".", order, H5_ITER_INC, i, buf, ln + 1, H5P_DEFAULT)
                    names.append(buf.value.decode())
                if names is not None:
                    break
            return names or []

    def __iter__(self): return iter(self.keys())
    def __len__(self): return len(self.keys())
    def values(self): return [self[k] for k in self.keys()]
    def items(self): return [(k, self[k]) for k in self.keys()]

    def create_group(self, name, track_order=None):
        self._writable()
        lib, path = _need(), self._abs(name)
        if self._exists(path):
            raise ValueError(f"Unable to create group (name {name!r} already exists)")
        lcpl = lib.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        gcpl = H5P_DEFAULT
        if track_order:
            gcpl = lib.H5Pcreate(_g("H5P_CLS_GROUP_CREATE_ID_g"))
            lib.H5Pset_link_creation_order(gcpl, 3)       # H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED
        gid = lib.H5Gcreate2(self._file._fid, path.encode(), lcpl, gcpl, H5P_DEFAULT)
        lib.H5Pclose(lcpl)
        if gcpl:
            lib.H5Pclose(gcpl)
        _check(gid, f"create group {path}")
        lib.H5Gclose(gid)
        return Group(self._file, path)

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def create_dataset(self, name, shape=None, dtype=None, data=None, compression=None, compression_opts=None, chunks=None, **_unused):
        self._writable()
        lib, path = _need(), self._abs(name)
        if data is None:
            arr = np.zeros(shape if shape is not None else (), dtype=dtype or np.float32)
        else:
            arr = np.asarray(data)
            if arr.dtype == object:
                raise TypeError(f"maniskill_amd.hdf5: object arrays are not supported ({name})")
            if dtype is not None:
                arr = arr.astype(dtype, copy=False)
            if shape is not None and tuple(np.atleast_1d(shape)) != arr.shape:
                arr = arr.reshape(shape)
        arr = _c(arr)
        if self._exists(path):
            raise ValueError(f"Unable to create dataset (name {name!r} already exists)")
        tid, own = _h5_type(arr.dtype)
        if arr.ndim == 0:
            sid = lib.H5Screate(0)
        else:
            dims = (hsize_t * arr.ndim)(*arr.shape)
            sid = lib.H5Screate_simple(arr.ndim, dims, None)
        lcpl = lib.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        dcpl = H5P_DEFAULT
        if (compression in ("gzip", True) or isinstance(compression, int) or chunks) and arr.ndim > 0 and arr.size > 0:
            dcpl = lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            if chunks in (None, True):       # ~1 MiB chunks along the first axis (h5py guesses similarly)
                row = max(int(arr[0].nbytes) if arr.shape[0] else arr.itemsize, 1)
                chunks = (max(1, min(arr.shape[0], (1 << 20) // row)),) + tuple(arr.shape[1:])
            cd = (hsize_t * arr.ndim)(*[max(int(c), 1) for c in chunks])
            lib.H5Pset_chunk(dcpl, arr.ndim, cd)
            if compression:
                level = compression if isinstance(compression, int) and not isinstance(compression, bool) else (4 if compression_opts is None else int(compression_opts))
                lib.H5Pset_deflate(dcpl, level)
        did = lib.H5Dcreate2(self._file._fid, path.encode(), tid, sid, lcpl, dcpl, H5P_DEFAULT)
        try:
            _check(did, f"create dataset {path}")
            if arr.size:
                buf = arr.astype(np.int8) if arr.dtype == np.bool_ else arr
                _check(lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.ctypes.data_as(C.c_void_p)), f"write {path}")
        finally:
            if did >= 0:
                lib.H5Dclose(did)
            lib.H5Sclose(sid); lib.H5Pclose(lcpl)
            if dcpl:
                lib.H5Pclose(dcpl)
            if own:
                lib.H5Tclose(tid)
        return Dataset(self._file, path)

    def __setitem__(self, name, obj):
        self._writable()
        if isinstance(obj, _Node):      # another name for the same object (record.py clean_trajectories: h5[new] = h5[old]; del h5[old])
            lib, path = _need(), self._abs(name)
            lcpl = lib.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
            lib.H5Pset_create_intermediate_group(lcpl, 1)
            r = lib.H5Lcreate_hard(obj.file._fid, obj.name.encode(), self._file._fid, path.encode(), lcpl, H5P_DEFAULT)
            lib.H5Pclose(lcpl)
            _check(r, f"link {path} -> {obj.name}")
        else:
            self.create_dataset(name, data=obj)

    def __delitem__(self, name):
        self._writable()
        path = self._abs(name)
        if not self._exists(path):
            raise KeyError(name)
        _check(_need().H5Ldelete(self._file._fid, path.encode(), H5P_DEFAULT), f"delete {path}")

    def visititems(self, func):
        def walk(group, prefix):
            for k in group.keys():
                obj = group[k]
                rel = f"{prefix}{k}"
                r = func(rel, obj)
                if r is not None:
                    return r
                if isinstance(obj, Group):
                    r = walk(obj, rel + "/")
                    if r is not None:
                        return r
            return None
        return walk(self, "")

    def visit(self, func):
        return self.visititems(lambda name, obj: func(name))

    def copy(self, source, dest, name=None):
        """h5py's Group.copy for the merge tool (trajectory/merge_trajectory.py): a deep copy of ``source`` under ``dest``"""
        src = self[source] if isinstance(source, str) else source
        dst_group = dest if isinstance(dest, Group) else self.require_group(dest)
        name = name or src.name.rsplit("/", 1)[-1]
        if isinstance(src, Dataset):
            d = dst_group.create_dataset(name, data=src._read(), compression=src.compression)
            for k, v in src.attrs.items():
                d.attrs[k] = v
            return
        g = dst_group.create_group(name, track_order=True)
        for k, v in src.attrs.items():
            g.attrs[k] = v
        for k in src.keys():
            g.copy(src[k], g, k)

    def __repr__(self):
        return f'<HDF5 group "{self._path}" ({len(self)} members)>'


class File(Group):
    """``h5py.File(name, mode)``: 'r' read-only, 'r+' read / write, 'w' create / truncate, 'w-' / 'x' create or fail, 'a' read / write / create."""

    def __init__(self, name, mode="r", **_unused):
        lib = _need()
        name = os.fspath(name)
        self.filename, self.mode = name, mode
        self._fid = None
        b = name.encode()
        if mode == "r":
            fid = lib.H5Fopen(b, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "r+":
            fid = lib.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT)
        elif mode == "w":
            fid = lib.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode in ("w-", "x"):
            fid = lib.H5Fcreate(b, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "a":
            fid = lib.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(name) else lib.H5Fcreate(b, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError(f"invalid mode {mode!r}")
        if fid < 0:
            raise OSError(f"Unable to open file {name!r} in mode {mode!r} (missing, not an HDF5 file, or not writable)")
        self._fid = fid
        super().__init__(self, "/")

    def flush(self):
        if self._fid is not None:
            _need().H5Fflush(self._fid, 1)

    def close(self):
        if self._fid is not None:
            _need().H5Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __bool__(self):
        return self._fid is not None

    def __repr__(self):
        return f'<HDF5 file "{os.path.basename(self.filename)}" (mode {self.mode})>' if self._fid is not None else "<Closed HDF5 file>"
