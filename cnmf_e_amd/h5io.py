"""Read-only access to HDF5 files through the system's libhdf5 (ctypes; h5py is not in this image).

What the data plane needs from the three HDF5-based inputs of the reference:
  * the blocked `data_<p1>_<p2>_<w>.mat` file distribute_data.m:127-173 writes with `-v7.3` (datasets `Y_r0_r1_c0_c1`, the index
    vectors and `dims`) and get_patch_data.m:50-93 reads block by block;
  * a `.h5` / `.hdf5` recording with one dataset in the root group (smod_bigread2.m:338-355, get_data_dimension.m:32-35);
  * a v7.3 `.mat` recording holding `Y` (+ `Ysiz`) or one array (smod_bigread2.m:378-400).
A MATLAB array of size [a b c] is an HDF5 dataset of dims (c, b, a): the same bytes, MATLAB column-major = HDF5 row-major of the reversed
dims; `h5read` reverses the dims of a foreign file the same way.  So a frame range of a d1 x d2 x T array is a hyperslab over the FIRST HDF5
dimension and arrives as (frames, d2, d1) -- every frame already in the reference's pixel order (pixel = (c-1) d1 + r).

Only what is needed is bound: open, list the root group, shape / element type / MATLAB_class of a dataset, read a hyperslab.  libhdf5 is looked
up in CNMFE_HDF5_LIB, the loader path, then the usual prefixes; a missing library is an error here, not a fallback.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import glob
import os

import numpy as np

_L = None
_hid = ctypes.c_int64
_hsize = ctypes.c_uint64

_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3


def _candidates():
    env = os.environ.get("CNMFE_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*", "/opt/hdf5*/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def lib():
    """the loaded libhdf5 with prototypes set; raises RuntimeError when there is none"""
    global _L, _hid
    if _L is not None:
        return _L
    tried, L = [], None
    for p in _candidates():
        try:
            L = ctypes.CDLL(p)
            break
        except OSError as e:
            tried.append("%s (%s)" % (p, e))
    if L is None:
        raise RuntimeError("libhdf5 not found: set CNMFE_HDF5_LIB to the shared library (tried: %s)" % ("; ".join(tried) or "loader path and usual prefixes"))
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    L.H5open()
    L.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        _hid = ctypes.c_int                                               # hid_t was an int before 1.10
    L.version = (maj.value, mnr.value, rel.value)
    P, I, S = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    protos = {
        "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]), "H5Fclose": (I, [_hid]),
        "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]), "H5Dclose": (I, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (I, [_hid, _hid, _hid, _hid, _hid, P]),
        "H5Sget_simple_extent_ndims": (I, [_hid]), "H5Sget_simple_extent_dims": (I, [_hid, P, P]),
        "H5Screate_simple": (_hid, [I, P, P]), "H5Sselect_hyperslab": (I, [_hid, I, P, P, P, P]), "H5Sclose": (I, [_hid]),
        "H5Tget_class": (I, [_hid]), "H5Tget_size": (S, [_hid]), "H5Tget_sign": (I, [_hid]), "H5Tget_native_type": (_hid, [_hid, I]),
        "H5Tclose": (I, [_hid]),
        "H5Aexists": (I, [_hid, ctypes.c_char_p]), "H5Aopen": (_hid, [_hid, ctypes.c_char_p, _hid]), "H5Aget_type": (_hid, [_hid]),
        "H5Aread": (I, [_hid, _hid, P]), "H5Aclose": (I, [_hid]),
        "H5Eset_auto2": (I, [_hid, P, P]),
    }
    for name, (res, args) in protos.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.iter_cb = ctypes.CFUNCTYPE(ctypes.c_int, _hid, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)
    # from libhdf5 1.12 on H5Literate is a macro: the library exports H5Literate1 (this callback signature) and H5Literate2 (H5L_info2_t, which
    # the callback never looks at) instead
    it = None
    for name in ("H5Literate", "H5Literate1", "H5Literate2"):
        try:
            it = getattr(L, name)
            break
        except AttributeError:
            continue
    if it is None:
        raise RuntimeError("libhdf5 %d.%d.%d exports none of H5Literate, H5Literate1, H5Literate2" % L.version)
    it.restype = I
    it.argtypes = [_hid, I, I, P, L.iter_cb, P]
    L.iterate = it
    L.H5Eset_auto2(0, None, None)                                         # errors come back as return codes and are raised below, not printed
    _L = L
    return L


def is_hdf5(path):
    """the HDF5 signature sits at byte 0, 512, 1024, ... (a MATLAB v7.3 file keeps its text header in a 512-byte user block)"""
    sig = b"\x89HDF\r\n\x1a\n"
    with open(path, "rb") as f:
        off = 0
        size = os.fstat(f.fileno()).st_size
        while off + 8 <= size and off <= (1 << 20):
            f.seek(off)
            if f.read(8) == sig:
                return True
            off = 512 if off == 0 else off * 2
    return False


class H5File:
    """one file, read-only.  Dataset handles stay open until close() (a block is read in many frame slabs)."""

    def __init__(self, path):
        self.L = lib()
        self.path = os.fspath(path)
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        self.fid = self.L.H5Fopen(self.path.encode(), 0, 0)               # H5F_ACC_RDONLY, H5P_DEFAULT
        if self.fid < 0:
            raise OSError("%s is not an HDF5 file libhdf5 %d.%d.%d can open (a MATLAB file must have been saved with -v7.3)" % ((self.path,) + self.L.version))
        self._dsets = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        for d, s, t, _, _ in self._dsets.values():
            self.L.H5Tclose(t); self.L.H5Sclose(s); self.L.H5Dclose(d)
        self._dsets = {}
        if self.fid >= 0:
            self.L.H5Fclose(self.fid)
            self.fid = -1

    def names(self):
        """links of the root group in name order (groups such as MATLAB's #refs# included; `has` tells the datasets apart)"""
        out = []

        def cb(_g, name, _info, _data):
            out.append(name.decode())
            return 0
        keep = self.L.iter_cb(cb)
        if self.L.iterate(self.fid, 0, 0, None, keep, None) < 0:        # H5_INDEX_NAME, H5_ITER_INC
            raise OSError("cannot list %s" % self.path)
        return out

    def _open(self, name):
        e = self._dsets.get(name)
        if e is None:
            L = self.L
            d = L.H5Dopen2(self.fid, name.encode(), 0)
            if d < 0:
                raise KeyError("%s has no dataset %r" % (self.path, name))
            s, ft = L.H5Dget_space(d), L.H5Dget_type(d)
            nd = L.H5Sget_simple_extent_ndims(s)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(s, dims, None)
            cls, size = L.H5Tget_class(ft), L.H5Tget_size(ft)
            if cls == _H5T_INTEGER:
                dt = np.dtype("%s%d" % ("i" if L.H5Tget_sign(ft) == 1 else "u", size))
            elif cls == _H5T_FLOAT:
                dt = np.dtype("f%d" % size)
            else:
                dt = None                                                 # references (cell arrays), strings, compounds: not read here
            t = L.H5Tget_native_type(ft, 1)                               # H5T_DIR_ASCEND: the in-memory type of this machine (byte order converted by the library)
            L.H5Tclose(ft)
            e = self._dsets[name] = (d, s, t, tuple(int(v) for v in dims[:nd]), dt)
        return e

    def has(self, name):
        try:
            self._open(name)
            return True
        except KeyError:
            return False

    def shape(self, name):
        """HDF5 dims (slowest first) = the MATLAB size reversed"""
        return self._open(name)[3]

    def dtype(self, name):
        dt = self._open(name)[4]
        if dt is None:
            raise TypeError("dataset %r of %s is neither integer nor floating point" % (name, self.path))
        return dt

    def matlab_class(self, name):
        """the MATLAB_class attribute of a v7.3 variable ('uint16', 'single', 'char', 'cell', ...), None for a plain HDF5 dataset"""
        L = self.L
        d = self._open(name)[0]
        if L.H5Aexists(d, b"MATLAB_class") <= 0:
            return None
        a = L.H5Aopen(d, b"MATLAB_class", 0)
        t = L.H5Aget_type(a)
        try:
            if L.H5Tget_class(t) != _H5T_STRING:
                return None
            buf = ctypes.create_string_buffer(L.H5Tget_size(t) + 1)
            if L.H5Aread(a, t, buf) < 0:
                return None
            return buf.value.decode()
        finally:
            L.H5Tclose(t); L.H5Aclose(a)

    def read(self, name, start=None, count=None):
        """the hyperslab [start, start + count) of a dataset as a C-ordered array of HDF5 dims (whole dataset by default)"""
        L = self.L
        d, s, t, dims, dt = self._open(name)
        if dt is None:
            raise TypeError("dataset %r of %s is neither integer nor floating point" % (name, self.path))
        nd = len(dims)
        start = tuple(0 for _ in dims) if start is None else tuple(int(v) for v in start)
        count = tuple(n - a for n, a in zip(dims, start)) if count is None else tuple(int(v) for v in count)
        if len(start) != nd or len(count) != nd or any(a < 0 or c < 0 or a + c > n for a, c, n in zip(start, count, dims)):
            raise IndexError("hyperslab %s + %s outside dataset %r of dims %s" % (start, count, name, dims))
        out = np.empty(count, dtype=dt)
        if out.size == 0:
            return out
        if nd == 0:
            if L.H5Dread(d, t, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError("H5Dread failed on %r" % name)
            return out
        st, ct = (_hsize * nd)(*start), (_hsize * nd)(*count)
        if L.H5Sselect_hyperslab(s, 0, st, None, ct, None) < 0:           # H5S_SELECT_SET
            raise OSError("H5Sselect_hyperslab failed on %r" % name)
        ms = L.H5Screate_simple(nd, ct, None)
        try:
            if L.H5Dread(d, t, ms, s, 0, out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError("H5Dread failed on %r of %s (a filter this libhdf5 was built without?)" % (name, self.path))
        finally:
            L.H5Sclose(ms)
        return out

    # -- MATLAB views ------------------------------------------------------------------
    def matlab_value(self, name):
        """a small numeric / char variable the way MATLAB shows it: numeric -> array of MATLAB's size (column-major data), char -> str"""
        a = self.read(name)
        if self.matlab_class(name) == "char":
            return "".join(chr(int(v)) for v in a.T.reshape(-1, order="F"))
        return a.T                                                        # reversed dims, same memory: MATLAB's array

    def matlab_size(self, name):
        return tuple(reversed(self.shape(name)))
