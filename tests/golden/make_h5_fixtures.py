"""Writes the HDF5 fixtures of tests/test_h5io.py with h5py -- an HDF5 writer independent of cnmf_e_amd/h5io.py (ctypes over libhdf5).

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py        (the image's conda interpreter has h5py 3.3 / HDF5 1.10.6; the system one has not)

Files (all of one seeded 32 x 28 x 30 recording, kept as h5_video.npy in MATLAB's d1 x d2 x T order):
  h5_mat_data.mat     the blocked file of endoscope/distribute_data.m:127-173 for patch_dims [16 14], w_overlap 3, laid out the way MATLAB's
                      `-v7.3` writer does: 512-byte user block with the text header, every variable a dataset of REVERSED dims with a MATLAB_class
                      attribute, char data as uint16 codes, cell arrays as object references into /#refs#, the Y_r0_r1_c0_c1 blocks chunked + deflate.
  h5_recording.h5     one dataset in the root group, dims (T, d2, d1) as h5read expects for a d1 x d2 x T movie; big-endian float32, chunked.
  h5_recording5d.hdf5 the 5-D layout get_data_dimension.m:32-35 indexes with [2 3 5]: MATLAB size [1 d1 d2 1 T], contiguous int16.
  h5_recording.mat    v7.3 recording with Y (uint8) and Ysiz (smod_bigread2.m:378-400).
The block geometry is restated here from distribute_data.m:56-110 on its own, so the test also checks cnmf_e_amd.sources2d against it.
"""
import math
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
D1, D2, T = 32, 28, 30
PATCH, W = (16, 14), 3


def matlab_header(f_path):
    """the 512-byte user block MATLAB puts in front of the HDF5 superblock"""
    text = b"MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Sun Sep 27 12:00:00 2026 HDF5 schema 1.00 ."
    head = text + b" " * (116 - len(text)) + b"\x00" * 8 + b"\x00\x02" + b"IM"
    with open(f_path, "r+b") as f:
        f.write(head + b"\x00" * (512 - len(head)))


def put(f, name, value, cls, **kw):
    """MATLAB array `value` (numpy, MATLAB's shape) -> dataset of reversed dims"""
    a = np.asarray(value)
    if a.ndim < 2:
        a = a.reshape(1, -1) if a.ndim == 1 else a.reshape(1, 1)
    d = f.create_dataset(name, data=np.ascontiguousarray(a.T), **kw)       # a.T: reversed dims, same column-major memory
    d.attrs.create("MATLAB_class", np.bytes_(cls))
    return d


def put_char(f, name, s):
    d = put(f, name, np.array([ord(c) for c in s], dtype=np.uint16).reshape(1, -1), "char")
    d.attrs.create("MATLAB_int_decode", np.int32(2))


def edges(d, n, last_fix, minw):
    """patch_idx_r / patch_idx_c, distribute_data.m:56-79"""
    if n <= 1:
        return [1, d]
    e = [int(math.ceil(1 + (d - 1) * i / n)) for i in range(n + 1)]
    if last_fix:
        e[-1] = d
    if e[1] - e[0] < minw:
        e = list(range(1, d + 1, minw)); e[-1] = d
    return e


def cuts(d, pidx, w):
    """block_idx_r / block_idx_c, distribute_data.m:91-100"""
    v = [min(max(p - 1 - w, 1), d) for p in pidx] + [min(max(p + w, 1), d) for p in pidx]
    return sorted(set(v))


def main():
    rng = np.random.default_rng(7)
    Y = rng.integers(0, 4096, size=(D1, D2, T), dtype=np.uint16)           # MATLAB order d1 x d2 x T
    np.save(os.path.join(HERE, "h5_video.npy"), Y)

    minw = 2 * W + 3
    nrp, ncp = int(math.floor(D1 / PATCH[0] + 0.5)), int(math.floor(D2 / PATCH[1] + 0.5))
    pr, pc = edges(D1, nrp, True, minw), edges(D2, ncp, False, minw)
    br, bc = cuts(D1, pr, W), cuts(D2, pc, W)
    path = os.path.join(HERE, "h5_mat_data.mat")
    with h5py.File(path, "w", userblock_size=512, libver="earliest") as f:
        put_char(f, "file_name", "/data/recording.tif")
        put(f, "patch_idx_r", np.array(pr, dtype=np.float64), "double")
        put(f, "patch_idx_c", np.array(pc, dtype=np.float64), "double")
        put(f, "nr_patch", np.float64(len(pr) - 1), "double")
        put(f, "nc_patch", np.float64(len(pc) - 1), "double")
        put(f, "block_idx_r", np.array(br, dtype=np.float64).reshape(-1, 1), "double")       # block_idx_r(:) is a column
        put(f, "block_idx_c", np.array(bc, dtype=np.float64).reshape(-1, 1), "double")
        put(f, "nr_block", np.float64(len(br) - 1), "double")
        put(f, "nc_block", np.float64(len(bc) - 1), "double")
        put(f, "w_overlap", np.float64(W), "double")
        put(f, "patch_dims", np.array(PATCH, dtype=np.float64), "double")
        put(f, "dims", np.array([D1, D2, T], dtype=np.float64), "double")
        put_char(f, "dtype", "uint16")
        for m in range(len(br) - 1):
            for n in range(len(bc) - 1):
                r0, r1, c0, c1 = br[m], br[m + 1], bc[n], bc[n + 1]
                blk = Y[r0 - 1:r1, c0 - 1:c1, :]
                put(f, "Y_%d_%d_%d_%d" % (r0, r1, c0, c1), blk, "uint16", chunks=(min(T, 8), blk.shape[1], blk.shape[0]), compression="gzip", compression_opts=3)
        # patch_pos as MATLAB stores a cell: a dataset of object references into /#refs#
        refs = f.create_group("#refs#")
        cell = np.empty((len(pc) - 1, len(pr) - 1), dtype=h5py.ref_dtype)  # reversed dims of the nr_patch x nc_patch cell
        k = 0
        for m in range(len(pr) - 1):
            for n in range(len(pc) - 1):
                pos = [pr[m], pr[m + 1] - (m != len(pr) - 2), pc[n], pc[n + 1] - (n != len(pc) - 2)]
                d = put(refs, "e%d" % k, np.array(pos, dtype=np.float64), "double"); k += 1
                cell[n, m] = d.ref
        d = f.create_dataset("patch_pos", data=cell)
        d.attrs.create("MATLAB_class", np.bytes_("cell"))
    matlab_header(path)

    with h5py.File(os.path.join(HERE, "h5_recording.h5"), "w") as f:
        f.create_dataset("mov", data=np.ascontiguousarray(Y.T).astype(">f4"), chunks=(5, D2, D1))
    with h5py.File(os.path.join(HERE, "h5_recording5d.hdf5"), "w") as f:
        f.create_dataset("images", data=np.ascontiguousarray(Y.reshape(1, D1, D2, 1, T).T).astype(np.int16))
    path = os.path.join(HERE, "h5_recording.mat")
    with h5py.File(path, "w", userblock_size=512) as f:
        put(f, "Y", (Y >> 4).astype(np.uint8), "uint8", chunks=(10, D2, D1), compression="gzip")
        put(f, "Ysiz", np.array([D1, D2, T], dtype=np.float64).reshape(-1, 1), "double")
    matlab_header(path)
    for n in ("h5_video.npy", "h5_mat_data.mat", "h5_recording.h5", "h5_recording5d.hdf5", "h5_recording.mat"):
        print(n, os.path.getsize(os.path.join(HERE, n)))


if __name__ == "__main__":
    main()
