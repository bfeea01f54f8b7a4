"""HDF5 side of the data plane: cnmf_e_amd/h5io.py (ctypes over libhdf5) and the PatchedVideo readers for the reference's blocked
`mat_data` file (distribute_data.m:127-173 / get_patch_data.m:50-93), `.h5` recordings (smod_bigread2.m:338-355) and v7.3 `.mat` recordings
(:378-400).  The fixtures under tests/golden/h5_* were written by h5py (tests/golden/make_h5_fixtures.py), not by the code under test."""
import os

import numpy as np
import pytest

from cnmf_e_amd import h5io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

try:
    h5io.lib()
except RuntimeError as e:                                                  # no libhdf5 on this machine: the readers raise the same error when used
    pytest.skip(str(e), allow_module_level=True)


def video():
    return np.load(os.path.join(GOLD, "h5_video.npy"))                     # d1 x d2 x T as MATLAB holds it


def test_reader_on_a_v73_file():
    """names, MATLAB views of numeric / char variables, class attribute, cell arrays are told apart, hyperslabs, error paths"""
    Y = video()
    with h5io.H5File(os.path.join(GOLD, "h5_mat_data.mat")) as f:
        names = f.names()
        assert "dims" in names and "#refs#" in names and "Y_1_4_1_4" in names
        assert f.matlab_value("dims").tolist() == [[32.0, 28.0, 30.0]] and f.matlab_class("dims") == "double"
        assert f.matlab_value("block_idx_r").shape == (6, 1)               # block_idx_r(:) is a column in MATLAB
        assert f.matlab_value("dtype") == "uint16" and f.matlab_value("file_name") == "/data/recording.tif"
        assert f.matlab_class("patch_pos") == "cell" and not f.has("#refs#") and not f.has("no_such_variable")
        with pytest.raises(TypeError):
            f.dtype("patch_pos")
        with pytest.raises(KeyError):
            f.shape("no_such_variable")
        n = "Y_13_20_11_18"
        assert f.shape(n) == (30, 8, 8) and f.matlab_size(n) == (8, 8, 30) and f.dtype(n) == np.uint16 and f.matlab_class(n) == "uint16"
        assert np.array_equal(f.read(n).T, Y[12:20, 10:18, :])             # whole block: reversed dims, same memory as MATLAB's array
        part = f.read(n, (4, 2, 1), (7, 3, 5))                             # frames 5..11, columns 13..15, rows 14..18 (deflate chunks of 8 frames are crossed)
        assert np.array_equal(part, Y[13:18, 12:15, 4:11].T)
        assert f.read(n, (0, 0, 0), (0, 8, 8)).shape == (0, 8, 8)
        with pytest.raises(IndexError):
            f.read(n, (25, 0, 0), (6, 8, 8))
    assert h5io.is_hdf5(os.path.join(GOLD, "h5_mat_data.mat")) and h5io.is_hdf5(os.path.join(GOLD, "h5_recording.h5"))
    assert not h5io.is_hdf5(os.path.join(GOLD, "h5_video.npy"))
    with pytest.raises(OSError):
        h5io.H5File(os.path.join(GOLD, "h5_video.npy"))
    with pytest.raises(FileNotFoundError):
        h5io.H5File(os.path.join(GOLD, "missing.mat"))


def test_mat_data_file_geometry_is_the_one_the_host_mirror_computes():
    """the cut lines h5py wrote from its own restatement of distribute_data.m:56-110 = storage_block_index of the patch edges of distribute_geometry;
    every stored block is the video's rectangle"""
    from cnmf_e_amd.sources2d import distribute_geometry, mat_data_info, storage_block_index
    Y = video()
    info = mat_data_info(os.path.join(GOLD, "h5_mat_data.mat"))
    assert info["dims"] == (32, 28, 30) and info["patch_dims"] == (16, 14) and info["w_overlap"] == 3 and info["dtype"] == np.uint16
    (nrp, ncp), patch_pos, block_pos = distribute_geometry(32, 28, info["patch_dims"], info["w_overlap"])
    pr = [int(patch_pos[(m, 0)][0]) for m in range(nrp)] + [32]
    pc = [int(patch_pos[(0, n)][2]) for n in range(ncp)] + [28]
    assert np.array_equal(storage_block_index(32, pr, 3), info["block_idx_r"]) and np.array_equal(storage_block_index(28, pc, 3), info["block_idx_c"])
    br, bc = info["block_idx_r"], info["block_idx_c"]
    assert set(info["blocks"]) == {(br[m], br[m + 1], bc[n], bc[n + 1]) for m in range(len(br) - 1) for n in range(len(bc) - 1)}
    with h5io.H5File(os.path.join(GOLD, "h5_mat_data.mat")) as f:
        for (r0, r1, c0, c1), name in info["blocks"].items():
            assert np.array_equal(f.read(name).T, Y[r0 - 1:r1, c0 - 1:c1, :]), name
    # every block with halo lies inside the union of the storage blocks get_patch_data.m:50-60 picks for its patch
    for idx, b in block_pos.items():
        p = patch_pos[idx]
        lo_r, hi_r = br[br <= p[0]].max(), br[br >= p[1]].min()
        lo_c, hi_c = bc[bc <= p[2]].max(), bc[bc >= p[3]].min()
        assert (lo_r, hi_r, lo_c, hi_c) == tuple(int(v) for v in b), idx


def _blocks(loader, T=30, rank=0, world=1):
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo
    eng = FakeEngine()
    v = PatchedVideo(32, 28, T, [16, 14], 3, eng, rank=rank, world_size=world)
    loader(v)
    return {idx: eng.p[v.pid[idx]]["Y"].copy() for idx in v.owned}


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2)])
def test_upload_from_mat_data_matches_upload_from_full(rank, world):
    """blocks put together from the stored blocks = blocks cut from the whole video, for all frames and for a frame range, on every rank's share"""
    Y = video()
    Y_td = np.ascontiguousarray(Y.reshape(32 * 28, 30, order="F").T)      # T x d, pixels column-major
    path = os.path.join(GOLD, "h5_mat_data.mat")
    ref = _blocks(lambda v: v.upload_from_full(Y_td, chunk=30), rank=rank, world=world)
    got = _blocks(lambda v: v.upload_from_mat_data(path, chunk=7), rank=rank, world=world)
    assert set(got) == set(ref) and len(ref) == (4 if world == 1 else 2)
    for idx in ref:
        assert np.array_equal(got[idx], ref[idx]), idx
    ref = _blocks(lambda v: v.upload_from_full(Y_td[6:26], chunk=30), T=20, rank=rank, world=world)
    got = _blocks(lambda v: v.upload_from_mat_data(path, chunk=64, frame0=6), T=20, rank=rank, world=world)
    for idx in ref:
        assert np.array_equal(got[idx], ref[idx]), idx
    with pytest.raises(ValueError):
        _blocks(lambda v: v.upload_from_mat_data(path, frame0=11), T=20)   # frames 12..31 of 30
    with pytest.raises(ValueError):
        _blocks(lambda v: v.upload_from_mat_data(os.path.join(GOLD, "h5_recording.h5")))      # no Y_r0_r1_c0_c1 block in it


def test_upload_from_hdf5_recordings():
    """a big-endian float32 .h5 movie, the 5-D int16 layout, a v7.3 .mat with Y + Ysiz: every block gets the frames upload_from_full would give it"""
    Y = video()
    Y_td = np.ascontiguousarray(Y.reshape(32 * 28, 30, order="F").T)
    ref = _blocks(lambda v: v.upload_from_full(Y_td, chunk=30))
    for fn, expect, kw in (("h5_recording.h5", ref, {}), ("h5_recording.h5", ref, {"dataset": "mov", "chunk": 4}), ("h5_recording5d.hdf5", ref, {"chunk": 11}),
                           ("h5_recording.mat", {k: a // 16 for k, a in ref.items()}, {"chunk": 9})):
        got = _blocks(lambda v: v.upload_from_hdf5(os.path.join(GOLD, fn), **kw))
        for idx in expect:
            assert np.array_equal(np.asarray(got[idx], dtype=np.float64), np.asarray(expect[idx], dtype=np.float64)), (fn, idx)
    got = _blocks(lambda v: v.upload_from_hdf5(os.path.join(GOLD, "h5_recording.h5"), frame0=10, chunk=6), T=15)
    want = _blocks(lambda v: v.upload_from_full(Y_td[10:25], chunk=30), T=15)
    for idx in want:
        assert np.array_equal(np.asarray(got[idx], dtype=np.float64), np.asarray(want[idx], dtype=np.float64)), idx
    with pytest.raises(ValueError):
        _blocks(lambda v: v.upload_from_hdf5(os.path.join(GOLD, "h5_mat_data.mat")))          # 20 candidate datasets: the caller has to name one
    with pytest.raises(ValueError):
        _blocks(lambda v: v.upload_from_hdf5(os.path.join(GOLD, "h5_mat_data.mat"), dataset="Y_1_4_1_4"))   # not the field of view
    with pytest.raises(KeyError):
        _blocks(lambda v: v.upload_from_hdf5(os.path.join(GOLD, "h5_recording.h5"), dataset="Y"))
