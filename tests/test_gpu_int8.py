"""Round 5's int8 matrix-pipe paths (cnmf_e_amd/csrc/gram_i8.hpp, win_proj_i8.hpp) and the temporal projection's read-order copy (vproj.hip, k_tile_video)
against the fp64 kernels they replace, on one engine and the same uploads: options gram_i8 / win_i8 / proj_i8 (vproj_i8.hpp: the temporal projection on the int8 pipe) 1
(defaults) vs 0; and the read-order copy of the fp64 temporal projection (proj_tiled, what runs when proj_i8 is off) against the frame-major video.

The int8 paths accumulate EXACTLY (int32) over a 32-bit fixed-point quantisation of the data, so they differ from the fp64 kernels only by that quantisation:
1e-8 of W in the CPU emulation (scripts/probes/gram_i8_emulation.py), a last-bit flip of the fp32 weights on the GPU.  Through the default settings of every other
parity test they are also checked against the float64 oracle; the sizes here are the awkward ones (frame counts off every padding, patches inside blocks, frame
stride 2, rings of 120 offsets, no footprints, crowded lists)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

from parity_util import rel


def _case(d1, d2, T, K, r, seed, pdims=None, min_sep=5):
    from cnmf_e_amd import synth
    f = synth.make_factors(d1, d2, T, K, seed, gSig=1.5, gSiz=7, min_sep=min_sep)
    return f, synth.make_video(f, np.float32)


def _run(opts, f, Y, d1, d2, T, r, pdims, iters=2, alg="hals"):
    """two full iterations with the given option values on a FRESH engine (the tables and copies belong to an upload); returns W per patch, A, C and the kernel names that ran"""
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    eng = Engine(0)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
        video.upload_from_full(Y)
        s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=alg, maxIter=3), f.A_init, f.C_init, f.sn)
        eng.profile(True)
        for _ in range(iters):
            s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        W = [eng.ring_csr(video.pid[idx]).data.astype(np.float64) for idx in video.owned]
        A = s.A.toarray().astype(np.float64); C = np.asarray(s.C, dtype=np.float64).copy()
        names = {k for k, v in eng.profile_table().items() if v["calls"]}
        return W, A, C, names
    finally:
        eng.close()


@pytest.mark.parametrize("dims,T,r,K,pdims", [((70, 50), 203, 15, 8, None),          # T off every padding (64-frame steps, 16-frame stages, 4-frame chunks)
                                              ((64, 60), 96, 15, 6, [32, 30]),       # 2 x 2 patches inside blocks with halo
                                              ((75, 66), 130, 18, 9, None),          # radius 18: 120 offsets, block displacement 3
                                              ((48, 44), 1000, 5, 30, None)])        # crowded: lists of more than 32 traces per block (two trace groups per item)
def test_int8_paths_equal_the_fp64_kernels(dims, T, r, K, pdims):
    d1, d2 = dims
    f, Y = _case(d1, d2, T, K, r, 23, min_sep=3 if K > 20 else 5)
    W1, A1, C1, n1 = _run({"win_i8_planes": 4}, f, Y, d1, d2, T, r, pdims)   # (all four digit planes in the window projection: the int8 pipe's sums are then the fp64 pipe's up to 2^-32; the default reads three, below)
    W0, A0, C0, n0 = _run({"gram_i8": 0, "win_i8": 0, "proj_i8": 0, "proj_tiled": 0}, f, Y, d1, d2, T, r, pdims)
    assert "bg_gram_i8" in n1 and "bg_trace_gram" in n1 and "temporal_dig_pixmajor" in n1 and "temporal_panel_dig" in n1, n1      # the new paths really ran ...
    assert "bg_gram_f64" in n0 and "bg_gram_i8" not in n0 and "bg_trace_gram" not in n0 and "temporal_tile_video" not in n0 and "temporal_panel_dig" not in n0, n0      # ... and really did not
    for a, b in zip(W1, W0):
        assert np.all(np.isfinite(a))
        assert rel(a, b) <= 5e-7, rel(a, b)                      # observed 2e-8 .. 9e-8 (last-bit flips of the fp32 weights)
    assert np.array_equal(A1 != 0, A0 != 0)
    assert rel(A1, A0) <= 2e-6 and rel(C1, C0) <= 2e-6, (rel(A1, A0), rel(C1, C0))
    # the fp64 temporal projection on its read-order copy of the video (what a patch without resident digit planes runs)
    W2, A2, C2, n2 = _run({"proj_i8": 0, "win_i8_planes": 4}, f, Y, d1, d2, T, r, pdims)
    assert "temporal_tile_video" in n2 and "temporal_panel_dig" not in n2, n2
    assert rel(A2, A0) <= 2e-6 and rel(C2, C0) <= 2e-6, (rel(A2, A0), rel(C2, C0))


def test_int8_table_of_a_strided_fit_and_without_footprints():
    """frame stride 2 (fit_ring_model.m:84-87: the digit planes are built from every second frame and are NOT kept for the window projection) and a fit without any
    footprint (K = 0: the table alone)"""
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2, T, r = 44, 40, 1300, 5
    f, Y = _case(d1, d2, T, 4, r, 41)
    res = {}
    for i8 in (1, 0):
        eng = Engine(0)
        try:
            eng.set_option("gram_i8", i8)
            video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
            video.upload_from_full(Y)
            eng.ring_init(0, r)
            Wcsr = eng.ring_csr(0)
            v = -np.ones(Wcsr.nnz, np.float32) * 0.01                # 6 positive weights per row: pmax = 6 -> nk = 600 -> stride 2
            for i in range(Wcsr.shape[0]):
                a, b = Wcsr.indptr[i], Wcsr.indptr[i + 1]
                v[a:a + min(6, b - a)] = 0.02
            eng.ring_set_values(0, v)
            _, info = eng.fit_ring_model(0, f.A_init.tocsc().astype(np.float32), np.ascontiguousarray(f.C_init, dtype=np.float32))
            assert info["frame_stride"] == 2, info
            w2 = eng.ring_csr(0).data.astype(np.float64)
            eng.ring_init(0, r)
            _, info = eng.fit_ring_model(0, None, None)
            res[i8] = (w2, eng.ring_csr(0).data.astype(np.float64))
        finally:
            eng.close()
    for a, b in zip(res[1], res[0]):
        assert np.all(np.isfinite(a)) and rel(a, b) <= 5e-7, rel(a, b)


def test_int8_projections_of_a_recording_longer_than_one_int32_segment():
    """T = 26000 > 24576 frames, frame stride 9 in the fit (fit_ring_model.m:84-87): the digit planes of EVERY frame feed the spatial update's table -- its sums run
    over frames and are kept inside frame segments of at most 24576 frames -- and the temporal projection (sums over pixels); both against the fp64 kernels on the
    same upload"""
    d1, d2, T, r, K = 36, 32, 26000, 5, 4
    f, Y = _case(d1, d2, T, K, r, 29)
    W1, A1, C1, n1 = _run({}, f, Y, d1, d2, T, r, None)
    W0, A0, C0, n0 = _run({"win_i8": 0, "proj_i8": 0}, f, Y, d1, d2, T, r, None)
    assert "temporal_panel_dig" in n1 and "spatial_trace_dig" in n1, n1          # the int8 table of the spatial update and the int8 temporal projection ran ...
    assert "temporal_panel_dig" not in n0 and "spatial_trace_dig" not in n0, n0  # ... and did not
    for a, b in zip(W1, W0):
        assert np.all(np.isfinite(a)) and rel(a, b) <= 5e-7, rel(a, b)
    assert np.array_equal(A1 != 0, A0 != 0)
    assert rel(A1, A0) <= 2e-6 and rel(C1, C0) <= 2e-6, (rel(A1, A0), rel(C1, C0))


def test_window_projection_on_three_digit_planes_is_the_default_with_a_stated_cost():
    """win_i8_planes = 3 (the default since the end of round 6): the fit's window projection reads 24-bit samples of the video.  Against all four planes (= the fp64 kernels
    up to 2^-32; recordings of at least 2048 frames: shorter ones keep four) the weights move by a few 1e-6 of the largest one at most -- the oracle parity tests keep their 2e-6, SURVEY.md 8(c) asks 1e-3 -- and A, C follow within 1e-5"""
    d1, d2, T, r, K = 64, 60, 2400, 15, 6                       # (at least 2048 frames: below that the default keeps four planes)
    f, Y = _case(d1, d2, T, K, r, 31)
    W4, A4, C4, n4 = _run({"win_i8_planes": 4}, f, Y, d1, d2, T, r, None)
    W3, A3, C3, n3 = _run({}, f, Y, d1, d2, T, r, None)
    assert "bg_win_proj" in n3 and "bg_win_proj" in n4
    worst = max(rel(a, b) for a, b in zip(W3, W4))
    assert 0 < worst <= 5e-6, worst                     # (it IS a different sum: equal arrays would mean the default reads four planes)
    assert np.array_equal(A3 != 0, A4 != 0)
    assert rel(A3, A4) <= 1e-5 and rel(C3, C4) <= 1e-5, (rel(A3, A4), rel(C3, C4))
