"""The ring solve out of PACKED systems (cnmf_e_amd/csrc/ring_solve_packed.hpp; option solve_packed, default 1) against the table path it replaces
(k_cov_correct + k_ring_solve5, solve_packed = 0) and, through the default settings of every other parity test, against the oracle.

Both paths compute fit_ring_model.m:92-108 from the same fp64 table of the video and the same footprint corrections; the packed path applies the
corrections to a pixel's system in registers instead of sweeping the table first, so the two agree to the rounding of the corrections' sum -- far below the
fp32 storage of W (1e-7)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

from parity_util import rel


@pytest.fixture()
def eng():
    from cnmf_e_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _video(eng, d1, d2, T, K, r, seed, pdims=None, gSig=1.5, gSiz=7, min_sep=5):
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    f = synth.make_factors(d1, d2, T, K, seed, gSig=gSig, gSiz=gSiz, min_sep=min_sep)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    return f, Y, video


def _fits(eng, r, seq, packed):
    eng.set_option("solve_packed", packed)
    eng.ring_init(0, r)
    out = []
    for Ai, Ci in seq:
        _, info = eng.fit_ring_model(0, Ai, Ci)
        out.append((eng.ring_csr(0).data.astype(np.float64), eng.b0(0).astype(np.float64), info["frame_stride"], info["first_run"]))
    return out


@pytest.mark.parametrize("dims,T,r,K", [((40, 36), 600, 5, 5), ((64, 60), 300, 15, 8), ((40, 36), 9200, 5, 5), ((75, 66), 160, 18, 9), ((80, 72), 128, 20, 6),
                                        ((70, 64), 200, 15, 40)])
def test_packed_solve_equals_the_table_path(eng, dims, T, r, K):
    """A sequence of fits with changing A, C on one patch: first run (uniform W_old), footprints scaled, no footprints at all, a subset, the full set again;
    T = 9200 brings in the frame stride 2 of fit_ring_model.m:84-87 (a second packed copy); K = 40 on 70 x 64 puts 10-25 neurons around every ring
    (more than the RSP_NS = 8 staged before the system is loaded: the later rounds); radii 18 / 20 (120 / 124 offsets) are the 8-tile instantiations"""
    d1, d2 = dims
    f, Y, video = _video(eng, d1, d2, T, K, r, 11, min_sep=3 if K > 20 else 5)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    seq = [(A, Cm), (A * 0.8, Cm * 1.2), (None, None), (A[:, :2].tocsc(), Cm[:2]), (A, Cm)]
    try:
        eng.set_option("debug", 1)
        tab, pk = _fits(eng, r, seq, 0), _fits(eng, r, seq, 1)
        for (wt, bt, kt, ft), (wp, bp, kp, fp) in zip(tab, pk):
            assert kt == kp and ft == fp
            assert np.all(np.isfinite(wp))
            assert np.linalg.norm(wp - wt) <= 5e-7 * np.linalg.norm(wt), np.linalg.norm(wp - wt) / np.linalg.norm(wt)
            assert np.array_equal(bt, bp)
        if T > 9000:
            assert max(k for _, _, k, _ in pk) == 2
    finally:
        eng.set_option("debug", 0); eng.set_option("solve_packed", 1)


def test_more_footprints_over_a_pixel_than_the_cache_holds(eng):
    """seven footprints over the same pixels (the kernel caches RSP_CAP = 4 entries of a pixel's row of A in LDS and reads longer rows where they lie),
    among other, disjoint ones"""
    d1, d2, T, r = 48, 44, 200, 5
    f, Y, video = _video(eng, d1, d2, T, 5, r, 7)
    A = f.A_init.tocsc().astype(np.float32)
    base = A[:, 0].toarray().ravel()
    cols = [sp.csc_matrix((base * (1.0 + 0.1 * j))[:, None]) for j in range(7)] + [A[:, j] for j in range(1, 5)]
    A7 = sp.hstack(cols).tocsc().astype(np.float32)
    assert np.diff(A7.tocsr().indptr).max() >= 7
    rng = np.random.default_rng(3)
    C7 = np.ascontiguousarray(np.vstack([f.C_init[:1] * (1 + 0.2 * rng.random((7, 1))) + rng.random((7, T)), f.C_init[1:5]]), dtype=np.float32)
    a = _fits(eng, r, [(A7, C7), (A7 * 0.9, C7)], 0)
    b = _fits(eng, r, [(A7, C7), (A7 * 0.9, C7)], 1)
    eng.set_option("solve_packed", 1)
    for (wt, bt, _, _), (wp, bp, _, _) in zip(a, b):
        assert np.linalg.norm(wp - wt) <= 5e-7 * np.linalg.norm(wt) and np.array_equal(bt, bp)


@pytest.mark.parametrize("pdims,r,ssub", [([32, 32], 5, 1), ([40, 36], 15, 1), ([48, 44], 6, 2)])
def test_packed_iterations_on_patches_against_the_oracle(eng, pdims, r, ssub):
    """2 x 2 patches (patch != block: ring pixels in the halo, neighbours outside the field of view) and the low-resolution patches of bg_ssub = 2, two full
    iterations against the oracle with the packed solve (the default) and against the engine's own table path"""
    import cnmfe_oracle as orc
    from cnmf_e_amd.sources2d import Sources2D, Options
    d1, d2, T, K = 2 * pdims[0] - 6, 2 * pdims[1] - 4, 240, 9
    res = {}
    for packed in (1, 0):
        eng.set_option("solve_packed", packed)
        f, Y, video = _video(eng, d1, d2, T, K, r, 5, pdims=pdims)
        s = Sources2D(video, Options(ring_radius=r, maxIter=3, bg_ssub=ssub), f.A_init, f.C_init, f.sn)
        for _ in range(2):
            s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        res[packed] = (s.A.toarray().copy(), np.array(s.C).copy())
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, pdims, r, f.A_init.astype(np.float32), f.C_init, f.sn, maxIter=3, bg_ssub=ssub)
    for _ in range(2):
        o.update_background_parallel(); o.update_spatial_parallel(); o.update_temporal_parallel()
    eng.set_option("solve_packed", 1)
    assert rel(res[1][0], res[0][0]) <= 2e-6 and rel(res[1][1], res[0][1]) <= 2e-6
    assert rel(res[1][0], o.A.toarray()) <= 5e-6 and rel(res[1][1], o.C) <= 5e-6
