"""The sweep-free ("virtual") residual of round 4 (cnmf_e_amd/csrc/vproj.hip): after a background fit cnmfe_residual only records the
request; cnmfe_update_spatial reads Ysig C' out of the table P = Yc Cc' (U = P - W P) and cnmfe_hals_temporal projects the centred video through
B = A - W'A.  These tests pin that path against the swept residual (option r1_virtual = 0: the engine of rounds 1-3, itself pinned against the oracle) on the
same inputs -- single patches and patches inside blocks (halo neurons: the pending footprint term on top), ring radii 5 / 15 / 18, odd sizes (partial 16 x 16
blocks), the three spatial algorithms, bound and unbound trace matrices, the table left by the fit and the table built by the update -- and against the
oracle at method level (update_spatial_parallel.m:162-216, update_temporal_parallel.m:149-186)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

from parity_util import rel


@pytest.fixture(scope="module")
def eng():
    from cnmf_e_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _setup(eng, d1, d2, T, K, r, seed, pdims=None, gsig=1.5, gsiz=7):
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    f = synth.make_factors(d1, d2, T, K, seed, gSig=gsig, gSiz=gsiz, min_sep=5)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    return f, Y, video


def _dilated_mask(A, d1, d2):
    """the footprints grown by one pixel (3 x 3 box) inside the field of view: a search mask a little larger than the footprint"""
    from scipy.ndimage import binary_dilation
    img = (A.toarray() > 0).reshape(d1, d2, -1, order="F")
    out = np.stack([binary_dilation(img[:, :, k], structure=np.ones((3, 3), bool)) for k in range(img.shape[2])], axis=2)
    return sp.csc_matrix(out.reshape(d1 * d2, -1, order="F")).astype(np.float32)


def _calls(tab, name):
    return tab.get(name, {"calls": 0})["calls"]


@pytest.mark.parametrize("dims,r,T", [((64, 64), 15, 160), ((70, 50), 15, 203), ((44, 40), 5, 96), ((75, 66), 18, 128)])
@pytest.mark.parametrize("bound", [False, True])
def test_virtual_projections_equal_the_swept_ones(eng, dims, r, T, bound):
    """one patch = the field of view, engine level: residual -> update_spatial -> residual -> hals_temporal with and without the sweep.  `bound`: the traces
    are the engine's bound matrix (then the spatial update finds the fit's table), else a host matrix (it builds its own)"""
    d1, d2 = dims
    K = 6
    f, Y, video = _setup(eng, d1, d2, T, K, r, 31)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    IND = _dilated_mask(A, d1, d2)
    prev = np.array([0, 3, 4])
    def run(virtual, alg):
        eng.set_option("r1_virtual", virtual)
        if bound:
            eng.bind_traces(Cm)
        else:
            eng.bind_traces(None)
        eng.fit_ring_model(0, A, Cm)
        eng.profile(True); eng.profile_reset()
        eng.residual(0, A[:, prev], Cm[prev] if not bound else _rows(Cm, prev))     # a footprint term on top (what a patch with halo neurons asks for)
        Anew = eng.update_spatial(0, alg, A, Cm, IND, sn=f.sn if alg == "hals_thresh" else None, param=20 if alg == "nnls" else 3)
        eng.residual(0, A, Cm)
        c = eng.hals_temporal(0, Anew, Cm, 3)
        tab = eng.profile_table(); eng.profile(False)
        sn = eng.get_sn(0)                                   # a consumer of Ysig itself: the virtual residual is realized (swept) now
        return Anew, c, tab, sn
    def _rows(M, ind):
        from cnmf_e_amd.engine import BoundRows
        return BoundRows(M, ind)
    try:
        for alg in ("hals", "hals_thresh", "nnls"):
            (a0, c0, t0, sn0), (a1, c1, t1, sn1) = run(0, alg), run(1, alg)
            sweeps = lambda t: sum(v["calls"] for k, v in t.items() if k.startswith("residual_r1"))
            assert sweeps(t0) == 1 and sweeps(t1) == 0, (t0.keys(), t1.keys())
            assert _calls(t1, "spatial_from_ptab") == 1 and _calls(t1, "temporal_proj_B") >= 1 and _calls(t1, "spatial_proj_U") == 0 and _calls(t1, "temporal_proj_U") == 0
            assert _calls(t1, "spatial_ptab_proj") == (0 if bound else 1), t1.keys()      # bound traces: the fit's window projection left the table
            d0, d1_ = a0.toarray(), a1.toarray()
            scale = np.abs(d0).max()
            off = np.abs(d0 - d1_) > 2e-5 * scale
            assert off.sum() <= 2 and rel(d1_[~off], d0[~off]) <= 2e-6, (alg, off.sum(), rel(d1_, d0))      # (a thresholded entry may flip on a 1e-7 difference of U)
            if off.sum() == 0:
                for x, z in zip(c0, c1):
                    assert rel(z, x) <= 3e-6, (alg, rel(z, x))
            assert np.allclose(sn0, sn1, rtol=2e-5)
    finally:
        eng.set_option("r1_virtual", 1)
        eng.bind_traces(None)


@pytest.mark.parametrize("alg", ["hals", "nnls"])
@pytest.mark.parametrize("pdims,r", [([32, 32], 5), ([40, 36], 15)])
def test_virtual_iterations_on_patches_against_the_oracle(eng, alg, pdims, r):
    """method level, 2 x 2 patches inside blocks (halo neurons, the pending term through both projections), two iterations against the oracle and against the
    swept engine"""
    import cnmfe_oracle as orc
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K = 2 * pdims[0], 2 * pdims[1], 150, 10
    f, Y, _ = _setup(eng, d1, d2, T, K, r, 41, pdims)
    res = {}
    for virtual in (1, 0):
        eng.set_option("r1_virtual", virtual)
        video = PatchedVideo(d1, d2, T, pdims, r, eng)
        video.upload_from_full(Y)
        s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=alg, maxIter=3), f.A_init, f.C_init, f.sn)
        eng.profile(True); eng.profile_reset()
        for _ in range(2):
            s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        tab = eng.profile_table(); eng.profile(False)
        res[virtual] = (s.A.toarray(), np.asarray(s.C).copy(), tab)
    eng.set_option("r1_virtual", 1)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, pdims, r, f.A_init.astype(np.float32), f.C_init, f.sn, spatial_algorithm=alg, maxIter=3)
    for _ in range(2):
        o.update_background_parallel(); o.update_spatial_parallel(); o.update_temporal_parallel()
    Av, Cv, tv = res[1]; As, Cs, ts = res[0]
    assert sum(v["calls"] for k, v in tv.items() if k.startswith("residual_r1")) == 0 and _calls(tv, "temporal_proj_B") > 0 and _calls(tv, "spatial_from_ptab") > 0
    assert sum(v["calls"] for k, v in ts.items() if k.startswith("residual_r1")) > 0
    Ao = o.A.toarray()
    assert rel(Av, Ao) <= 5e-6 and rel(Cv, o.C) <= 5e-6, (rel(Av, Ao), rel(Cv, o.C))
    assert rel(Av, As) <= 5e-6 and rel(Cv, Cs) <= 5e-6


def test_virtual_residual_serves_every_consumer(eng):
    """whoever needs Ysig itself after a virtual residual gets the swept values: an export, GetSn, fast_temporal, compute_RSS, reconstruct_background"""
    d1, d2, T, r, K = 48, 44, 128, 5, 5
    f, Y, video = _setup(eng, d1, d2, T, K, r, 51)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.fit_ring_model(0, A, Cm)
    b0 = eng.b0(0)
    out = {}
    try:
        for virtual in (0, 1):
            eng.set_option("r1_virtual", virtual)
            got = []
            for consumer in ("export", "sn", "fast", "rss", "bg"):
                eng.set_b0(0, b0)                            # invalidates the resident residual
                eng.residual(0, A, Cm)                       # virtual (or swept)
                if consumer == "export":
                    got.append(eng.residual(0, A[:, :2], Cm[:2], want=True))
                elif consumer == "sn":
                    got.append(eng.get_sn(0))
                elif consumer == "fast":
                    got.append(eng.fast_temporal(0, A)[0])
                elif consumer == "rss":
                    got.append(np.array([eng.compute_rss(0, A, Cm, b0, b0)]))
                else:
                    got.append(eng.reconstruct_background(0, b0, b0, 3, 17))
            out[virtual] = got
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b) or rel(b, a) <= 1e-6
    finally:
        eng.set_option("r1_virtual", 1)


def test_virtual_spatial_update_with_changed_traces_builds_its_own_table(eng):
    """the fit's table belongs to the traces of the fit: after the bound matrix changes (a new generation) or with fewer / other neurons the update must not read it"""
    import cnmfe_oracle as orc
    d1, d2, T, r, K = 56, 48, 140, 5, 7
    f, Y, video = _setup(eng, d1, d2, T, K, r, 61)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    C2 = np.ascontiguousarray(Cm[::-1] * 1.25 + 0.5, dtype=np.float32)            # other traces (and another order) for the same footprints
    IND = sp.csc_matrix(A.toarray() > 0).astype(np.float32)
    try:
        ref = {}
        for virtual in (0, 1):
            eng.set_option("r1_virtual", virtual)
            eng.bind_traces(Cm)
            eng.fit_ring_model(0, A, Cm)
            eng.profile(True); eng.profile_reset()
            eng.residual(0, None, None)
            a_same = eng.update_spatial(0, "hals", A, Cm, IND)
            eng.bind_traces(C2)                               # a new generation of the bound matrix
            a_other = eng.update_spatial(0, "hals", A, C2, IND)
            sub = np.array([1, 2, 5])
            a_sub = eng.update_spatial(0, "hals", A[:, sub], C2[sub], IND[:, sub])
            tab = eng.profile_table(); eng.profile(False)
            ref[virtual] = (a_same.toarray(), a_other.toarray(), a_sub.toarray(), tab)
        tv = ref[1][3]
        assert _calls(tv, "spatial_from_ptab") == 3 and _calls(tv, "spatial_ptab_proj") == 2, tv.keys()      # the first update read the fit's table, the other two built theirs
        for x, z in zip(ref[0][:3], ref[1][:3]):
            assert rel(z, x) <= 3e-6
    finally:
        eng.set_option("r1_virtual", 1)
        eng.bind_traces(None)