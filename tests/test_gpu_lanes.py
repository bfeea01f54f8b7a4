"""Execution lanes (option "lanes", csrc/common.hpp cnmfe_lane): the patches of a context alternate between two (or more) streams, each with its own set of the
context's scratch; calls on what the patches share join the lanes.  Everything a patch computes is independent of the other patches inside an update
(update_*_parallel.m: parfor), so a run on two or three lanes must give what the run on one gives -- the same kernels on the same inputs, the stitch accumulated in
the same order: EQUAL arrays, not close ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lanes, dims, pdims, T, K, r, seed, opt_kw, iters=2, update_sn=False, opts=None):
    from cnmf_e_amd import synth
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2 = dims
    f = synth.make_factors(d1, d2, T, K, seed, gSig=1.5, gSiz=7, min_sep=4)
    Y = synth.make_video(f, np.float32)
    eng = Engine(0)
    try:
        eng.set_option("lanes", lanes)
        for k, v in (opts or {}).items():
            eng.set_option(k, v)
        video = PatchedVideo(d1, d2, T, pdims, r, eng)
        video.upload_from_full(Y)
        s = Sources2D(video, Options(ring_radius=r, **opt_kw), f.A_init, f.C_init, f.sn)
        out = []
        for it in range(iters):
            s.update_background_parallel()
            W = [s.get_W(idx).data.copy() for idx in video.owned]
            b0n = np.array(s.b0_new, dtype=np.float64)
            s.update_spatial_parallel(update_sn=update_sn and it == 0)
            A = s.A.toarray()
            s.update_temporal_parallel()
            C = np.asarray(s.C, dtype=np.float32).copy()
            out.append((W, b0n, A, C, np.asarray(s.P["sn"]).copy()))
        rss = s.compute_RSS()[0]
        return out, rss
    finally:
        eng.close()


def _same(a, b, what):
    assert np.all(np.isfinite(a)), what
    assert np.array_equal(a, b), (what, float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()))


CASES = [
    ("hals 2x2", (48, 44), [24, 22], 300, 8, 5, dict(spatial_algorithm="hals", maxIter=3), False),
    ("nnls 3x2 update_sn", (60, 44), [20, 22], 256, 9, 5, dict(spatial_algorithm="nnls", maxIter=2), True),
    ("deconv 2x2", (44, 40), [22, 20], 400, 6, 5, dict(spatial_algorithm="hals", maxIter=2, deconv_flag=True), False),
    ("bg_ssub 2x2", (64, 56), [32, 28], 200, 6, 8, dict(spatial_algorithm="hals", maxIter=2, bg_ssub=2), False),
    ("outliers 2x3", (44, 60), [22, 20], 300, 8, 5, dict(spatial_algorithm="hals_thresh", maxIter=2, thresh_outlier=4.0), False),
]


@pytest.mark.parametrize("name,dims,pdims,T,K,r,kw,usn", CASES, ids=[c[0] for c in CASES])
def test_two_lanes_equal_one(name, dims, pdims, T, K, r, kw, usn):
    ref, rss1 = _run(1, dims, pdims, T, K, r, 31, kw, update_sn=usn)
    got, rss2 = _run(2, dims, pdims, T, K, r, 31, kw, update_sn=usn)
    for it, ((W1, b1, A1, C1, sn1), (W2, b2, A2, C2, sn2)) in enumerate(zip(ref, got)):
        for p, (a, b) in enumerate(zip(W1, W2)):
            _same(b, a, "W of patch %d, iteration %d" % (p, it))
        _same(b2, b1, "b0_new %d" % it); _same(A2, A1, "A %d" % it); _same(C2, C1, "C %d" % it); _same(sn2, sn1, "sn %d" % it)
    assert rss1 == rss2, (rss1, rss2)


def test_three_lanes_and_the_swept_residual():
    """three lanes over six patches, and the residual as a sweep (r1_virtual = 0: resident Ysig per patch, the footprint terms folded in place)"""
    kw = dict(spatial_algorithm="hals", maxIter=2)
    ref, _ = _run(1, (60, 44), [20, 22], 256, 9, 5, 37, kw, opts={"r1_virtual": 0})
    got, _ = _run(3, (60, 44), [20, 22], 256, 9, 5, 37, kw, opts={"r1_virtual": 0})
    for (W1, b1, A1, C1, _), (W2, b2, A2, C2, _) in zip(ref, got):
        for a, b in zip(W1, W2):
            _same(b, a, "W")
        _same(A2, A1, "A"); _same(C2, C1, "C")


def test_lanes_are_set_before_the_patches():
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd._lib import CnmfeError
    from cnmf_e_amd.sources2d import PatchedVideo
    eng = Engine(0)
    try:
        PatchedVideo(24, 22, 40, [12, 11], 3, eng)
        with pytest.raises(CnmfeError):
            eng.set_option("lanes", 2)
        with pytest.raises(CnmfeError):
            eng.set_option("lanes", 9)
    finally:
        eng.close()
