"""Committed fixtures (tests/golden/, made by tests/golden/make_golden.py from the float64 restatement -- NOT from MATLAB,
see that script's header).  CPU: the oracle still reproduces them and they contain the planted truth.  GPU: the HIP engine
against the committed numbers."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


from parity_util import rel


def _oracle_iteration(z):
    import cnmfe_oracle as orc
    d1, d2, T, r = int(z["d1"]), int(z["d2"]), int(z["T"]), int(z["r"])
    o = orc.OracleSources2D(z["Y"].T.reshape(d1, d2, T, order="F"), d1, d2, T, list(z["patch"]), r, z["A_init"], z["C_init"], z["sn"],
                            spatial_algorithm="hals", maxIter=3)
    o.update_background_parallel(); o.update_spatial_parallel(); o.update_temporal_parallel()
    return o


def test_oracle_reproduces_iteration_fixture():
    z = np.load(os.path.join(GOLD, "iteration_2x2.npz"))
    o = _oracle_iteration(z)
    assert rel(o.A.toarray(), z["A_after_spatial"]) <= 1e-10
    assert rel(o.C, z["C_after_temporal"]) <= 1e-10
    assert rel(o.b0_new, z["b0_new"]) <= 1e-10
    for idx in sorted(o.W):
        W = np.asarray(o.W[idx].todense() if hasattr(o.W[idx], "todense") else o.W[idx])
        assert rel(W, z["W_%d_%d" % idx]) <= 1e-8


def test_oasis_fixture_contains_planted_truth():
    import oasis_oracle as oo
    z = np.load(os.path.join(GOLD, "oasis_ar1.npz"))
    assert np.all(np.abs(z["g"] - 0.93) < 0.02)                            # planted AR(1) coefficient
    assert np.all(np.abs(z["b"] - np.array([0.5, 1.5, 0.0])) < 0.1)       # planted baselines
    assert np.all(np.abs(z["sn"] - 0.3) < 0.3)                            # planted noise 0.3 (spikes inflate the estimate)
    y = z["y"].astype(np.float64)
    for i in range(y.shape[0]):
        c, s, b, g = oo.deconvolveCa_ar1_foopsi(y[i], oo.GetSn(y[i]), smin=-5.0, optimize_pars=True, optimize_b=True)
        assert rel(c, z["c"][i]) <= 1e-9 and abs(g - z["g"][i]) <= 1e-12 and abs(b - z["b"][i]) <= 1e-10


@pytest.mark.gpu
def test_engine_matches_iteration_fixture():
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    z = np.load(os.path.join(GOLD, "iteration_2x2.npz"))
    d1, d2, T, r = int(z["d1"]), int(z["d2"]), int(z["T"]), int(z["r"])
    eng = Engine(0)
    try:
        video = PatchedVideo(d1, d2, T, list(z["patch"]), r, eng)
        video.upload_from_full(z["Y"])
        s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=3), sp.csc_matrix(z["A_init"]), z["C_init"], z["sn"])
        s.update_background_parallel()
        for idx in video.owned:
            Wg = s.get_W(idx).toarray()
            assert rel(Wg, z["W_%d_%d" % idx]) <= 5e-7            # observed 3e-8 (fp64 table + fp64 solve, W stored in fp32)
        s.update_spatial_parallel()
        Ag, Ar = s.A.toarray(), z["A_after_spatial"]
        assert ((Ag != 0) != (Ar != 0)).sum() == 0
        same = (Ag != 0) == (Ar != 0)
        assert rel(Ag[same], Ar[same]) <= 3e-6
        s.update_temporal_parallel()
        assert rel(s.C, z["C_after_temporal"]) <= 1e-6
        assert np.allclose(s.b0_new, z["b0_new"], rtol=1e-6, atol=2e-4)
    finally:
        eng.close()


@pytest.mark.gpu
def test_engine_matches_oasis_fixture():
    from cnmf_e_amd.engine import Engine
    z = np.load(os.path.join(GOLD, "oasis_ar1.npz"))
    eng = Engine(0)
    try:
        C, Craw, S, kp, sn = eng.deconv_temporal(z["y"], dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0))
        assert np.max(np.abs(sn - z["sn"]) / z["sn"]) <= 2e-4
        assert np.max(np.abs(kp - z["g"])) <= 2e-3
        assert rel(C, z["c"]) <= 2.5e-3                              # fp32 pools vs float64; Brent's search amplifies the difference
    finally:
        eng.close()
