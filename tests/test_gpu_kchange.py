"""K changes between the update calls: the demo merges and deletes neurons between them (demos/demo_large_data_1p.m:166,175-178,203-209:
merge_high_corr, remove_false_positives, merge_neurons_dist_corr -- host methods off this engine's path that REPLACE obj.A / obj.C with fewer columns / rows
while obj.A_prev / obj.C_prev, W and b0 stay what the last background update left).  The engine keeps state keyed on the previous (A_prev, C_prev) per patch:
the bound trace matrix, pending footprint terms, the P = Yc Cc' table of the last fit, kept covariance tables, temporal job buffers.  After an iteration
three neurons are deleted and two merged (new column = sum of the footprints, new trace = the weighted mean) on BOTH sides; the next background -> spatial ->
temporal round, and a spatial -> temporal round WITHOUT a background update in between (the K-changed branch of :205-209), must match the oracle given the
same edited (A, C)."""
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


def _edit(A, C, C_raw):
    """delete neurons 2, 7, 9; merge 0 and 1 into column 0 (footprints added, traces averaged with the footprints' energies as weights)"""
    A = np.asarray(A.todense() if sp.issparse(A) else A, dtype=np.float64).copy()
    C = np.asarray(C, dtype=np.float64).copy(); C_raw = np.asarray(C_raw, dtype=np.float64).copy()
    w0, w1 = (A[:, 0] ** 2).sum(), (A[:, 1] ** 2).sum()
    A[:, 0] = A[:, 0] + A[:, 1]
    C[0] = (w0 * C[0] + w1 * C[1]) / (w0 + w1); C_raw[0] = (w0 * C_raw[0] + w1 * C_raw[1]) / (w0 + w1)
    keep = np.setdiff1d(np.arange(A.shape[1]), [1, 2, 7, 9])
    return sp.csc_matrix(A[:, keep]), C[keep], C_raw[keep]


@pytest.mark.parametrize("deconv", [False, True])
@pytest.mark.parametrize("pdims", [None, [36, 36]])
def test_k_changes_between_the_update_calls(eng, pdims, deconv):
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 72, 72, 600, 12, 15
    f = synth.make_factors(d1, d2, T, K, 17, gSig=2.0, gSiz=9, min_sep=6)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=3, deconv_flag=deconv), f.A_init, f.C_init, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, pdims or [d1, d2], r, f.A_init.astype(np.float32), f.C_init, f.sn,
                            spatial_algorithm="hals", maxIter=3, deconv_options={} if deconv else None)
    def both(name, *a):
        getattr(s, name)(*a); getattr(o, name)(*a)
    def check(tag, tolA=3e-6, tolC=5e-6):
        eA, eC, eR = rel(s.A.toarray(), o.A.toarray()), rel(np.asarray(s.C), o.C), rel(np.asarray(s.C_raw), o.C_raw)
        assert s.A.shape == o.A.shape and np.asarray(s.C).shape == o.C.shape
        assert eA <= tolA and eC <= tolC and eR <= tolC, (tag, eA, eC, eR)
    for m in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"):
        both(m)
    check("iteration 1")
    # ---- the edit: on each side from its own state (they agree to 1e-6), K = 12 -> 8 ----
    A2, C2, R2 = _edit(s.A, s.C, s.C_raw); s.set_components(A2, C2, R2)
    A2o, C2o, R2o = _edit(o.A, o.C, o.C_raw); o.A, o.C, o.C_raw = sp.csc_matrix(A2o), C2o, R2o
    assert s.A.shape[1] == 8 and s.A_prev.shape[1] == 12
    # the K-changed branch of demo_large_data_1p.m:205-209: spatial + temporal straight away (A_prev, C_prev, W, b0 of the last background update)
    both("update_spatial_parallel"); both("update_temporal_parallel")
    check("spatial + temporal after the edit")
    # and a full round with the new K (:199-201)
    for m in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"):
        both(m)
    check("iteration after the edit")
    for idx in video.order:
        Wc = sp.csr_matrix(o.W[idx]); Wc.sort_indices()
        assert rel(s.get_W(idx).data, Wc.data) <= 2e-6
    # delete through the mirror of obj.delete (Sources2D.m:762-811), then a temporal update alone
    s.delete([3]); keep = np.setdiff1d(np.arange(o.A.shape[1]), [3])
    o.A, o.C, o.C_raw = sp.csc_matrix(o.A)[:, keep], o.C[keep], o.C_raw[keep]
    both("update_temporal_parallel")
    check("temporal after delete")
