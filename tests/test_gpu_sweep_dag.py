"""The Gauss-Seidel sweeps of an update as ONE dependency graph over (sweep, neuron) items (csrc/factor.hip dag_schedule, option sweep_dag, default 1) against the
level-synchronous schedule of rounds 1-5 (sweep_dag = 0): every item reads and writes what it reads and writes in the sequential loops of HALS_spatial.m:36-44 /
HALS_temporal.m:59-68, so the results are EQUAL arrays -- spatial HALS, temporal HALS, the in-sweep deconvolution (whose last-sweep outputs S, C_raw travel as a flag of
the item), one patch and several (temporal jobs), one lane and two."""
import numpy as np
import pytest

from test_gpu_lanes import _run, _same

pytestmark = pytest.mark.gpu

CASES = [
    ("hals one patch", (48, 44), [48, 44], 300, 14, 5, dict(spatial_algorithm="hals", maxIter=4), 1),
    ("hals_thresh 2x2", (48, 44), [24, 22], 300, 10, 5, dict(spatial_algorithm="hals_thresh", maxIter=3), 1),
    ("deconv one patch", (44, 40), [44, 40], 400, 10, 5, dict(spatial_algorithm="hals", maxIter=3, deconv_flag=True), 1),
    ("deconv 2x2 two lanes", (44, 40), [22, 20], 400, 8, 5, dict(spatial_algorithm="hals", maxIter=3, deconv_flag=True), 2),
    ("crowded", (40, 36), [40, 36], 256, 30, 5, dict(spatial_algorithm="hals", maxIter=5), 1),
]


@pytest.mark.parametrize("name,dims,pdims,T,K,r,kw,lanes", CASES, ids=[c[0] for c in CASES])
def test_graph_schedule_equals_the_level_synchronous_sweeps(name, dims, pdims, T, K, r, kw, lanes):
    ref, rss1 = _run(lanes, dims, pdims, T, K, r, 41, kw, opts={"sweep_dag": 0})
    got, rss2 = _run(lanes, dims, pdims, T, K, r, 41, kw)
    for it, ((W1, b1, A1, C1, sn1), (W2, b2, A2, C2, sn2)) in enumerate(zip(ref, got)):
        for a, b in zip(W1, W2):
            _same(b, a, "W %d" % it)
        _same(A2, A1, "A %d" % it); _same(C2, C1, "C %d" % it)
    assert rss1 == rss2


def test_graph_schedule_launches_fewer_levels():
    """neurons spread over a field of view: the chains of different sweeps do not line up and the graph over 5 sweeps is shallower than 5 x (levels of a sweep)"""
    from cnmf_e_amd import synth
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 96, 88, 128, 40, 5
    f = synth.make_factors(d1, d2, T, K, 43, gSig=1.5, gSiz=7, min_sep=4)
    Y = synth.make_video(f, np.float32)
    calls = {}
    for dag in (0, 1):
        eng = Engine(0)
        try:
            eng.set_option("sweep_dag", dag)
            video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
            video.upload_from_full(Y)
            s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
            s.update_background_parallel(); s.update_spatial_parallel()
            eng.profile(True); eng.profile_reset()
            s.update_temporal_parallel()
            eng.synchronize()
            calls[dag] = eng.profile_table()["temporal_hals_level"]["calls"]
        finally:
            eng.close()
    assert calls[1] < calls[0], calls
