"""bg_ssub > 1 without the sweep (round 5: option ssub_virtual: 2 always, 1 -- the default -- on patches of 5e8 samples and more; cnmf_e_amd/csrc/vproj.hip vproj_spatial_ssub / vproj_temporal_ssub, ssub.hip ssub_realize):
cnmfe_residual_ssub only RECORDS the request, the spatial and the temporal update take Ysig C' and A' Ysig through the resampling maps
(update_spatial_parallel.m:167-178, update_temporal_parallel.m:153-165), every other consumer realises the residual (low-resolution sweep + upsample) first.

Checked here against the SWEPT residual of the same build (ssub_virtual = 0: the path every bg_ssub parity test pinned against the oracle in rounds 2-4; those tests now
run sweep-free by default, so both forms are also compared with the float64 oracle): the sweep forms Ysig in fp32, the projections through the maps sum in fp64, so the
two differ by the sweep's own rounding (observed 1e-7 .. 4e-7 of max|A|, max|C|)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu

from parity_util import rel

SWEEP = {"ssub_up_fused", "ssub_up_cols", "residual_r1", "residual_r1_generic", "spatial_proj_U", "temporal_proj_U"}
MAPS = {"spatial_proj_rows", "spatial_ssub_combine", "temporal_proj_rows", "temporal_build_B", "temporal_proj_B"}


def _run(virt, f, Y, d1, d2, T, r, pdims, ssub, alg="hals", iters=2, update_sn=False):
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    eng = Engine(0)
    try:
        eng.set_option("ssub_virtual", 2 if virt else 0)      # (1, the default, takes the sweep-free form only on patches large enough for it to pay)
        video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
        video.upload_from_full(Y)
        s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=alg, maxIter=3, bg_ssub=ssub), f.A_init, f.C_init, f.sn)
        eng.profile(True)
        for it in range(iters):
            s.update_background_parallel()
            s.update_spatial_parallel(update_sn=update_sn and it == 0)
            s.update_temporal_parallel()
        names = {k for k, v in eng.profile_table().items() if v["calls"]}
        return s.A.toarray().astype(np.float64), np.asarray(s.C, dtype=np.float64).copy(), np.asarray(s.P["sn"], dtype=np.float64).copy(), names
    finally:
        eng.close()


@pytest.mark.parametrize("dims,T,K,r,pdims,ssub,alg", [((96, 80), 400, 12, 10, None, 2, "hals"),          # one patch = the field of view
                                                       ((75, 66), 203, 9, 18, [40, 33], 2, "hals"),       # 2 x 2 patches with halo: patch rows != block rows, odd sizes, r = 18 -> 9
                                                       ((66, 60), 150, 8, 15, [33, 30], 3, "nnls"),       # bg_ssub = 3: 10-tap transposed maps, ring radius 5
                                                       ((64, 64), 260, 10, 10, None, 2, "hals_thresh")])
def test_projections_through_the_resampling_maps_equal_the_sweep(dims, T, K, r, pdims, ssub, alg):
    from cnmf_e_amd import synth
    d1, d2 = dims
    f = synth.make_factors(d1, d2, T, K, 7, gSig=1.5, gSiz=7, min_sep=4)
    Y = synth.make_video(f, np.float32)
    A1, C1, _, n1 = _run(1, f, Y, d1, d2, T, r, pdims, ssub, alg)
    A0, C0, _, n0 = _run(0, f, Y, d1, d2, T, r, pdims, ssub, alg)
    assert MAPS <= n1 and not (SWEEP & n1), sorted(n1)                   # no sweep, no upsample, no projection of a realised Ysig ...
    assert {"spatial_proj_U", "temporal_proj_U"} <= n0 and not ({"spatial_proj_rows", "temporal_proj_rows"} & n0), sorted(n0)      # ... and the swept form is what it was
    assert np.array_equal(A1 != 0, A0 != 0) or alg != "hals"             # (a threshold / an active set may flip on a last-bit difference)
    assert rel(A1, A0) <= 2e-6 and rel(C1, C0) <= 2e-6, (rel(A1, A0), rel(C1, C0))


def test_a_consumer_of_ysig_itself_realises_the_recorded_residual():
    """update_sn = true makes the spatial update read GetSn(Ysig) (update_spatial_parallel.m:191-194): the recorded residual is realised by the low-resolution sweep and
    its upsample, the pending footprint term is folded into it (residual_materialize), and everything downstream equals the swept run"""
    from cnmf_e_amd import synth
    d1, d2, T, K, r = 80, 64, 320, 10, 10
    f = synth.make_factors(d1, d2, T, K, 11, gSig=1.5, gSiz=7, min_sep=4)
    Y = synth.make_video(f, np.float32)
    A1, C1, sn1, n1 = _run(1, f, Y, d1, d2, T, r, [40, 32], 2, update_sn=True)
    A0, C0, sn0, n0 = _run(0, f, Y, d1, d2, T, r, [40, 32], 2, update_sn=True)
    assert "ssub_up_fused" in n1 or "ssub_up_cols" in n1, sorted(n1)    # realised in the first iteration ...
    assert "spatial_proj_rows" in n1, sorted(n1)                         # ... sweep-free in the second
    assert rel(sn1, sn0) <= 2e-6, rel(sn1, sn0)
    assert rel(A1, A0) <= 2e-6 and rel(C1, C0) <= 2e-6, (rel(A1, A0), rel(C1, C0))
