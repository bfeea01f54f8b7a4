"""Round 6's two variants of the packed ring solve against the kernel they build on (k_ring_solve6, itself checked against the oracle by the parity tests):

* option solve_staged (default 1, cnmf_e_amd/csrc/ring_solve_staged.hpp): the footprints' U~ and A sampled out of per-neuron windows instead of CSR rows and slot
  tables -- the same neurons in the same order, the same arithmetic: the weights must be BIT-identical;
* option solve_inv (default 0, ring_solve_inv.hpp): the fit out of cached explicit inverses of the video's systems (Woodbury over the footprints' rank-2 terms +
  a Neumann series for the ridge's drift) -- a different algorithm for fit_ring_model.m:106; agreement to fp32 rounding of W, incl. the pixels it must leave to the
  factorising kernel (more than 8 neurons around a ring; two footprints meeting the ring in the same single pixel: a singular block of the small system)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    from cnmf_e_amd.engine import Engine
    e = Engine(0)
    yield e
    for k, v in (("solve_staged", 1), ("solve_inv", 0), ("solve_probe", 0)):
        e.set_option(k, v)
    e.close()


def _video(eng, d1, d2, T, K, r, seed, pdims=None, **kw):
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    f = synth.make_factors(d1, d2, T, K, seed, **kw)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    return f, Y, video


def _seq(f):
    A0, A1 = f.A_init.tocsc().astype(np.float32), f.A_true.tocsc().astype(np.float32)
    C0, C1 = np.ascontiguousarray(f.C_init, dtype=np.float32), np.ascontiguousarray(f.C_true, dtype=np.float32)
    mix = lambda t: ((1 - t) * A0 + t * A1).tocsc().astype(np.float32)
    return [(A0, C0), (mix(0.3), ((0.7 * C0 + 0.3 * C1)).astype(np.float32)), (None, None), (A0[:, :2].tocsc(), C0[:2]), (A1, C1)]


def _fits(eng, pid, r, seq, **opts):
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.ring_init(pid, r)
    out = []
    for Ai, Ci in seq:
        eng.fit_ring_model(pid, Ai, Ci)
        out.append(eng.ring_csr(pid).data.copy())
    return out


@pytest.mark.parametrize("dims,T,r,K,kw", [((40, 36), 600, 5, 5, {}), ((64, 60), 300, 15, 8, {}), ((96, 80), 400, 15, 12, {}), ((40, 36), 9200, 5, 5, {}),
                                           ((70, 64), 200, 15, 40, dict(gSig=1.5, gSiz=7, min_sep=3)), ((75, 66), 160, 18, 9, {})])
def test_staged_solve_is_bit_identical(eng, dims, T, r, K, kw):
    """first run, footprints moving, no footprints, a subset, the truth; frame stride 2 (T = 9200); 10-25 neurons around every ring (K = 40: several staging
    rounds); radius 18 (8 tiles: the staged kernel does not take it, the fit falls back)"""
    d1, d2 = dims
    f, Y, video = _video(eng, d1, d2, T, K, r, 11, **kw)
    seq = _seq(f)
    a = _fits(eng, 0, r, seq, solve_staged=0, solve_inv=0)
    b = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=0)
    for wa, wb in zip(a, b):
        assert np.all(np.isfinite(wb))
        assert np.array_equal(wa.view(np.uint32), wb.view(np.uint32))


def test_staged_solve_on_patches(eng):
    """2 x 2 patches: ring pixels in the halo, windows clipped by the block region"""
    d1, d2, T, K, r = 58, 68, 240, 9, 15
    f, Y, video = _video(eng, d1, d2, T, K, r, 5, pdims=[32, 36])
    A = f.A_init.tocsc().astype(np.float32)
    C = np.ascontiguousarray(f.C_init, dtype=np.float32)
    res = {}
    for mode in (0, 1):
        eng.set_option("solve_staged", mode)
        out = []
        for idx in video.owned:
            pid = video.pid[idx]
            eng.ring_init(pid, r)
            rows = video.block_pix[idx]
            Ab = A[rows].tocsc()
            for sc in (1.0, 0.8):
                eng.fit_ring_model(pid, (Ab * sc).tocsc().astype(np.float32), C)
                out.append(eng.ring_csr(pid).data.copy())
        res[mode] = out
    for wa, wb in zip(res[0], res[1]):
        assert np.array_equal(wa.view(np.uint32), wb.view(np.uint32))


@pytest.mark.parametrize("dims,T,r,K,kw,mode", [((64, 60), 300, 15, 8, {}, 2), ((96, 80), 400, 15, 12, {}, 1), ((40, 36), 9200, 5, 5, {}, 2),
                                                ((70, 64), 200, 15, 40, dict(gSig=1.5, gSiz=7, min_sep=3), 2)])
def test_solve_out_of_cached_inverses(eng, dims, T, r, K, kw, mode):
    """solve_inv = 2 builds the inverses in front of the first fit (at the ridge of a fit without footprints: the series has the most to do), 1 in front of the
    second at the ridge the first left; K = 40: most pixels have more than 8 neurons around them and go to the factorising kernel"""
    d1, d2 = dims
    f, Y, video = _video(eng, d1, d2, T, K, r, 11, **kw)
    seq = _seq(f)
    a = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=0)
    b = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=mode, solve_probe=512)
    st = eng.ring_solve_stats(0)
    assert st["left_over"] >= 0 and st["series_terms"] >= 0
    for wa, wb in zip(a, b):
        assert np.all(np.isfinite(wb))
        assert np.abs(wb - wa).max() <= 5e-7 * np.abs(wa).max(), np.abs(wb - wa).max() / np.abs(wa).max()


def test_cached_inverses_with_footprints_that_meet_the_ring_in_one_pixel(eng):
    """two footprints whose only pixel on some centre's ring is the SAME pixel: their columns of A~ are linearly dependent there, the first block of the small
    system is singular -- the pivot test must hand those centres to the factorising kernel"""
    d1, d2, T, r = 64, 60, 300, 15
    f, Y, video = _video(eng, d1, d2, T, 4, r, 3)
    A = f.A_init.tocsc().astype(np.float32).tolil()
    q = 30 * d1 + 30                                            # one extra pixel shared by two more neurons
    extra = sp.lil_matrix((d1 * d2, 2), dtype=np.float32)
    extra[q, 0] = 0.7; extra[q, 1] = 0.3
    A2 = sp.hstack([A.tocsc(), extra.tocsc()]).tocsc().astype(np.float32)
    rng = np.random.default_rng(0)
    C2 = np.ascontiguousarray(np.vstack([f.C_init, np.abs(rng.normal(0, 1, (2, T)))]), dtype=np.float32)
    seq = [(A2, C2), ((A2 * 0.9).tocsc().astype(np.float32), C2)]
    a = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=0)
    b = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=2)
    assert eng.ring_solve_stats(0)["left_over"] > 0
    for wa, wb in zip(a, b):
        assert np.all(np.isfinite(wb))
        assert np.abs(wb - wa).max() <= 5e-7 * np.abs(wa).max()


def test_staged_solve_falls_back_for_a_footprint_larger_than_a_window(eng):
    """a footprint whose bounding box, dilated by two ring radii, has more entries than a neuron's window holds (two pixels in opposite corners of a 200 x 180
    field of view): the fit must take k_ring_solve6 for this call -- and give the same weights"""
    d1, d2, T, r = 200, 180, 120, 15
    f, Y, video = _video(eng, d1, d2, T, 6, r, 9)
    A = f.A_init.tocsc().astype(np.float32).tolil()
    A[3 * d1 + 4, 0] = 0.5; A[(d2 - 5) * d1 + (d1 - 6), 0] = 0.4
    A = A.tocsc().astype(np.float32)
    C = np.ascontiguousarray(f.C_init, dtype=np.float32)
    seq = [(A, C), ((A * 0.9).tocsc().astype(np.float32), C)]
    a = _fits(eng, 0, r, seq, solve_staged=0, solve_inv=0)
    b = _fits(eng, 0, r, seq, solve_staged=1, solve_inv=0)
    for wa, wb in zip(a, b):
        assert np.all(np.isfinite(wb)) and np.array_equal(wa.view(np.uint32), wb.view(np.uint32))
