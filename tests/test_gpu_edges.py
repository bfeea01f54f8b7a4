"""Edge cases of the hot path on the GPU: empty / ragged inputs, argument and state errors of the C ABI
(the reference has no tests of its own -- SURVEY.md section 4 -- so these follow its code paths: isempty(A) in
fit_ring_model.m:14-16, patches without neurons in update_*_parallel.m:121-124,123, num_neighbors in get_nhood.m:17-25)."""
import ctypes as C
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


def _video(eng, d1, d2, T, K, r, seed, pdims=None):
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    f = synth.make_factors(d1, d2, T, K, seed, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    return f, Y, video


@pytest.mark.parametrize("T", [61, 202, 303])
def test_ragged_frame_counts(eng, T):
    """T not a multiple of 4 (the resident video is 4-frame interleaved) nor of 16 (the Gram stage): full iteration vs oracle"""
    import cnmfe_oracle as orc
    from cnmf_e_amd.sources2d import Sources2D, Options
    d1, d2, K, r = 40, 36, 4, 5
    f, Y, video = _video(eng, d1, d2, T, K, r, 3)
    s = Sources2D(video, Options(ring_radius=r, maxIter=3), f.A_init, f.C_init, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [d1, d2], r, f.A_init.astype(np.float32), f.C_init, f.sn, maxIter=3)
    s.update_background_parallel(); o.update_background_parallel()
    s.update_spatial_parallel(); o.update_spatial_parallel()
    s.update_temporal_parallel(); o.update_temporal_parallel()
    assert rel(s.C, o.C) <= 3e-6 and rel(s.A.toarray(), o.A.toarray()) <= 3e-6


def test_no_neurons_at_all(eng):
    """K = 0: isempty(A) -> A = ones, C = zeros (fit_ring_model.m:14-16); spatial/temporal updates return empty factors"""
    import cnmfe_oracle as orc
    from cnmf_e_amd.sources2d import Sources2D, Options
    d1, d2, T, r = 36, 32, 120, 5
    f, Y, video = _video(eng, d1, d2, T, 3, r, 5)
    A0 = sp.csc_matrix((d1 * d2, 0), dtype=np.float32); C0 = np.zeros((0, T), np.float32)
    s = Sources2D(video, Options(ring_radius=r, maxIter=2), A0, C0, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [d1, d2], r, A0, C0, f.sn, maxIter=2)
    s.update_background_parallel(); o.update_background_parallel()
    Wg = s.get_W((0, 0)); Wr = sp.csr_matrix(o.W[(0, 0)]); Wr.sort_indices()
    assert rel(Wg.data, Wr.data) <= 5e-7
    assert np.allclose(s.b0_new, o.b0_new, rtol=1e-6, atol=2e-4)
    s.update_spatial_parallel(); s.update_temporal_parallel()
    assert s.A.shape == (d1 * d2, 0) and s.C.shape == (0, T)


def test_patch_without_neurons_is_skipped(eng):
    """2x2 patches, all footprints in one corner: the other patches take the `continue` branches (update_*_parallel.m:121-124, :123, :188-199)"""
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, r = 48, 44, 150, 5
    f = synth.make_factors(d1, d2, T, 6, 9, gSig=1.5, gSiz=7, min_sep=5)
    rr, cc = np.meshgrid(np.arange(d1), np.arange(d2), indexing="ij")
    cols = []
    for (r0, c0) in ((7, 6), (11, 9)):                                  # two footprints inside the top-left patch, away from its halo
        g = np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2 * 1.5 ** 2)); g[g < 0.05] = 0
        cols.append(sp.csc_matrix(g.reshape(-1, 1, order="F").astype(np.float32)))
    A0 = sp.hstack(cols).tocsc(); C0 = f.C_init[:2]
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, [24, 22], r, eng); video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, maxIter=3), A0, C0, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [24, 22], r, A0.astype(np.float32), C0, f.sn, maxIter=3)
    for _ in range(2):
        s.update_background_parallel(); o.update_background_parallel()
        s.update_spatial_parallel(); o.update_spatial_parallel()
        s.update_temporal_parallel(); o.update_temporal_parallel()
    assert rel(s.C, o.C) <= 2e-6
    assert np.allclose(s.b0_new, o.b0_new, rtol=1e-6, atol=2e-4)


def test_num_neighbors_subsampled_ring(eng):
    """num_neighbors < ring size (get_nhood.m:17-25): the generic R1 kernel and the fit on a thinned ring"""
    import cnmfe_oracle as orc
    d1, d2, T, r, nn = 44, 40, 100, 8, 20
    f, Y, video = _video(eng, d1, d2, T, 4, r, 21)
    eng.ring_init(0, r, nn)
    rs, cs = orc.get_nhood(r, nn)
    W0 = orc.build_ring_W(video.patch_pos[(0, 0)], video.block_pos[(0, 0)], d1, d2, rs, cs).tocsr(); W0.sort_indices()
    Wg = eng.ring_csr(0)
    assert np.array_equal(Wg.indices, W0.indices) and np.allclose(Wg.data, W0.data, rtol=1e-6)
    A = f.A_init.tocsc().astype(np.float32)
    eng.fit_ring_model(0, A, f.C_init)
    Wr, b0r = orc.fit_ring_model(Y.T.astype(np.float64), A.astype(np.float64), f.C_init, W0, np.nan, None, np.ones(d1 * d2, bool), True)
    Wr = sp.csr_matrix(Wr); Wr.sort_indices()
    assert rel(eng.ring_csr(0).data, Wr.data) <= 5e-7
    got = eng.residual(0, A, f.C_init, want=True).T
    ref = orc.residual_ysig(Y.T.astype(np.float64), A.astype(np.float64), f.C_init, sp.csr_matrix((eng.ring_csr(0).data, Wr.indices, Wr.indptr), shape=Wr.shape),
                            eng.b0(0).astype(np.float64), np.ones(d1 * d2, bool))
    assert rel(got, ref) <= 1e-4          # fitted weights (cancellation) in fp32 vs float64


def test_abi_state_and_argument_errors(eng):
    from cnmf_e_amd import _lib as L
    d1, d2, T, r = 30, 28, 64, 5
    f, Y, video = _video(eng, d1, d2, T, 3, r, 2)
    A = f.A_init.tocsc().astype(np.float32)
    with pytest.raises(L.CnmfeError):                      # ring not initialised yet
        eng.fit_ring_model(0, A, f.C_init)
    eng.ring_init(0, r)
    with pytest.raises(L.CnmfeError):                      # HALS before the residual of this patch exists
        eng.hals_temporal(0, A, f.C_init, 2)
    with pytest.raises(L.CnmfeError):                      # unknown patch, straight at the C ABI
        L.check(L.lib.cnmfe_residual(eng._ctx, 7, 0, None, None, None, None, L.ROWMAJOR, None, L.HOST))
    with pytest.raises(L.CnmfeError, match="cnmfe_set_noise"):      # the outlier branch needs the noise levels of the block first
        eng.fit_ring_model(0, A, f.C_init, thresh_outlier=3.0)
    with pytest.raises(L.CnmfeError, match="deferred"):             # nothing to fetch: no deferred spatial update on this context
        out = np.zeros(4, np.float32)
        L.check(L.lib.cnmfe_update_spatial_fetch(eng._ctx, out.ctypes.data_as(L.f32p), 4))
    with pytest.raises(L.CnmfeError, match="stitch_begin"):         # finish without begin (sync and async flavours)
        L.check(L.lib.cnmfe_stitch_finish(eng._ctx, 1, None, L.ROWMAJOR))
    p_ = L.lib.cnmfe_host_alloc(1 << 16)
    assert p_
    try:
        with pytest.raises(L.CnmfeError, match="stitch_begin"):
            L.check(L.lib.cnmfe_stitch_finish_async(eng._ctx, 1, C.cast(p_, L.f32p)))
    finally:
        L.lib.cnmfe_host_free(p_)
    L.check(L.lib.cnmfe_stitch_wait(eng._ctx))                      # nothing in flight: returns at once
    with pytest.raises(L.CnmfeError, match="cnmfe_background_ssub"):    # the bg_ssub objective without its preparation
        rss = C.c_double()
        L.check(L.lib.cnmfe_compute_rss_ssub(eng._ctx, 0, 0, None, None, None, None, L.ROWMAJOR, np.zeros(d1 * d2, np.float32).ctypes.data_as(L.f32p), C.byref(rss)))
    with pytest.raises(L.CnmfeError):                               # estimate_noise over more frames than the video has
        eng.estimate_noise(0, T + 1)
    cp = np.array([0, 2], np.int64); ri = np.array([5, 3], np.int32); va = np.ones(2, np.float32)     # rows of a column not ascending
    Cm = np.zeros((1, T), np.float32)
    with pytest.raises(L.CnmfeError):
        L.check(L.lib.cnmfe_residual(eng._ctx, 0, 1, cp.ctypes.data_as(L.i64p), ri.ctypes.data_as(L.i32p), va.ctypes.data_as(L.f32p),
                                     Cm.ctypes.data_as(L.f32p), L.ROWMAJOR, None, L.HOST))
    ri2 = np.array([3, d1 * d2 + 9], np.int32)                                                           # row index outside the block
    with pytest.raises(L.CnmfeError):
        L.check(L.lib.cnmfe_residual(eng._ctx, 0, 1, cp.ctypes.data_as(L.i64p), ri2.ctypes.data_as(L.i32p), va.ctypes.data_as(L.f32p),
                                     Cm.ctypes.data_as(L.f32p), L.ROWMAJOR, None, L.HOST))
    with pytest.raises(L.CnmfeError):                      # ring radius out of range
        eng.ring_init(0, 0)
    with pytest.raises(L.CnmfeError):                      # unknown tunable
        eng.set_option("no_such_option", 1)
    eng.residual(0, A, f.C_init)                            # ... and the context is still usable afterwards
    C2, Craw, aa = eng.hals_temporal(0, A, f.C_init, 2)
    assert np.isfinite(C2).all() and C2.shape == f.C_init.shape


def test_deconvolution_rejects_unsupported_lengths(eng):
    from cnmf_e_amd import _lib as L
    with pytest.raises(L.CnmfeError):
        eng.deconv_temporal(np.zeros((2, 32), np.float32), dict(type="ar1", method="foopsi"))       # T < 64
    with pytest.raises((L.CnmfeError, NotImplementedError, ValueError)):
        eng.deconv_temporal(np.zeros((2, 128), np.float32), dict(type="ar2", method="foopsi"))     # only AR(1) is built


def test_bound_traces_equal_uploaded_traces(eng):
    """cnmfe_traces_bind: (NULL, CNMFE_BOUND) must behave exactly like passing the matrix; a stale or mismatching binding is an error"""
    from cnmf_e_amd import _lib as L
    d1, d2, T, r = 36, 32, 96, 5
    f, Y, video = _video(eng, d1, d2, T, 4, r, 8)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.bind_traces(None)
    eng.fit_ring_model(0, A, Cm); W_plain = eng.ring_csr(0).data.copy()
    y_plain = eng.residual(0, A, Cm, want=True)
    c_plain = eng.hals_temporal(0, A, Cm, 3)
    eng.ring_init(0, r)
    eng.bind_traces(Cm)                                     # from here on `Cm` (this very object) goes as (NULL, CNMFE_BOUND)
    eng.fit_ring_model(0, A, Cm); W_bound = eng.ring_csr(0).data.copy()
    y_bound = eng.residual(0, A, Cm, want=True)
    c_bound = eng.hals_temporal(0, A, Cm, 3)
    assert np.array_equal(W_plain, W_bound) and np.array_equal(y_plain, y_bound)
    for a, b in zip(c_plain, c_bound):
        assert np.array_equal(a, b)
    c_copy = eng.hals_temporal(0, A, Cm.copy(), 3)          # an equal but different array object is simply uploaded
    assert np.array_equal(c_copy[1], c_plain[1])
    with pytest.raises(L.CnmfeError):                       # BOUND with a K that does not match the bound matrix
        L.check(L.lib.cnmfe_residual(eng._ctx, 0, 2, np.array([0, 1, 2], np.int64).ctypes.data_as(L.i64p), np.array([3, 5], np.int32).ctypes.data_as(L.i32p),
                                     np.ones(2, np.float32).ctypes.data_as(L.f32p), None, L.BOUND, None, L.HOST))
    eng.bind_traces(None)
    with pytest.raises(L.CnmfeError):                       # nothing bound any more
        L.check(L.lib.cnmfe_residual(eng._ctx, 0, 2, np.array([0, 1, 2], np.int64).ctypes.data_as(L.i64p), np.array([3, 5], np.int32).ctypes.data_as(L.i32p),
                                     np.ones(2, np.float32).ctypes.data_as(L.f32p), None, L.BOUND, None, L.HOST))


def test_incremental_residual_equals_full_sweep(eng):
    """A second cnmfe_residual under the same W, b0 is the resident Ysig plus the difference of two footprint terms (r1_delta, default on):
    every transition (none -> A, A -> A', A -> none, none -> none) must agree with the full ring sweep, and a change of W or b0 must
    force the sweep"""
    d1, d2, T, r = 44, 40, 96, 5
    f, Y, video = _video(eng, d1, d2, T, 6, r, 9)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.fit_ring_model(0, A, Cm)
    sel = [np.arange(0), np.arange(6), np.array([1, 4]), np.arange(0), np.arange(0), np.array([0, 2, 3, 5])]
    def run(delta):
        eng.set_option("r1_delta", delta)
        eng.set_b0(0, eng.b0(0))                        # invalidates the resident Ysig: the first call is a full sweep
        out = []
        for s in sel:
            out.append(eng.residual(0, A[:, s] if len(s) else None, Cm[s] if len(s) else None, want=True))
        return out
    try:
        full, inc = run(0), run(1)
        tab = None
        scale = max(np.abs(x).max() for x in full)
        for a, b in zip(full, inc):
            assert np.abs(a - b).max() <= 2e-6 * scale
        eng.set_option("r1_lazy", 0)                        # (lazy: the term would stay pending until somebody needs Ysig itself)
        eng.profile(True); eng.profile_reset()
        eng.residual(0, A, Cm)                              # delta (A' -> A)
        eng.fit_ring_model(0, A, Cm); eng.set_b0(0, eng.b0(0))      # new W, b0 (b0 rounded to fp32 so that the round trip below is exact)
        eng.residual(0, A[:, :3], Cm[:3])                   # must be a sweep
        tab = eng.profile_table(); eng.profile(False)
        assert tab["residual_delta"]["calls"] == 1
        assert sum(v["calls"] for k, v in tab.items() if k.startswith("residual_r1")) == 1
        y_after = eng.residual(0, A[:, :3], Cm[:3], want=True)
        eng.set_option("r1_delta", 0); eng.set_b0(0, eng.b0(0))
        y_ref = eng.residual(0, A[:, :3], Cm[:3], want=True)
        assert np.abs(y_after - y_ref).max() <= 2e-6 * scale
    finally:
        eng.set_option("r1_delta", 1); eng.set_option("r1_lazy", 1)


def test_pending_footprint_term_through_hals_temporal(eng):
    """r1_lazy (default): the second residual of an iteration leaves its footprint term pending; cnmfe_hals_temporal adds A' (W A)(C - mean C)
    to A' Ysig instead of sweeping the video again, every other consumer folds the term into Ysig first"""
    d1, d2, T, r = 44, 40, 128, 5
    f, Y, video = _video(eng, d1, d2, T, 6, r, 13)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.fit_ring_model(0, A, Cm)
    half = np.array([0, 2, 5])
    def run(lazy):
        eng.set_option("r1_lazy", lazy)
        eng.set_b0(0, eng.b0(0))
        eng.residual(0, A[:, half], Cm[half])               # sweep with a first term applied
        eng.profile(True); eng.profile_reset()
        eng.residual(0, A, Cm)                              # second residual: delta now (lazy = 0) or pending (lazy = 1)
        c = eng.hals_temporal(0, A, Cm, 3)
        n_delta = eng.profile_table().get("residual_delta", {"calls": 0})["calls"]
        sn = eng.get_sn(0) if T >= 64 else None             # a consumer of Ysig itself: forces the fold
        y = eng.residual(0, A, Cm, want=True)
        eng.profile(False)
        return c, n_delta, sn, y
    try:
        (c0, n0, sn0, y0), (c1, n1, sn1, y1) = run(0), run(1)
        assert n0 == 1 and n1 == 0
        for a, b in zip(c0, c1):
            assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())
        assert np.allclose(sn0, sn1, rtol=1e-5)
        assert np.abs(y0 - y1).max() <= 2e-6 * np.abs(y0).max()
    finally:
        eng.set_option("r1_lazy", 1)


@pytest.mark.parametrize("T,r", [(96, 5), (9200, 5)])
def test_incremental_gram_equals_direct_fp64(eng, T, r):
    """The incremental ring regression (video table once + footprint corrections per fit) against the direct fp64 Gram of Bf, over a
    sequence of fits with changing A, C -- including no footprints at all, a footprint set that shrinks, and (T = 9200 > 2 * 100 * p)
    the frame stride 2 of fit_ring_model.m:84-87 on the second fit, which rebuilds the kept table for the new stride"""
    d1, d2 = 40, 36
    f, Y, video = _video(eng, d1, d2, T, 5, r, 11)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    seq = [(A, Cm), (A * 0.8, Cm * 1.2), (None, None), (A[:, :2].tocsc(), Cm[:2]), (A, Cm)]
    def run(incr):
        eng.set_option("gram_incremental", incr); eng.set_option("gram_i8", 1 if incr else 0)      # (the reference: the direct Gram on the fp64 pipe)
        eng.ring_init(0, r)
        out = []
        for Ai, Ci in seq:
            _, info = eng.fit_ring_model(0, Ai, Ci)
            out.append((eng.ring_csr(0).data.astype(np.float64), eng.b0(0).astype(np.float64), info["frame_stride"]))
        return out
    try:
        eng.set_option("debug", 1)
        direct, inc = run(0), run(1)
        for (wd, bd, kd), (wi, bi, ki) in zip(direct, inc):
            assert kd == ki
            assert np.all(np.isfinite(wi))
            assert np.linalg.norm(wi - wd) <= 2e-6 * np.linalg.norm(wd), np.linalg.norm(wi - wd) / np.linalg.norm(wd)
            assert np.array_equal(bd, bi)
        if T > 9000:
            assert max(k for _, _, k in inc) == 2
    finally:
        eng.set_option("debug", 0); eng.set_option("gram_incremental", 1); eng.set_option("gram_i8", 1)


def test_bound_rows_equal_uploaded_rows(eng):
    """(ind, CNMFE_BOUND_ROWS): a subset of the bound trace matrix gathered on the device gives the same results as uploading C[ind];
    an index outside the bound matrix is an error, not a read"""
    from cnmf_e_amd import _lib as L
    from cnmf_e_amd.engine import BoundRows
    d1, d2, T, r = 36, 32, 96, 5
    f, Y, video = _video(eng, d1, d2, T, 6, r, 21)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    ind = np.array([4, 1, 3])
    eng.bind_traces(None)
    eng.fit_ring_model(0, A[:, ind], Cm[ind]); W0 = eng.ring_csr(0).data.copy()
    y0 = eng.residual(0, A[:, ind], Cm[ind], want=True)
    c0 = eng.hals_temporal(0, A[:, ind], Cm[ind], 3)
    eng.ring_init(0, r)
    eng.bind_traces(Cm)
    rows = BoundRows(Cm, ind)
    assert np.array_equal(np.asarray(rows), Cm[ind])
    eng.fit_ring_model(0, A[:, ind], rows); W1 = eng.ring_csr(0).data.copy()
    y1 = eng.residual(0, A[:, ind], rows, want=True)
    c1 = eng.hals_temporal(0, A[:, ind], rows, 3)
    assert np.array_equal(W0, W1) and np.array_equal(y0, y1)
    for a, b in zip(c0, c1):
        assert np.array_equal(a, b)
    bad = np.array([0, 6], np.int32)
    with pytest.raises(L.CnmfeError):
        L.check(L.lib.cnmfe_residual(eng._ctx, 0, 2, np.array([0, 1, 2], np.int64).ctypes.data_as(L.i64p), np.array([3, 5], np.int32).ctypes.data_as(L.i32p),
                                     np.ones(2, np.float32).ctypes.data_as(L.f32p), bad.ctypes.data_as(L.f32p), L.BOUND_ROWS, None, L.HOST))
    eng.bind_traces(None)


def test_bind_traces_from_device_pointer(eng):
    """cnmfe_traces_bind accepts a device pointer (the sharded temporal update binds the all-reduced C without a host round trip)"""
    import torch
    d1, d2, T, r = 36, 32, 100, 5
    f, Y, video = _video(eng, d1, d2, T, 4, r, 23)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.bind_traces(None)
    eng.fit_ring_model(0, A, Cm)
    y0 = eng.residual(0, A, Cm, want=True)
    t = torch.from_numpy(Cm).to("cuda:0").contiguous()
    torch.cuda.synchronize()
    eng.bind_traces(Cm, device_ptr=t.data_ptr())
    eng.set_b0(0, eng.b0(0)); y_ref = eng.residual(0, A, Cm.copy(), want=True)     # (b0 now fp32-rounded: reference for the bound run)
    eng.set_b0(0, eng.b0(0)); y1 = eng.residual(0, A, Cm, want=True)               # Cm is the bound identity -> (NULL, CNMFE_BOUND)
    assert np.array_equal(y_ref, y1)
    assert np.abs(y0 - y1).max() <= 1e-5 * np.abs(y0).max()
    eng.bind_traces(None)


def test_device_traces_handle(eng):
    """DeviceTraces: a trace matrix that stays in a torch device tensor -- bound device-to-device, subsets via BoundRows, host copy on demand"""
    import torch
    from cnmf_e_amd.engine import BoundRows, DeviceTraces
    d1, d2, T, r = 36, 32, 100, 5
    f, Y, video = _video(eng, d1, d2, T, 5, r, 27)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.bind_traces(None)
    eng.fit_ring_model(0, A, Cm); eng.set_b0(0, eng.b0(0))
    y0 = eng.residual(0, A, Cm, want=True)
    ind = np.array([0, 3, 4])
    eng.set_b0(0, eng.b0(0)); y0s = eng.residual(0, A[:, ind], Cm[ind], want=True)
    h = DeviceTraces(torch.from_numpy(Cm).to("cuda:0")); torch.cuda.synchronize()
    eng.bind_traces(h)
    assert h._host is None                                              # nothing downloaded so far
    assert np.allclose(h.mean(axis=1, dtype=np.float64), Cm.mean(axis=1, dtype=np.float64), rtol=1e-6) and h._host is None
    eng.set_b0(0, eng.b0(0)); y1 = eng.residual(0, A, h, want=True)
    eng.set_b0(0, eng.b0(0)); y1s = eng.residual(0, A[:, ind], BoundRows(h, ind), want=True)
    assert h._host is None
    assert np.array_equal(y0, y1) and np.array_equal(y0s, y1s)
    assert np.array_equal(np.asarray(h), Cm) and h._host is not None
    eng.bind_traces(None)


def test_nccl_collective_branches():
    """the sharded branches of the three update methods on a real RCCL group (one rank, force_collectives): all-gathers of A, the in-place
    all-reduce of the engine's stitch accumulator, the packed deconvolution exchange (scripts/nccl_smoke.py, in its own process: it owns a
    process group)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "nccl_smoke.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("forced collectives") == 2 and "compute_RSS ok" in out.stdout


def test_ring_change_rebuilds_the_kept_table(eng):
    """cnmfe_ring_init with another ring (radius, num_neighbors) changes which covariance sub-tiles are needed: the table of the video kept by
    the incremental fit must be rebuilt (debug = 1 NaN-poisons never-computed entries, so a stale table cannot go unnoticed)"""
    d1, d2, T = 44, 40, 96
    f, Y, video = _video(eng, d1, d2, T, 4, 5, 33)
    A = f.A_init.tocsc().astype(np.float32); Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.set_option("debug", 1)
    try:
        out = {}
        for incr in (0, 1):
            eng.set_option("gram_incremental", incr); eng.set_option("gram_i8", 1 if incr else 0)      # (the reference: the direct Gram on the fp64 pipe)
            res = []
            for r_, nn in ((5, None), (8, None), (8, 20), (5, None)):
                eng.ring_init(0, r_, nn) if nn else eng.ring_init(0, r_)
                eng.fit_ring_model(0, A, Cm)
                w = eng.ring_csr(0).data.astype(np.float64)
                assert np.all(np.isfinite(w))
                res.append(w)
            out[incr] = res
        for a, b in zip(out[0], out[1]):
            assert a.shape == b.shape and np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(a)
    finally:
        eng.set_option("debug", 0); eng.set_option("gram_incremental", 1); eng.set_option("gram_i8", 1)


def _ar1_traces(K, T, g=0.95, sn=0.3, seed=5, rate=0.01, amp=1.5):
    rng = np.random.default_rng(seed)
    Y = np.zeros((K, T), np.float32)
    for k in range(K):
        s = (rng.random(T) < rate) * amp
        c = np.zeros(T)
        for t in range(T):
            c[t] = (g * c[t - 1] if t else 0.0) + s[t]
        Y[k] = c + 0.5 + sn * rng.standard_normal(T)
    return Y


@pytest.mark.parametrize("T", [3999, 4096])
def test_deconv_many_pools_takes_the_global_fallbacks(eng, T):
    """smin = 0 (non-negativity only): a noisy trace keeps ~T/3 pools, far more than the LDS mirror of the pool stack, the LDS copies of the
    update_g sums and the staged warm pass hold at this T, so k_deconv runs its global-memory fallbacks and the dense-event (one-sample)
    stretches of the wave pass.  With smin = 0 the problem is convex and the active set unique: the oracle's c is matched closely.
    Odd T also exercises the unpaired last Welch segment and the order statistics at odd length."""
    import oasis_oracle as oo
    Y = _ar1_traces(3, T)
    opts = dict(type="ar1", method="foopsi", smin=0.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Y, opts)
    Cr, Crawr, Sr, parsr, snr = oo.deconvTemporal(Y.astype(np.float64), smin=0.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    # observed (round 3, fminbnd without FMA contraction): gamma 1.9e-8, traces 2.9e-8, spike counts equal
    assert np.allclose(sng, snr, rtol=5e-6)
    assert np.allclose(parsg, parsr, atol=2e-6), (parsg, parsr)
    for k in range(Y.shape[0]):
        assert (Sr[k] > 0).sum() > 480                                        # the regime the test is about (LDS holds <= ~250 / 512 pools here)
        assert rel(Cg[k], Cr[k]) <= 5e-6, (k, rel(Cg[k], Cr[k]))
        assert rel(Crawg[k], Crawr[k]) <= 5e-6
        assert abs((Sg[k] > 0).sum() - (Sr[k] > 0).sum()) <= 1


def test_deconv_without_the_optimisation_loops(eng):
    """optimize_pars = optimize_b = false: one cold pass per trace, nothing else (deconvolveCa.m:120-122 with both flags off)"""
    import oasis_oracle as oo
    Y = _ar1_traces(4, 2500, seed=9)
    opts = dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=False, optimize_b=False, max_tau=100.0)
    Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Y, opts)
    Cr, Crawr, Sr, parsr, snr = oo.deconvTemporal(Y.astype(np.float64), smin=-5.0, optimize_pars=False, optimize_b=False, max_tau=100.0)
    assert np.allclose(sng, snr, rtol=5e-6) and np.allclose(parsg, parsr, atol=2e-6)
    for k in range(Y.shape[0]):
        assert rel(Cg[k], Cr[k]) <= 1e-6, (k, rel(Cg[k], Cr[k]))
        assert abs((Sg[k] > 0).sum() - (Sr[k] > 0).sum()) <= 1


@pytest.mark.parametrize("T", [18000, 24000, 50000, 100000])
def test_deconv_long_traces(eng, T):
    """T = 18000: trace (72 KB) + scratch (72 KB) is the largest image k_deconv<false> keeps in LDS; T = 24000 runs k_deconv<true> (trace and
    output staging in global memory, LDS = the Welch transform's scratch); T = 50000 (round 4: nfft = 16384) also keeps the Welch twiddle / window tables in
    global memory, LDS = re | im alone; T = 100000 (round 6: nfft = 32768) splits the transform into the 16384-point transforms of the even and the odd samples,
    one of them in LDS at a time.  Beyond nfft = 32768 (T > 147456) the call is refused (deconvolveCa.m:61 / GetSn.m:33 take any T)."""
    import oasis_oracle as oo
    from cnmf_e_amd._lib import CnmfeError
    Y = _ar1_traces(2, T, seed=21, rate=0.004)
    opts = dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Y, opts)
    Cr, Crawr, Sr, parsr, snr = oo.deconvTemporal(Y[:1].astype(np.float64), smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    print("long trace T=%d: sn %.2e gamma %.2e C %.2e Craw %.2e spikes %d/%d" % (T, abs(sng[0] / snr[0] - 1), abs(parsg[0] - parsr[0]), rel(Cg[0], Cr[0]), rel(Crawg[0], Crawr[0]),
                                                                               (Sg[0] > 0).sum(), (Sr[0] > 0).sum()))
    assert np.allclose(sng[0], snr[0], rtol=2e-5) and abs(parsg[0] - parsr[0]) < 2e-6
    assert rel(Cg[0], Cr[0]) <= 2e-5 and rel(Crawg[0], Crawr[0]) <= 2e-5
    assert abs((Sg[0] > 0).sum() - (Sr[0] > 0).sum()) <= 1
    with pytest.raises(CnmfeError):
        eng.deconv_temporal(np.zeros((1, 150000), np.float32), opts)


@pytest.mark.parametrize("T", [24000, 50000, 100000])
def test_get_sn_of_long_recordings(eng, T):
    """per-pixel GetSn (update_spatial_parallel.m:191-194, Sources2D.m:328-379 -> GetSn.m:33-47) beyond the 20400 frames a trace + its Welch transform fit the
    LDS with: the segments are read out of the interleaved video, the transform alone sits in LDS (T = 50000: nfft = 16384, twiddles in global memory; T = 100000:
    nfft = 32768 as two 16384-point halves).  Both
    flavours -- the raw video (estimate_noise) and the background-subtracted one (update_sn) -- against the oracle's GetSn on the same rows."""
    import oasis_oracle as oo
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2, r = 16, 14, 3
    f = synth.make_factors(d1, d2, T, 1, 71, gSig=1.5, gSiz=5, min_sep=4)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
    video.upload_from_full(Y)
    eng.ring_init(0, r)
    raw = eng.estimate_noise(0, T)
    ref_raw = np.array([oo.GetSn(Y[:, m].astype(np.float64)) for m in range(0, d1 * d2, 7)])
    assert np.max(np.abs(raw[::7] - ref_raw) / ref_raw) <= 2e-4
    Ysig = eng.residual(0, None, None, want=True)
    got = eng.get_sn(0)
    ref = np.array([oo.GetSn(Ysig[:, m].astype(np.float64)) for m in range(0, d1 * d2, 7)])
    assert np.max(np.abs(got[::7] - ref) / ref) <= 2e-4


def test_data_plane_uint16_tiff_to_device(eng, tmp_path):
    """a uint16 multi-page TIFF goes up block by block in its own element type (converted on the device); the resident blocks are the
    ones upload_from_full of the float video gives: same Ymean, same residual export"""
    from PIL import Image
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2, T, r = 40, 36, 130, 5
    rng = np.random.default_rng(11)
    vol = rng.integers(100, 3000, size=(T, d1, d2)).astype(np.uint16)
    tif = str(tmp_path / "rec.tif")
    pages = [Image.fromarray(vol[t]) for t in range(T)]
    pages[0].save(tif, save_all=True, append_images=pages[1:])
    v = PatchedVideo(d1, d2, T, [20, 18], r, eng)
    v.upload_from_tiff(tif, chunk=32)
    Y_td = vol.transpose(0, 2, 1).reshape(T, d1 * d2).astype(np.float64)
    for idx in v.owned:
        ym = eng.ymean(v.pid[idx])
        assert np.allclose(ym, Y_td[:, v.block_pix[idx]].mean(axis=0), rtol=1e-6)
        eng.ring_init(v.pid[idx], r)
        out = eng.residual(v.pid[idx], None, None, want=True)                   # fresh ring (uniform mean), b0 = 0
        assert out.shape == (T, v.patch_pix[idx].size) and np.isfinite(out).all()


def test_data_plane_hdf5_files_to_device(eng):
    """the reference's blocked mat_data file (uint16 blocks, deflate) and an int16 5-D .hdf5 recording go up block by block; the resident blocks have
    the means of the video's rectangles (fixtures written by h5py: tests/golden/make_h5_fixtures.py)"""
    from cnmf_e_amd import h5io
    from cnmf_e_amd.sources2d import PatchedVideo, mat_data_info
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    try:
        h5io.lib()
    except RuntimeError as e:
        pytest.skip(str(e))
    Y = np.load(os.path.join(gold, "h5_video.npy")).astype(np.float64)     # d1 x d2 x T
    info = mat_data_info(os.path.join(gold, "h5_mat_data.mat"))
    d1, d2, T = info["dims"]
    Y_td = Y.reshape(d1 * d2, T, order="F").T
    for loader in (lambda v: v.upload_from_mat_data(os.path.join(gold, "h5_mat_data.mat"), chunk=16),
                   lambda v: v.upload_from_hdf5(os.path.join(gold, "h5_recording5d.hdf5"), chunk=13)):
        v = PatchedVideo(d1, d2, T, info["patch_dims"], info["w_overlap"], eng)
        loader(v)
        for idx in v.owned:
            assert np.allclose(eng.ymean(v.pid[idx]), Y_td[:, v.block_pix[idx]].mean(axis=0), rtol=1e-6), idx


def test_deconv_degenerate_traces_terminate_with_finite_output(eng):
    """all-zero, constant, single-spike and monotone traces: no stable AR(1) estimate / zero noise level / NaN intermediates must neither hang
    the per-trace kernel nor leak NaN or Inf into C and S (deconvolveCa.m:84-89,206)"""
    T = 400
    Y = np.zeros((5, T), np.float32)
    Y[1] = 3.25
    Y[2, 100] = 10.0
    Y[3] = np.linspace(0, 5, T, dtype=np.float32)
    Y[4] = _ar1_traces(1, T, seed=2)[0]
    opts = dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Y, opts)
    for a in (Cg, Crawg, Sg, parsg, sng):
        assert np.isfinite(a).all()
    assert np.all(Sg >= 0) and (Sg[4] > 0).any()


def test_stitch_temporal_through_the_abi_with_rccl(eng, monkeypatch):
    """cnmfe_stitch_begin / _add / _temporal (update_temporal_parallel.m:264-286): two patches' traces accumulated on the device, the exchange
    taken through RCCL itself (ncclCommInitAll + ncclAllReduce on this one GPU: CNMFE_STITCH_RCCL=1), division, row minima, binding --
    against the NumPy statement of the same lines; then the bound result is usable as (NULL, CNMFE_BOUND)."""
    import ctypes as C
    from cnmf_e_amd import _lib as L, synth
    from cnmf_e_amd.sources2d import PatchedVideo
    monkeypatch.setenv("CNMFE_STITCH_RCCL", "1")
    d1, d2, T, K, r = 44, 40, 300, 6, 5
    f = synth.make_factors(d1, d2, T, K, 31, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, [22, 40], r, eng)                 # two patches stacked vertically
    video.upload_from_full(Y)
    acc = np.zeros((K, T)); aat = np.zeros(K)
    L.check(L.lib.cnmfe_stitch_begin(eng._ctx, K, T))
    for idx in video.owned:
        pid = video.pid[idx]
        eng.ring_init(pid, r)
        eng.fit_ring_model(pid, None, None)
        eng.residual(pid, None, None)
        A_p = f.A_init.tocsr()[video.patch_pix[idx]].tocsc().astype(np.float32)
        ind = np.nonzero(np.asarray(A_p.sum(axis=0)).ravel() > 0)[0]
        _, Craw, aa = eng.hals_temporal(pid, A_p[:, ind], f.C_init[ind], 3, want_C=False)      # (host copy only for the reference below)
        eng.stitch_add(ind)
        acc[ind] += Craw.astype(np.float64) * aa[:, None]; aat[ind] += aa
    aat[aat == 0] = 1
    ref = acc / aat[:, None]
    ref = ref - ref.min(axis=1, keepdims=True)
    out = np.empty((K, T), dtype=np.float32)
    ctxs = (L.c_ctx * 1)(eng._ctx)
    L.check(L.lib.cnmfe_stitch_temporal(ctxs, 1, 1, out.ctypes.data_as(L.f32p), L.ROWMAJOR))
    assert rel(out, ref) <= 1e-6, rel(out, ref)
    # the stitched matrix is now the context's bound trace matrix
    pid = video.pid[video.owned[0]]
    A_p = f.A_init.tocsr()[video.patch_pix[video.owned[0]]].tocsc().astype(np.float32)
    eng._bound = out
    c_bound = eng.hals_temporal(pid, A_p, out, 2)
    eng.bind_traces(None)
    c_plain = eng.hals_temporal(pid, A_p, out.copy(), 2)
    assert np.array_equal(c_bound[1], c_plain[1])
    with pytest.raises(L.CnmfeError):
        eng.stitch_add(np.arange(2))                                     # no open accumulator any more


@pytest.mark.parametrize("maxN", [3, 20, 27])
def test_nnls_crowded_pixels(eng, maxN):
    """nnls_spatial.m:35-37 puts no bound on how many search masks cover a pixel; only the passive set is bounded (maxIter = maxN, :76,:86).
    30 neurons share one neighbourhood, so the central pixels lie in up to 30 masks (more than the demo's maxN = 20)."""
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2, T, K, r = 32, 32, 500, 30, 5
    rng = np.random.default_rng(41)
    f = synth.make_factors(d1, d2, T, K, 41, gSig=2.5, gSiz=11, min_sep=1)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
    video.upload_from_full(Y)
    pid = video.pid[video.order[0]]
    eng.ring_init(pid, r)
    eng.fit_ring_model(pid, None, None)
    ysig = eng.residual(pid, None, None, want=True).T.astype(np.float64)
    yy, xx = np.mgrid[:d1, :d2]
    IND = np.zeros((d1 * d2, K), bool)
    for k in range(K):                                    # a disc of radius 9..12 around a centre near the middle of the FOV
        cy, cx = rng.uniform(12, 20, 2)
        IND[:, k] = (((yy - cy) ** 2 + (xx - cx) ** 2) <= rng.uniform(9, 12) ** 2).ravel(order="F")
    assert IND.sum(axis=1).max() > 20
    INDs = sp.csc_matrix(IND)
    C = f.C_true.astype(np.float32) if hasattr(f, "C_true") else f.C_init
    A0 = sp.csc_matrix((d1 * d2, K), dtype=np.float32)
    got = eng.update_spatial(pid, "nnls", A0, C, INDs, f.sn, maxN).toarray()
    ref = orc.nnls_spatial(ysig, A0, C, INDs, maxN)
    assert ((ref != 0).sum(axis=1) <= maxN).all()
    # a variable that leaves the passive set keeps s + a*(mu - s) with a = s/(s - mu) (:102-106): zero up to rounding, i.e. 0 or ~1e-17 of the
    # pixel's weights depending on the order of operations -- the only entries whose support may differ
    mism = (got != 0) != (ref != 0)
    assert np.all(np.abs(np.where(mism, got, 0)) + np.abs(np.where(mism, ref, 0)) <= 1e-9 * np.abs(ref).max()), np.abs(got - ref)[mism].max()
    assert mism.sum() <= 0.01 * (ref != 0).sum(), mism.sum()
    assert rel(got, ref) <= 5e-6, rel(got, ref)


@pytest.mark.parametrize("pdims,r,T,thr", [(None, 4, 3072, 1.5), ([24, 40], 5, 1024, 0.5), (None, 7, 1024, 2.0)])
def test_outlier_branch_with_exact_decisions(eng, pdims, r, T, thr):
    """fit_ring_model.m:50-67 (thresh_outlier not NaN): clip the patch rows of Bf at W_old*Bf + thresh*sn, keep the frames with few outliers.
    The `>` of :53 is a discrete decision, so this video is built to make the engine's fp32 Bf EXACT (8-bit integers, T a multiple of 1024:
    the pixel means and the centred values fit a float32 mantissa; no footprints) -- every decision then matches the float64 oracle and W
    agrees as tightly as in the linear branch.  Case 1 selects frames (24 offsets -> nmax = 2400 < T), the others keep them all; case 2 has
    patches smaller than their blocks (ind_patch).  Two fits: ring pattern as W_old, then the fitted weights as W_old."""
    import cnmfe_oracle as orc
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2 = 48, 40
    rng = np.random.default_rng(5)
    base = rng.integers(100, 200, (1, d1 * d2))
    drift = (40 * np.sin(np.arange(T) / 37.0)[:, None] * rng.uniform(0.5, 1.5, (1, d1 * d2))).astype(np.int64)
    Y = (base + drift + rng.integers(0, 60, (T, d1 * d2))).astype(np.float32)
    Y[rng.random((T, d1 * d2)) < 0.02] += 90.0                              # sparse positive events: what the branch is meant to clip
    assert Y.min() >= 0 and Y.max() < 512 and T % 1024 == 0
    sn = rng.uniform(5.0, 15.0, d1 * d2).astype(np.float32)
    video = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, eng)
    video.upload_from_full(Y)
    for idx in video.order:
        pid, bp = video.pid[idx], video.block_pix[idx]
        ip = np.zeros(bp.size, dtype=bool); ip[video.ind_patch[idx]] = True
        eng.ring_init(pid, r)
        eng.set_noise(pid, sn[bp])
        Yb = Y[:, bp].T.astype(np.float64)
        for fit in range(2):
            W_old = eng.ring_csr(pid)
            _, info = eng.fit_ring_model(pid, None, None, thresh_outlier=thr)
            assert info["frame_stride"] == 1 and info["first_run"] == (fit == 0)
            Wr, b0r = orc.fit_ring_model(Yb, None, None, W_old, thr, sn[bp][ip], ip, True)
            Wg = eng.ring_csr(pid); Wr = sp.csr_matrix(Wr); Wr.sort_indices()
            assert np.array_equal(Wg.indices, Wr.indices)
            e = rel(Wg.data, Wr.data)
            assert e <= 2e-6, (idx, fit, e)
    with pytest.raises(Exception, match="cnmfe_set_noise"):                 # a fresh patch without noise levels: CNMFE_ESTATE
        v2 = PatchedVideo(d1, d2, T, [d1, d2], r, eng); v2.upload_from_full(Y)
        eng.ring_init(v2.pid[v2.order[0]], r)
        eng.fit_ring_model(v2.pid[v2.order[0]], None, None, thresh_outlier=thr)


def test_lazy_traces_keep_their_iteration(eng):
    """The temporal update hands C back as a LazyHostTraces (device binding + copy into pinned memory on a second stream).  Two runs in lockstep on
    two contexts: one reads C after every iteration, the other keeps the lazy objects of three iterations unread until the very end -- each must
    still hold ITS iteration's values (own pinned buffer; the next stitch waits for the copy before it overwrites the bound matrix)."""
    from cnmf_e_amd.engine import Engine, LazyHostTraces
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    from cnmf_e_amd import synth
    d1, d2, T, K, r = 40, 36, 300, 5, 5
    f = synth.make_factors(d1, d2, T, K, 31, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    eng2 = Engine(0)
    try:
        runs = []
        for e in (eng, eng2):
            v = PatchedVideo(d1, d2, T, [20, 18], r, e)
            v.upload_from_full(Y)
            runs.append(Sources2D(v, Options(ring_radius=r, maxIter=3), f.A_init, f.C_init, f.sn))
        eager, lazy, kept = [], runs[1], []
        for it in range(3):
            for s in runs:
                s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
            eager.append(np.asarray(runs[0].C).copy())
            assert isinstance(lazy.C, LazyHostTraces) and not lazy.C._ready
            kept.append(lazy.C)
        for it in range(3):
            assert np.array_equal(np.asarray(kept[it]), eager[it]), it
        assert not np.array_equal(eager[0], eager[2])
        # arithmetic / reductions / indexing go through the host copy
        c = kept[2]
        assert c.shape == eager[2].shape and np.allclose(c.mean(axis=1), eager[2].mean(axis=1)) and np.array_equal(c[1], eager[2][1])
        assert np.array_equal(c - eager[2], np.zeros_like(eager[2])) and float(c.max()) == float(eager[2].max())
    finally:
        eng2.close()


def test_more_footprints_on_a_ring_than_the_lds_slots_hold(eng):
    """A pixel whose ring touches more than 32 footprints of A_prev (update_spatial_parallel.m:162-166 knows no such limit): until round 5 CNMFE_EUNSUPPORTED,
    now the (W * A_prev) row of such a pixel continues where the table lies (k_ring_wa: 32 slots in registers / LDS, the rest in the table's own rows, as
    many rows as there are footprints, up to 256) and the sweep's per-pixel fallback reads it from there.  Checked against the oracle's residual."""
    import cnmfe_oracle as orc
    d1, d2, T, r = 48, 48, 64, 5
    f, Y, video = _video(eng, d1, d2, T, 3, r, 7)
    eng.ring_init(0, r)
    eng.fit_ring_model(0, None, None)
    K = 44                                                       # 44 one-pixel footprints on the ring of pixel (24, 24)
    rs, cs = np.array(orc_nhood(r))
    rows = [(24 + int(cs[i % rs.size]) + (i // rs.size)) * d1 + 24 + int(rs[i % rs.size]) for i in range(K)]
    A = sp.csc_matrix((np.ones(K, np.float32), (rows, np.arange(K))), shape=(d1 * d2, K))
    Cm = np.random.default_rng(0).random((K, T)).astype(np.float32)
    W = eng.ring_csr(0).astype(np.float64)
    assert int(np.max(np.diff((abs(W) @ abs(A)).tocsr().indptr))) > 32          # (some ring does meet more than 32 of them)
    b0f = Y.astype(np.float64).mean(axis=0).astype(np.float32)   # b0 = the pixels' means, as fp32 (what set_b0 takes): the constant term Ymean - b0 the sweep adds in
    eng.set_b0(0, b0f)                                           # fp32 then stays small (at b0 = 0 it is ~1000 and its rounding, 3e-5, would hide what is checked here)
    b0 = b0f.astype(np.float64)
    Yb = Y.T.astype(np.float64)                                  # d x T: one patch = the field of view
    for Ksel in (K, 20, K):                                      # with more, with fewer, with more again (the table's rows grow and are reused)
        out = eng.residual(0, A[:, :Ksel], Cm[:Ksel], want=True)
        ref = orc.residual_ysig(Yb, A[:, :Ksel].astype(np.float64), Cm[:Ksel], W, b0, np.ones(d1 * d2, dtype=bool))
        # in units of the VIDEO (~1300 here: the sweep's fp32 arithmetic on the centred video leaves up to 2.3e-7 of that, whatever K -- scripts/dbg_cap.py); a footprint
        # term lost beyond slot 32 would be ~4e-6 of it, on the crowded pixels
        e = np.abs(out.T - ref).max(axis=1) / np.abs(Y).max()
        crowded = np.diff((abs(W) @ abs(A[:, :Ksel])).tocsr().indptr) > 32
        assert e.max() <= 5e-7 and (not crowded.any() or e[crowded].max() <= 1e-7), (Ksel, e.max(), e[crowded].max() if crowded.any() else None)


def orc_nhood(r):
    import cnmfe_oracle as orc
    rs, cs = orc.get_nhood(r)
    return np.ravel(rs), np.ravel(cs)


def test_bench_two_ranks_self_spawned_on_one_device():
    """`python bench.py --gpus 2 --config c4tiny` with NO launcher in the environment: bench.py starts its two ranks itself.  On this one-GPU box both
    ranks share cuda:0 and talk over gloo (CNMFE_BENCH_ONE_DEVICE=1: RCCL refuses two ranks on one device); what is checked is the launcher, the
    sharded c4 code path (2 x 2 patches over 2 ranks, all-gather of A, all-reduce of the stitch) and the line's n_gpus / rccl_ranks."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CNMFE_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo" and line["scaling"] == "strong"
    assert line["value"] > 0 and "c4tiny" in line["config"]["workload"]


def test_bench_eight_ranks_full_size_c4_on_one_device():
    """BASELINE configs[3] at FULL size through the code path of an 8-GPU run: `python bench.py --gpus 8 --config c4` spawns its eight ranks, each owns two of the
    sixteen 128 x 128 patches of the 512 x 512 x 10000 video; on this one-GPU box they share cuda:0 and talk over gloo (CNMFE_BENCH_ONE_DEVICE=1).  What is checked:
    the eight-rank run completes (all-gather of the spatial rows, sharded post-processing, the all-reduce of the 500 x 10000 stitch, lazy b0 reductions), reports
    8 ranks, and every rank's traces agree with the one-rank run of the same configuration (the seconds of eight processes sharing a GPU mean nothing)."""
    import json, subprocess
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 120e9:
        pytest.skip("needs a 128+ GB GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CNMFE_BENCH_ONE_DEVICE"] = "1"; env["CNMFE_BENCH_R1"] = "0"
    outs = {}
    for n in (8, 1):
        e = dict(env, CNMFE_BENCH_DUMP=os.path.join("/tmp", "cnmfe_c4_n%d.npy" % n))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", "c4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                           env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        line = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == n and line["rccl_ranks"] == n and line["value"] > 0 and "c4: 512x512x10000" in line["config"]["workload"]
        if n > 1:
            assert line["backend"] == "gloo" and line["scaling"] == "strong"
        outs[n] = np.load(e["CNMFE_BENCH_DUMP"])
        os.remove(e["CNMFE_BENCH_DUMP"])
    assert outs[8].shape == outs[1].shape == (500, 10000)
    assert rel(outs[8], outs[1]) <= 1e-6, rel(outs[8], outs[1])          # (the stitch adds the patches' contributions in another order across ranks: fp32 re-association)


def test_deconv_temporal_on_the_bound_matrix_equals_the_host_path(eng):
    """cnmfe_deconv_temporal_bound: the stitched C_raw deconvolved where it lies (C becomes the bound matrix, the five outputs arrive lazily in pinned
    memory) against cnmfe_deconv_temporal on the same values through host arrays -- bit for bit; afterwards a call that is handed the returned C passes
    (NULL, CNMFE_BOUND) and reads the deconvolved traces."""
    from cnmf_e_amd.engine import LazyHostTraces
    from cnmf_e_amd import _lib as L
    Y = _ar1_traces(6, 2100, seed=13)
    opts = dict(type="ar1", method="foopsi", smin=-5.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    Cr, Crawr, Sr, parsr, snr = eng.deconv_temporal(Y.copy(), opts)
    eng.bind_traces(Y)
    C, Craw, S, pars, sn = eng.deconv_temporal_bound(opts)
    assert isinstance(C, LazyHostTraces) and eng._bound is C
    assert np.array_equal(np.asarray(C), Cr) and np.array_equal(np.asarray(Craw), Crawr) and np.array_equal(np.asarray(S), Sr)
    assert np.array_equal(np.asarray(pars)[0], parsr) and np.array_equal(np.asarray(sn)[0], snr)
    ptr, order, _ = eng._targs(C, 6, 2100)
    assert ptr is None and order == L.BOUND
    # the bound matrix now holds C: a second deconvolution of the bound matrix deconvolves C, not Y
    C2 = np.asarray(eng.deconv_temporal_bound(opts)[0])
    assert np.array_equal(C2, eng.deconv_temporal(Cr.copy(), opts)[0])


@pytest.mark.parametrize("dims", [(64, 64), (70, 50)])
def test_deferred_footprint_term_through_update_spatial(eng, dims):
    """r1_defer (default, ring radius 15): a residual with a footprint term runs its sweep WITHOUT the term (the duo-role kernel) and leaves the term
    pending; cnmfe_update_spatial takes it in through its projection, U += (W A_prev)(Cc_prev Cc'), cnmfe_hals_temporal through A' as before, and a
    consumer of Ysig itself (GetSn, an exported residual) folds it in first.  Against the in-sweep term (r1_defer = 0) on a tile-aligned patch and on
    one with partial tiles; HALS and NNLS read the same U."""
    d1, d2 = dims
    T, r, K = 160, 15, 6
    f, Y, video = _video(eng, d1, d2, T, K, r, 21)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32)
    Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    eng.fit_ring_model(0, A, Cm)
    prev = np.array([0, 3, 4])                                  # the "halo neurons" of the term
    IND = sp.csc_matrix((A.toarray() > 0) | (np.roll(A.toarray(), 1, axis=0) > 0)).astype(np.float32)
    eng.set_option("r1_virtual", 0)                             # (this test is about the SWEPT residual's deferred term; the sweep-free path: test_gpu_virtual.py)
    def run(defer, alg):
        eng.set_option("r1_defer", defer)
        eng.set_b0(0, eng.b0(0))                                # invalidates the resident Ysig: the next residual is a sweep
        eng.profile(True); eng.profile_reset()
        eng.residual(0, A[:, prev], Cm[prev])
        Anew = eng.update_spatial(0, alg, A, Cm, IND, param=20 if alg == "nnls" else 3)
        tab = eng.profile_table()
        eng.residual(0, A, Cm)                                  # the temporal update's request: pending in both modes
        c = eng.hals_temporal(0, A, Cm, 3)
        sn = eng.get_sn(0)
        y = eng.residual(0, A, Cm, want=True)
        eng.profile(False)
        return Anew, tab, c, sn, y
    try:
        for alg in ("hals", "nnls"):
            (a0, t0, c0, sn0, y0), (a1, t1, c1, sn1, y1) = run(0, alg), run(1, alg)
            calls = lambda t, k: t.get(k, {"calls": 0})["calls"]
            assert calls(t0, "spatial_term_fold") == 0 and calls(t1, "spatial_term_fold") == 1 and calls(t1, "spatial_term_gram") == 1, (t0, t1)
            assert calls(t1, "residual_delta") == 0                                 # the deferred term never went through Ysig before the spatial update
            d0, d1_ = a0.toarray(), a1.toarray()
            assert np.array_equal(d0 > 0, d1_ > 0) or np.abs(d0 - d1_).max() <= 1e-5 * np.abs(d0).max()
            assert np.abs(d0 - d1_).max() <= 1e-5 * np.abs(d0).max()
            for x, z in zip(c0, c1):
                assert np.abs(x - z).max() <= 1e-5 * max(1.0, np.abs(x).max())
            assert np.allclose(sn0, sn1, rtol=1e-5)
            assert np.abs(y0 - y1).max() <= 2e-6 * np.abs(y0).max()
    finally:
        eng.set_option("r1_defer", 1); eng.set_option("r1_virtual", 1)


def test_two_frame_strides_keep_their_tables(eng):
    """The frame stride of a fit follows pmax, the largest number of positive weights of W_old (fit_ring_model.m:60,84-87): a recording whose
    T / (100 pmax) sits near an integer alternates between two strides from fit to fit.  The engine keeps the video's covariance table for BOTH
    (a second slot, filled when a second stride turns up): three fits at strides 2, 1, 2 build two tables, not three, and every W equals the
    direct fp64 Gram's."""
    d1, d2, T, r = 44, 40, 1200, 5
    f, Y, video = _video(eng, d1, d2, T, 4, r, 41)
    eng.ring_init(0, r)
    A = f.A_init.tocsc().astype(np.float32); Cm = np.ascontiguousarray(f.C_init, dtype=np.float32)
    Wcsr = eng.ring_csr(0)
    def w_with(npos):
        """ring weights with exactly min(npos, row length) positive entries per row (the rest negative): pmax = npos"""
        v = -np.ones(Wcsr.nnz, np.float32) * 0.01
        for i in range(Wcsr.shape[0]):
            a, b = Wcsr.indptr[i], Wcsr.indptr[i + 1]
            v[a:a + min(npos, b - a)] = 0.02
        return v
    out = {}
    try:
        for incr in (0, 1):
            eng.set_option("gram_incremental", incr); eng.set_option("gram_i8", 1 if incr else 0)      # (the reference: the direct Gram on the fp64 pipe)
            eng.profile(True); eng.profile_reset()
            res = []
            for npos, stride in ((6, 2), (12, 1), (6, 2)):
                eng.ring_set_values(0, w_with(npos))
                _, info = eng.fit_ring_model(0, A, Cm)
                assert info["frame_stride"] == stride and info["pmax"] == npos, info
                res.append(eng.ring_csr(0).data.astype(np.float64))
            tab = eng.profile_table(); eng.profile(False)
            out[incr] = res
            if incr:
                built = sum(tab[k]["calls"] for k in ("bg_gram_f64", "bg_gram_i8") if k in tab)   # (the video's table: int8 digits since round 5, fp64 for long recordings)
                assert built == 2, tab                                                   # one table per stride; the third fit found its own again
        for a, b in zip(out[0], out[1]):
            assert np.all(np.isfinite(a)) and np.linalg.norm(a - b) <= 2e-6 * np.linalg.norm(a)
    finally:
        eng.set_option("gram_incremental", 1); eng.set_option("gram_i8", 1)


@pytest.mark.parametrize("deconv", [False, True])
def test_temporal_jobs_across_patches_equal_the_per_patch_updates(deconv):
    """Several patches per context: cnmfe_hals_temporal_job sets every patch's temporal update up, cnmfe_temporal_jobs_sweep runs level l of ALL
    jobs in one launch, cnmfe_stitch_add_job adds each to the stitch.  The same kernels on the same operands as one cnmfe_hals_temporal[_deconv]
    per patch: two full iterations agree bit for bit (with and without the in-sweep deconvolution)."""
    from cnmf_e_amd import synth
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 96, 96, 500, 14, 5
    f = synth.make_factors(d1, d2, T, K, 17, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    res = []
    for jobs in (False, True):
        e = Engine(0)
        try:
            if not jobs:
                e.supports_temporal_jobs = False              # (instance attribute: this run takes the per-patch calls)
            v = PatchedVideo(d1, d2, T, [48, 48], r, e)
            v.upload_from_full(Y)
            s = Sources2D(v, Options(ring_radius=r, maxIter=3, deconv_flag=deconv), f.A_init, f.C_init, f.sn)
            if jobs:
                e.profile(True)
            for _ in range(2):
                s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
            if jobs:
                tab = e.profile_table()
                lv = tab["temporal_hals_deconv_level" if deconv else "temporal_hals_level"]["calls"]
            res.append((np.asarray(s.C).copy(), np.asarray(s.C_raw).copy(), s.A.copy()))
        finally:
            e.close()
    (c0, r0, a0), (c1, r1, a1) = res
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1) and (a0 != a1).nnz == 0
    assert np.isfinite(c1).all() and np.abs(c1).max() > 0
    assert lv <= 2 * 3 * 8, lv                                 # levels x maxIter launches per update for ALL four patches together, not per patch
