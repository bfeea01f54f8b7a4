"""search_method = 'dilate' (determine_search_location.m:89-98 + threshold_components.m) and spatial_constraints.circular
(circular_constraints.m): the product's scipy.ndimage implementation (cnmf_e_amd/hostops.py) against the oracle's explicit-shift /
flood-fill restatement, function by function and through two iterations of the three update methods (NumPy fake engine)."""
import numpy as np
import scipy.sparse as sp
import pytest

import cnmfe_oracle as orc
from cnmf_e_amd import hostops, synth


def _footprints(d1, d2, K, seed):
    """blobs with specks, a second lobe, holes and a neuron at the FOV border: everything the morphology has to decide about"""
    rng = np.random.default_rng(seed)
    f = synth.make_factors(d1, d2, 10, K, seed, gSig=2.0, gSiz=9, min_sep=4)
    A = f.A_true.toarray().astype(np.float64)
    for k in range(K):
        pix = rng.integers(0, d1 * d2, 8)
        A[pix, k] += rng.uniform(0.02, 0.5, 8)
        nz = np.nonzero(A[:, k])[0]
        A[rng.choice(nz, min(3, nz.size), replace=False), k] = 0.0
    A[:, 1] += 0.7 * np.roll(A[:, 0], 3 * d1 + 2)
    yy, xx = np.mgrid[:d1, :d2]
    A[:, 2] = np.exp(-((yy - 0.5) ** 2 + (xx - 1.0) ** 2) / 8.0).reshape(-1, order="F") * (rng.random(d1 * d2) > 0.1)
    A[:, 3] = 0.0                                                          # an empty component
    return A


@pytest.mark.parametrize("seed,dims", [(1, (40, 36)), (2, (31, 45))])
def test_threshold_components_and_dilate_match_oracle(seed, dims):
    d1, d2 = dims
    A = _footprints(d1, d2, 7, seed)
    for nb in (1, 0):
        got = hostops.threshold_components(sp.csc_matrix(A), d1, d2, nb=nb).toarray()
        ref = orc.threshold_components(A, d1, d2, nb=nb)
        assert np.array_equal(got, ref)
    assert np.array_equal(hostops.strel_disk(4), orc.strel_disk(4)) and hostops.strel_disk(3).sum() == 29
    for se in (orc.strel_disk(4), orc.strel_disk(3), np.ones((3, 3), bool)):
        got = hostops.search_location_dilate(sp.csc_matrix(A), d1, d2, se).toarray()
        ref = orc.determine_search_location_dilate(A, d1, d2, se)
        assert np.array_equal(got, ref)
        assert not ref[:, 3].any() and ref[:, 0].sum() > (A[:, 0] > 0).sum() * 0.8     # the empty column stays empty, the others grow


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_circular_constraints_match_oracle(seed):
    d1, d2 = 40, 36
    A = _footprints(d1, d2, 7, seed)
    changed = 0
    for k in range(A.shape[1]):
        img = A[:, k].reshape(d1, d2, order="F")
        got = hostops.circular_constraints(img)
        ref = orc.circular_constraints(img)
        assert np.array_equal(got, ref), k
        changed += int(not np.array_equal(ref, img))
    assert changed >= 5
    line = np.zeros((d1, d2)); line[7, 3:9] = 1.0                          # a one-pixel-high bounding box is returned as it is (:17-19)
    assert np.array_equal(hostops.circular_constraints(line), line) and np.array_equal(orc.circular_constraints(line), line)
    cols = hostops.circular_constraints_columns(sp.csc_matrix(A.astype(np.float32)), d1, d2).toarray()
    ref = np.stack([orc.circular_constraints(A.astype(np.float32)[:, k].reshape(d1, d2, order="F")).reshape(-1, order="F") for k in range(A.shape[1])], axis=1)
    assert np.allclose(cols, ref, rtol=1e-6, atol=0)


def test_methods_with_dilate_and_circular_match_oracle():
    """two iterations of update_{background,spatial,temporal}_parallel with search_method = 'dilate' (first call: strel('disk', 4, 0),
    second: strel('disk', bSiz, 0) -- update_spatial_parallel.m:56,63-65) and circular = true, on the NumPy fake engine"""
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 36, 32, 80, 5, 5
    f = synth.make_factors(d1, d2, T, K, 11, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    eng = FakeEngine()
    video = PatchedVideo(d1, d2, T, [18, 16], r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, maxIter=2, search_method="dilate", bSiz=2,
                                 spatial_constraints={"circular": True, "connected": True}), f.A_init, f.C_init, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [18, 16], r, f.A_init.astype(np.float32), f.C_init, f.sn, maxIter=2,
                            search_method="dilate", bSiz=2, circular=True)
    for it in range(2):
        for step in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"):
            if step == "update_spatial_parallel":
                s._search_location_owned()            # an extra evaluation (a prefetch, bench.py's roofline helper) must not consume options.se
            getattr(s, step)(); getattr(o, step)()
        Ag, Ar = s.A.toarray(), o.A.toarray()
        assert np.array_equal(Ag != 0, Ar != 0), it
        assert np.abs(Ag - Ar).max() <= 1e-5 * np.abs(Ar).max() and np.abs(np.asarray(s.C) - o.C).max() <= 1e-5 * np.abs(o.C).max(), it
    assert s.options.se is None and o.se is None


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_native_row_selection_equals_the_numpy_formulation(seed):
    """cnmfe_csc_select_rows (the library's host helper behind Sources2D._slice) against the index-arithmetic NumPy version of the same
    selection: `A(mask, ind)` with `ind = find(sum(A(mask,:),1) > 0)` (update_spatial_parallel.m:87-91) and `A(mask, cols)` (:96-97) --
    identical columns, pointers, row indices and values; empty candidates, empty selections, columns that only touch the mask with zeros or
    with values summing to <= 0, explicit zeros kept inside selected columns."""
    from cnmf_e_amd import sources2d as S
    rng = np.random.default_rng(seed)
    d1, d2, K = 40, 36, 30
    d = d1 * d2
    rows, cols, vals = [], [], []
    for k in range(K):
        if k % 7 == 3:
            continue                                                        # an empty column
        r0, c0 = rng.integers(0, d1 - 6), rng.integers(0, d2 - 6)
        rr, cc = np.meshgrid(np.arange(r0, r0 + 6), np.arange(c0, c0 + 6), indexing="ij")
        v = rng.standard_normal(36).astype(np.float32)
        v[rng.random(36) < 0.3] = 0.0                                       # explicit zeros
        if k % 5 == 0:
            v = -np.abs(v)                                                  # a column whose selected sum cannot be positive
        rows.append((cc * d1 + rr).ravel()); cols.append(np.full(36, k)); vals.append(v)
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(d, K), dtype=np.float32)
    A.sort_indices()
    assert A.indices.dtype == np.int32
    lut = np.full(d, -1, dtype=np.int32)
    rr, cc = np.meshgrid(np.arange(8, 30), np.arange(5, 25), indexing="ij")
    pix = np.sort((cc * d1 + rr).ravel())
    lut[pix] = np.arange(pix.size, dtype=np.int32)
    every = np.nonzero(np.diff(A.indptr) > 0)[0]
    for cand, keep_all in ((every, False), (every, True), (every[::3], False), (np.arange(K)[::2], True), (np.zeros(0, dtype=np.int64), False)):
        got = S._select_rows_native(A, lut, pix.size, cand, keep_all)
        assert got is not None, "the library's host helper did not run (int32 / float32 CSC)"
        ref = S._select_rows_numpy(A, lut, pix.size, np.asarray(cand, dtype=np.int64), keep_all)
        assert np.array_equal(got[0], ref[0])
        assert got[1].shape == ref[1].shape
        assert np.array_equal(got[1].indptr, ref[1].indptr) and np.array_equal(got[1].indices, ref[1].indices)
        assert np.array_equal(got[1].data, ref[1].data)
    # the public entry: rows_of goes through the helper and agrees with a dense selection
    ind, M = S.rows_of(A, lut, pix.size, span=(int(pix.min()), int(pix.max())))
    dense = A.toarray()[pix]
    assert np.array_equal(ind, np.nonzero(dense.sum(axis=0, dtype=np.float64) > 0)[0])
    assert np.array_equal(M.toarray(), dense[:, ind])
    # unsorted candidates are refused, not silently mis-sliced
    with pytest.raises(ValueError):
        S._select_rows_native(A, lut, pix.size, np.array([4, 2], dtype=np.int64), True)


def test_native_drop_zeros_equals_eliminate_zeros():
    """cnmfe_csc_drop_zeros (behind Engine.update_spatial's compact fetch) against scipy: the matrix without its stored zeros, and with the
    connectivity flags applied -- pointers, rows and values identical to csc(...).eliminate_zeros(); empty columns and an all-zero matrix."""
    import ctypes as C
    from cnmf_e_amd import _lib as L
    rng = np.random.default_rng(5)
    d, K = 300, 17
    M = sp.random(d, K, density=0.2, format="csc", random_state=3, dtype=np.float32)
    M.sort_indices()
    vals = M.data.copy(); vals[rng.random(vals.size) < 0.5] = 0.0
    keep = (rng.random(vals.size) < 0.8).astype(np.uint8)
    icp = M.indptr.astype(np.int64); iri = M.indices.astype(np.int32)
    for v, kp in ((vals, None), (vals, keep), (np.zeros_like(vals), None)):
        optr = np.empty(K + 1, dtype=np.int64); orow = np.empty(vals.size, dtype=np.int32); oval = np.empty(vals.size, dtype=np.float32)
        n = C.c_int64(-1)
        rc = L.lib.cnmfe_csc_drop_zeros(K, icp.ctypes.data, iri.ctypes.data, v.ctypes.data, None if kp is None else kp.ctypes.data,
                                        optr.ctypes.data, orow.ctypes.data, oval.ctypes.data, C.byref(n))
        assert rc == 0
        ref = sp.csc_matrix((v * (1 if kp is None else kp), iri.copy(), icp.copy()), shape=(d, K)); ref.eliminate_zeros()
        assert n.value == ref.nnz
        assert np.array_equal(optr, ref.indptr) and np.array_equal(orow[:n.value], ref.indices) and np.array_equal(oval[:n.value], ref.data)


@pytest.mark.parametrize("seed", [0, 3])
def test_fused_block_patch_selection_and_native_boxes(seed):
    """cnmfe_csc_select_block_patch = rows_of(block) then rows_of(patch, cols=ind) (update_temporal_parallel.m:83-91); cnmfe_csc_bbox = the NumPy boxes of _bbox_of"""
    from cnmf_e_amd import sources2d as S
    rng = np.random.default_rng(seed)
    d1, d2, K = 40, 36, 30
    d = d1 * d2
    rows, cols, vals = [], [], []
    for k in range(K):
        if k % 7 == 3:
            continue
        r0, c0 = rng.integers(0, d1 - 6), rng.integers(0, d2 - 6)
        rr, cc = np.meshgrid(np.arange(r0, r0 + 6), np.arange(c0, c0 + 6), indexing="ij")
        v = rng.standard_normal(36).astype(np.float32)
        v[rng.random(36) < 0.3] = 0.0
        if k % 5 == 0:
            v = -np.abs(v)
        rows.append((cc * d1 + rr).ravel()); cols.append(np.full(36, k)); vals.append(v)
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(d, K), dtype=np.float32)
    A.sort_indices()
    def lut_of(r0, r1, c0, c1):
        rr, cc = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing="ij")
        pix = np.sort((cc * d1 + rr).ravel())
        t = np.full(d, -1, dtype=np.int32); t[pix] = np.arange(pix.size, dtype=np.int32)
        return t, pix.size
    (tb, nb), (tp, npx) = lut_of(5, 32, 4, 28), lut_of(10, 27, 9, 23)
    every = np.nonzero(np.diff(A.indptr) > 0)[0]
    for cand in (every, every[::2], np.zeros(0, dtype=np.int64)):
        got = S._select_block_patch_native(A, tb, nb, tp, npx, cand)
        assert got is not None
        ind, Ab = S._select_rows_numpy(A, tb, nb, np.asarray(cand, dtype=np.int64), False)
        _, Ap = S._select_rows_numpy(A, tp, npx, ind, True)
        assert np.array_equal(got[0], ind)
        for M, R in ((got[1], Ab), (got[2], Ap)):
            assert M.shape == R.shape and np.array_equal(M.indptr, R.indptr) and np.array_equal(M.indices, R.indices) and np.array_equal(M.data, R.data)
    nz, rmin, rmax, cmin, cmax = S._bbox_native(A, d1)
    assert np.array_equal(nz, every)
    for j, k in enumerate(every):
        r = A.indices[A.indptr[k]:A.indptr[k + 1]]
        assert (rmin[j], rmax[j], cmin[j], cmax[j]) == ((r % d1).min(), (r % d1).max(), (r // d1).min(), (r // d1).max())
