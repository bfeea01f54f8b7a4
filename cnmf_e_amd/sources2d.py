"""Host-side mirror of the reference's ``Sources2D`` update methods on top of the HIP engine.

The reference is MATLAB (`ca_source_extraction/@Sources2D/`); there is no MATLAB in the build image,
so the host side above the C ABI is Python and keeps the reference's names, argument meaning and
error behaviour for the ONE path this project covers:

    update_background_parallel(use_parallel)            @Sources2D/update_background_parallel.m:1
    update_spatial_parallel(use_parallel, update_sn)    @Sources2D/update_spatial_parallel.m:1
    update_temporal_parallel(use_parallel, use_c_hat)   @Sources2D/update_temporal_parallel.m:1

Per-patch slicing (which neurons / pixels go to which patch, how results are stitched) is host
logic exactly as in the reference; every numerical kernel runs on the GPU through
``cnmf_e_amd.engine.Engine`` (no CPU fallback).  Patches can be sharded over ranks of a
``torch.distributed`` group (one process per GPU): background and spatial updates need no
collective on the data path, the temporal update does ONE all-reduce of ``aa .* C_raw`` and ``aa``
(update_temporal_parallel.m:269-280).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import ctypes
import os

import numpy as np
import scipy.sparse as sp

from . import _lib as L
from .engine import BoundRows, DeviceTraces, Engine

__all__ = ["Options", "PatchedVideo", "Sources2D", "distribute_geometry", "determine_search_location"]


# --------------------------------------------------------------------------------------
# options actually read by the hot path (CNMFSetParms.m)
# --------------------------------------------------------------------------------------
@dataclass
class Options:
    d1: int = 0
    d2: int = 0
    ring_radius: int = 15            # CNMFSetParms.m:106
    num_neighbors: int | None = None  # :109
    bg_ssub: int = 1                 # :107
    bg_acceleration: bool = True     # :108
    thresh_outlier: float = float("nan")   # :112
    background_model: str = "ring"   # demo_large_data_1p.m:54
    spatial_algorithm: str = "hals"  # :117  ('hals' | 'hals_thresh' | 'nnls')
    search_method: str = "ellipse"   # :48
    se: object = "disk4"             # strel('disk', 4, 0): element of the 'dilate' search; None = [] -> strel('disk', bSiz, 0)
    bSiz: int = 3                    # :35
    nb: int = 1                      # :23 (threshold_components leaves the last nb columns alone)
    nrgthr: float = 0.99             # :56
    min_size: float = 3.0            # :51
    max_size: float = 8.0            # :52
    dist: float = 3.0                # :53
    maxIter: int = 5                 # :36
    deconv_flag: bool = False        # :101
    deconv_options: dict = field(default_factory=lambda: {"type": "ar1", "method": "foopsi", "smin": -5.0,
                                                          "optimize_pars": True, "optimize_b": True, "max_tau": 100.0})   # demo_large_data_1p.m:38-43
    spatial_constraints: dict = field(default_factory=lambda: {"circular": False, "connected": True})   # :116


def _mround(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def distribute_geometry(d1, d2, patch_dims, w_overlap):
    """patch_pos / block_pos exactly as endoscope/distribute_data.m:38-100,161-173 lays them out
    (1-based inclusive [r0 r1 c0 c1]; block = patch grown by w_overlap+1, clipped)."""
    minw = 2 * w_overlap + 3
    pd = np.atleast_1d(np.asarray(patch_dims, dtype=np.float64)).copy()
    pd[pd < minw] = minw
    pd = np.array([pd[0], pd[0]]) if pd.size == 1 else pd[:2]
    nrp, ncp = int(_mround(d1 / pd[0])), int(_mround(d2 / pd[1]))

    def edges(n, dd, rows):
        if n <= 1:
            return np.array([1, dd], dtype=np.int64)
        e = np.ceil(np.linspace(1, dd, n + 1)).astype(np.int64)
        if rows:
            e[-1] = dd
        if e[1] - e[0] < minw:
            e = np.arange(1, dd + 1, minw, dtype=np.int64)
            e[-1] = dd
        return e

    er, ec = edges(nrp, d1, True), edges(ncp, d2, False)
    nrp, ncp = er.size - 1, ec.size - 1
    patch_pos, block_pos = {}, {}
    for m in range(nrp):
        for n in range(ncp):
            patch_pos[(m, n)] = np.array([er[m], er[m + 1] - (m != nrp - 1), ec[n], ec[n + 1] - (n != ncp - 1)], dtype=np.int64)
            block_pos[(m, n)] = np.array([max(1, er[m] - w_overlap - 1), min(d1, er[m + 1] + w_overlap),
                                          max(1, ec[n] - w_overlap - 1), min(d2, ec[n + 1] + w_overlap)], dtype=np.int64)
    return (nrp, ncp), patch_pos, block_pos


def nearest_rows(n, s):
    """0-based input rows that imresize(., 1/s, 'nearest') keeps along a dimension of length n (box kernel on u = x*s + 0.5*(1 - s), taps
    mirrored at the border; the same selection the engine's cnmfe_patch_derive makes): ceil(n/s) of them"""
    x = np.arange(1, -(-n // s) + 1, dtype=np.float64)
    ind = np.floor(x * s + 0.5 * (1 - s) + 0.5).astype(np.int64)          # 1-based
    ind = np.where(ind > n, 2 * n + 1 - ind, ind)
    return ind - 1


def storage_block_index(d, patch_idx, w_overlap):
    """block_idx_r / block_idx_c of distribute_data.m:91-97: the cut lines of the blocks the reference stores the video in"""
    idx = np.concatenate([np.asarray(patch_idx) - 1 - w_overlap, np.asarray(patch_idx) + w_overlap])
    return np.unique(np.clip(idx, 1, d))


def estimate_noise_image(sn_pix, block_idx_r, block_idx_c):
    """Sources2D.m:361-376 given the per-pixel estimates: the reference evaluates storage block [r0 r1] x [c0 c1] INCLUDING the row r1 it
    shares with the next block, then drops row END-1 (not the shared one) of every block but the last, likewise for columns.  Net effect:
    row b - 1 of the image holds the estimate of row b for every interior cut line b -- reproduced as it is."""
    out = np.array(sn_pix, copy=True)
    for b in np.asarray(block_idx_r)[1:-1]:
        out[b - 2, :] = out[b - 1, :]
    for b in np.asarray(block_idx_c)[1:-1]:
        out[:, b - 2] = out[:, b - 1]
    return out


def _rect_pixels(rect, d1):
    """global column-major pixel indices of a rectangle, in the rectangle's own column-major order"""
    r0, r1, c0, c1 = [int(v) for v in rect]
    rr = np.arange(r0 - 1, r1)
    cc = np.arange(c0 - 1, c1)
    return (cc[None, :] * d1 + rr[:, None]).reshape(-1, order="F")


def determine_search_location(A, d1, d2, min_size=3.0, max_size=8.0, dist=3.0, native=True):
    """'ellipse' search masks, utilities/determine_search_location.m:50-104 + utilities/com.m:20-28.
    Vectorised over neurons; returns a boolean CSC matrix (d x K)."""
    d, K = A.shape
    mom = _footprint_moments_native(A, d1, d2) if native and K > 0 and d < 2 ** 31 else None
    if mom is not None:
        empty, cmx, cmy, vxx, vxy, vyy = mom
    else:
        A = sp.csc_matrix(A, dtype=np.float64)
        A.sort_indices()
        coo = A.tocoo()
        pix, col, val = coo.row, coo.col, coo.data
        x = (pix % d1 + 1).astype(np.float64)
        y = (pix // d1 + 1).astype(np.float64)
        s = np.bincount(col, weights=val, minlength=K)
        empty = s == 0                                                        # :52
        ss = np.where(empty, 1.0, s)
        cmx = np.bincount(col, weights=val * x, minlength=K) / ss             # com.m:24
        cmy = np.bincount(col, weights=val * y, minlength=K) / ss
        cmx = np.clip(cmx, 0, d1); cmy = np.clip(cmy, 0, d2)                  # com.m:26-28
        dx, dy = x - cmx[col], y - cmy[col]
        vxx = np.bincount(col, weights=val * dx * dx, minlength=K) / ss       # :73
        vxy = np.bincount(col, weights=val * dx * dy, minlength=K) / ss
        vyy = np.bincount(col, weights=val * dy * dy, minlength=K) / ss
    M = np.stack([np.stack([vxx, vxy], -1), np.stack([vxy, vyy], -1)], -2)
    D, V = np.linalg.eigh(M)                                              # :74 ascending eigenvalues
    d11 = np.minimum(max_size ** 2, np.maximum(min_size ** 2, D[:, 0]))   # :81
    d22 = np.minimum(max_size ** 2, np.maximum(min_size ** 2, D[:, 1]))   # :82
    # candidate window: the ellipse reaches dist*sqrt(d22) from the centre along its long axis (d22 <= max_size^2); sizing the window by the
    # largest ellipse actually present instead of the largest possible one cuts the (K, W, W) arrays below ~3x for typical footprints
    R = min(int(np.ceil(dist * max_size)), int(np.ceil(dist * np.sqrt(float(d22.max()) if K else 0.0)))) + 1
    if native and K > 0 and d < 2 ** 31:
        # the window test (:84) in the library's host helper: per image column the run of rows, its ends settled with the same expression in the same
        # order as below -- identical masks (tests/test_host_logic.py) in ~0.2 ms instead of 7-20 ms of (K, W, W) array passes, which had become the
        # host's critical path between the ring fit and the spatial update once the fit took under 9 ms
        try:
            fn = L.lib.cnmfe_search_ellipse
        except (ImportError, OSError, AttributeError):
            fn = None
        if fn is not None:
            vk = np.ascontiguousarray(np.stack([V[:, 0, 0], V[:, 1, 0], V[:, 0, 1], V[:, 1, 1]], -1), dtype=np.float64)
            cx = np.ascontiguousarray(cmx, dtype=np.float64); cy = np.ascontiguousarray(cmy, dtype=np.float64)
            a11 = np.ascontiguousarray(d11, dtype=np.float64); a22 = np.ascontiguousarray(d22, dtype=np.float64)
            em = np.ascontiguousarray(empty, dtype=np.uint8)
            optr = np.empty(K + 1, dtype=np.int64); nn = np.zeros(1, dtype=np.int64)
            cap = int(K) * (2 * R + 1) ** 2
            orow = np.empty(max(cap, 1), dtype=np.int32)
            rc = fn(int(K), int(d1), int(d2), cx.ctypes.data, cy.ctypes.data, vk.ctypes.data, a11.ctypes.data, a22.ctypes.data, em.ctypes.data, float(dist), int(R),
                    cap, optr.ctypes.data, orow.ctypes.data, nn.ctypes.data)
            if rc != 0:
                raise ValueError(L.lib.cnmfe_last_error().decode())
            n = int(nn[0])
            IND = sp.csc_matrix((np.ones(n, dtype=bool), orow[:n].copy(), optr), shape=(d, K))
            IND.has_sorted_indices = True
            return IND
    off = np.arange(-R, R + 1)
    rows = np.floor(cmx)[:, None] + off[None, :]                           # (K, W) candidate rows (1-based)
    cols = np.floor(cmy)[:, None] + off[None, :]
    ex = rows - cmx[:, None]
    ey = cols - cmy[:, None]
    # (cor*V(:,1))^2/d11 + (cor*V(:,2))^2/d22 <= dist^2                    (:84)
    p1 = ex[:, :, None] * V[:, 0, 0][:, None, None] + ey[:, None, :] * V[:, 1, 0][:, None, None]
    p2 = ex[:, :, None] * V[:, 0, 1][:, None, None] + ey[:, None, :] * V[:, 1, 1][:, None, None]
    inside = np.sqrt(p1 ** 2 / d11[:, None, None] + p2 ** 2 / d22[:, None, None]) <= dist
    inside &= (rows >= 1)[:, :, None] & (rows <= d1)[:, :, None] & (cols >= 1)[:, None, :] & (cols <= d2)[:, None, :]
    inside &= ~empty[:, None, None]                                        # :102-104
    kk, ri, ci = np.nonzero(inside)
    gp = ((cols[kk, ci] - 1) * d1 + (rows[kk, ri] - 1)).astype(np.int64)
    IND = sp.csc_matrix((np.ones(kk.size, dtype=bool), (gp, kk)), shape=(d, K))
    IND.sort_indices()
    return IND


def _footprint_moments_native(A, d1, d2):
    """(empty, cmx, cmy, vxx, vxy, vyy) of determine_search_location by the library's host helper cnmfe_footprint_moments (the same sums in the same order as the
    bincount formulation, in one pass over A); None when the library is not built or A is not a sorted float32 / int32 CSC matrix"""
    if not sp.isspmatrix_csc(A) or A.data.dtype != np.float32 or A.indices.dtype != np.int32 or not A.has_sorted_indices:
        return None
    try:
        fn = L.lib.cnmfe_footprint_moments
    except (ImportError, OSError, AttributeError):
        return None
    K = A.shape[1]
    indptr = A.indptr if A.indptr.dtype == np.int64 else A.indptr.astype(np.int64)
    out = np.empty((6, K), dtype=np.float64); em = np.empty(K, dtype=np.uint8)
    rc = fn(int(K), int(d1), int(d2), indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data, out[0].ctypes.data, em.ctypes.data,
            out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data, out[4].ctypes.data, out[5].ctypes.data)
    if rc != 0:
        raise ValueError(L.lib.cnmfe_last_error().decode())
    return em.astype(bool), out[1], out[2], out[3], out[4], out[5]


def rows_of(A, lut, nloc, cols=None, span=None, cand=None):
    """A(rows, :) for the rows a lookup table selects (global pixel -> local row or -1; local order ascending with the global one), as
    (ind, A(rows, ind)) in CSC with ind = the columns whose selected entries sum to > 0 (`sum(A(mask,:),1) > 0`, the reference's neuron
    selection) -- or, with `cols` (ascending) given, A(rows, cols).  No CSR conversion and no per-row index array: columns that cannot reach
    the rows (first / last stored row outside `span` = (lowest, highest) selected global row) are dropped in O(K), the rest costs O(their nnz)."""
    A = A if sp.isspmatrix_csc(A) else sp.csc_matrix(A)
    if not A.has_sorted_indices:
        A = A.copy(); A.sort_indices()
    K = A.shape[1]
    if cols is None and cand is not None:
        cand = np.asarray(cand, dtype=np.int64)                      # the caller's prefilter (a superset of the columns that can reach the rows)
    elif cols is None:
        cnt_all = np.diff(A.indptr)
        cand = np.nonzero(cnt_all > 0)[0]
        if span is not None and cand.size:
            first = A.indices[A.indptr[cand]]; last = A.indices[A.indptr[cand + 1] - 1]
            cand = cand[(last >= span[0]) & (first <= span[1])]
    else:
        cand = np.asarray(cols, dtype=np.int64)
        if cand.size and np.any(np.diff(cand) < 0):
            raise ValueError("cols must be ascending")
    M = _select_rows_native(A, lut, nloc, cand, cols is not None)
    if M is not None:
        return M
    return _select_rows_numpy(A, lut, nloc, cand, cols is not None)


def _csc_fast(data, indices, indptr, shape):
    """a scipy CSC matrix around arrays that ARE a valid CSC structure with ascending row indices inside every column (what the library's host helpers return):
    the constructor's format checks and the later has_sorted_indices scans cost 50-100 us per matrix, and a rank of a sharded run builds ~10 small ones per patch
    and iteration (scripts/host_python_profile.py: 3.1 ms of Python per iteration for two 128 x 128 patches, a third of it here)"""
    M = sp.csc_matrix.__new__(sp.csc_matrix)
    M._shape = (int(shape[0]), int(shape[1])); M.maxprint = 50
    if indptr.dtype != indices.dtype:                          # (scipy's kernels want one index type: the column pointers follow the row indices, as its constructor does)
        indptr = indptr.astype(indices.dtype)
    M.data, M.indices, M.indptr = data, indices, indptr
    M.has_sorted_indices = True
    return M


def _csc_from_triplets(r, c, v, shape):
    """the CSC matrix of DISJOINT (row, column, value) triplets (ValueError on a pair given twice): one counting pass in the library's host helper
    (cnmfe_csc_from_triplets) -- scipy's COO -> CSC conversion sorts the whole list, and every rank of a sharded run assembles the WHOLE gathered A; falls back
    to scipy without the library"""
    r = np.ascontiguousarray(r, dtype=np.int32); c = np.ascontiguousarray(c, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.float32)
    try:
        fn = L.lib.cnmfe_csc_from_triplets
    except (ImportError, OSError, AttributeError):                # (no library, or a CNMFE_LIB build without the helper)
        return sp.csc_matrix((v, (r, c)), shape=shape)
    n = int(r.size)
    optr = np.empty(shape[1] + 1, dtype=np.int64); orow = np.empty(max(n, 1), dtype=np.int32); oval = np.empty(max(n, 1), dtype=np.float32)
    rc = fn(n, r.ctypes.data, c.ctypes.data, v.ctypes.data, int(shape[1]), int(shape[0]), optr.ctypes.data, orow.ctypes.data, oval.ctypes.data)
    if rc != 0:
        # PRECONDITION: disjoint triplets (the patches' rows are, update_spatial_parallel.m:324-334).  A pair given twice -- which scipy's conversion would sum
        # silently -- or an index out of range is an error of the caller, reported as such
        raise ValueError(L.lib.cnmfe_last_error().decode())
    return _csc_fast(oval[:n], orow[:n], optr, shape)


def _select_rows_native(A, lut, nloc, cand, keep_all):
    """rows_of's selection by the library's host helper cnmfe_csc_select_rows (one pass over the candidates' entries in C: ~10x the NumPy
    formulation below at the 100-neuron patches of a 4 x 4 decomposition, where six such slices per patch and iteration were most of the
    host's time).  None when the library is not built (host-logic tests on a checkout without it) or the matrix is not int32 / float32."""
    if A.indices.dtype != np.int32 or A.data.dtype != np.float32 or lut.dtype != np.int32:
        return None
    try:
        fn = L.lib.cnmfe_csc_select_rows
    except (ImportError, OSError):
        return None
    indptr = A.indptr if A.indptr.dtype == np.int64 else A.indptr.astype(np.int64)
    cand = np.ascontiguousarray(cand, dtype=np.int64)
    nc = int(cand.size)
    cap = int((indptr[cand + 1] - indptr[cand]).sum()) if nc else 0
    ind = np.empty(nc, dtype=np.int64); optr = np.empty(nc + 1, dtype=np.int64)
    orow = np.empty(max(cap, 1), dtype=np.int32); oval = np.empty(max(cap, 1), dtype=np.float32)
    nk = ctypes.c_int64(0)
    rc = fn(indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data, lut.ctypes.data, nc, cand.ctypes.data, 1 if keep_all else 0, cap,
            ind.ctypes.data, optr.ctypes.data, orow.ctypes.data, oval.ctypes.data, ctypes.byref(nk))
    if rc != 0:
        raise ValueError(L.lib.cnmfe_last_error().decode())
    nk = nk.value
    n = int(optr[nk])
    return ind[:nk], _csc_fast(oval[:n], orow[:n], optr[:nk + 1], (nloc, nk))


def _select_block_patch_native(A, lut_b, nb, lut_p, npx, cand):
    """(ind, A(block, ind), A(patch, ind)) in ONE pass of the library's host helper cnmfe_csc_select_block_patch -- what two rows_of calls give (the temporal update asks
    for both per patch, update_temporal_parallel.m:83-91, on the spatial -> temporal hand-over where the device waits for the host); None when the helper cannot run"""
    if A.indices.dtype != np.int32 or A.data.dtype != np.float32 or lut_b.dtype != np.int32 or lut_p.dtype != np.int32:
        return None
    try:
        fn = L.lib.cnmfe_csc_select_block_patch
    except (ImportError, OSError, AttributeError):
        return None
    indptr = A.indptr if A.indptr.dtype == np.int64 else A.indptr.astype(np.int64)
    cand = np.ascontiguousarray(cand, dtype=np.int64)
    nc = int(cand.size)
    cap = int((indptr[cand + 1] - indptr[cand]).sum()) if nc else 0
    ind = np.empty(nc, dtype=np.int64); bptr = np.empty(nc + 1, dtype=np.int64); pptr = np.empty(nc + 1, dtype=np.int64)
    brow = np.empty(max(cap, 1), dtype=np.int32); bval = np.empty(max(cap, 1), dtype=np.float32)
    prow = np.empty(max(cap, 1), dtype=np.int32); pval = np.empty(max(cap, 1), dtype=np.float32)
    nk = ctypes.c_int64(0)
    rc = fn(indptr.ctypes.data, A.indices.ctypes.data, A.data.ctypes.data, lut_b.ctypes.data, lut_p.ctypes.data, nc, cand.ctypes.data, cap,
            ind.ctypes.data, bptr.ctypes.data, brow.ctypes.data, bval.ctypes.data, pptr.ctypes.data, prow.ctypes.data, pval.ctypes.data, ctypes.byref(nk))
    if rc != 0:
        raise ValueError(L.lib.cnmfe_last_error().decode())
    nk = nk.value
    n_b, n_p = int(bptr[nk]), int(pptr[nk])
    return ind[:nk], _csc_fast(bval[:n_b], brow[:n_b], bptr[:nk + 1], (nb, nk)), _csc_fast(pval[:n_p], prow[:n_p], pptr[:nk + 1], (npx, nk))


def _bbox_native(A, d1):
    """(non-empty columns, first / last image row, first / last image column) of a sorted int32 CSC footprint matrix by cnmfe_csc_bbox; None when the helper cannot run"""
    if A.indices.dtype != np.int32 or not A.has_sorted_indices:
        return None
    try:
        fn = L.lib.cnmfe_csc_bbox
    except (ImportError, OSError, AttributeError):
        return None
    K = A.shape[1]
    indptr = A.indptr if A.indptr.dtype == np.int64 else A.indptr.astype(np.int64)
    nz = np.empty(max(K, 1), dtype=np.int64); box = np.empty((4, max(K, 1)), dtype=np.int32)
    n = ctypes.c_int64(0)
    rc = fn(int(K), int(d1), indptr.ctypes.data, A.indices.ctypes.data, nz.ctypes.data, box[0].ctypes.data, box[1].ctypes.data, box[2].ctypes.data, box[3].ctypes.data, ctypes.byref(n))
    if rc != 0:
        raise ValueError(L.lib.cnmfe_last_error().decode())
    n = n.value
    return nz[:n], box[0, :n], box[1, :n], box[2, :n], box[3, :n]


def _select_rows_numpy(A, lut, nloc, cand, keep_all):
    """the same selection with index arithmetic (a scipy column slice costs more than the data it moves here)"""
    K = A.shape[1]
    nS = int(cand.size)
    if nS != K:
        starts = A.indptr[cand].astype(np.int64); lens = (A.indptr[cand + 1] - A.indptr[cand]).astype(np.int64)
        tot = int(lens.sum())
        ends = np.cumsum(lens)
        src = np.arange(tot, dtype=np.int64) + np.repeat(starts - (ends - lens), lens)
        S_indices = A.indices[src]; S_data = A.data[src]
    else:
        lens = np.diff(A.indptr).astype(np.int64)
        S_indices = A.indices; S_data = A.data
    loc = lut[S_indices]
    keep = loc >= 0
    cid = np.repeat(np.arange(nS, dtype=np.int64), lens)
    if not keep_all:
        csum = np.bincount(cid[keep], weights=S_data[keep], minlength=nS)
        sub = np.nonzero(csum > 0)[0]
        if sub.size != nS:
            colsel = np.zeros(nS, dtype=bool); colsel[sub] = True
            keep &= colsel[cid]
        ind = cand[sub]
    else:
        sub = np.arange(nS); ind = cand
    cnt = np.bincount(cid[keep], minlength=nS)[sub]
    indptr = np.zeros(sub.size + 1, dtype=np.int64); np.cumsum(cnt, out=indptr[1:])
    M = sp.csc_matrix((S_data[keep], loc[keep].astype(np.int32), indptr), shape=(nloc, sub.size))
    return ind, M

# --------------------------------------------------------------------------------------
# the blocked, GPU-resident video ( == mat_data + get_patch_data of the reference )
# --------------------------------------------------------------------------------------
class PatchedVideo:
    """Geometry of distribute_data + one resident block per owned patch.

    rank/world_size shard the patches round-robin in MATLAB's linear patch order
    (SURVEY.md section 8(e)); each rank uploads only the blocks of the patches it owns.
    """

    def __init__(self, d1, d2, T, patch_dims, ring_radius, engine: Engine, rank=0, world_size=1):
        self.d1, self.d2, self.T = int(d1), int(d2), int(T)
        self.dims = (self.d1, self.d2, self.T)
        self.w_overlap = int(ring_radius)                                  # Sources2D.m:236
        (self.nr_patch, self.nc_patch), self.patch_pos, self.block_pos = distribute_geometry(d1, d2, patch_dims, ring_radius)
        # MATLAB linear index over the nr_patch x nc_patch cell: row index fastest
        self.order = [(m, n) for n in range(self.nc_patch) for m in range(self.nr_patch)]
        self.engine = engine
        self.rank, self.world_size = rank, world_size
        self.owned = [idx for i, idx in enumerate(self.order) if i % world_size == rank]
        self.pid = {idx: i for i, idx in enumerate(self.order)}
        self.patch_pix = {idx: _rect_pixels(self.patch_pos[idx], d1) for idx in self.order}
        self.block_pix = {idx: _rect_pixels(self.block_pos[idx], d1) for idx in self.order}
        self.ind_patch = {}
        self._halo = {}
        self._lut = {}
        for idx in self.order:
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mask = np.zeros((b[1] - b[0] + 1, b[3] - b[2] + 1), dtype=bool)
            mask[p[0] - b[0]:p[1] - b[0] + 1, p[2] - b[2]:p[3] - b[2] + 1] = True   # update_spatial_parallel.m:140-141
            self.ind_patch[idx] = np.nonzero(mask.reshape(-1, order="F"))[0]
        for idx in self.owned:
            engine.create_patch(self.pid[idx], self.patch_pos[idx], self.block_pos[idx], d1, d2, T)

    def lut(self, idx, kind):
        """(table, rows, (lowest, highest pixel)): global pixel -> row index inside the block / patch / halo of patch idx (-1 elsewhere), cached: lets `rows_of` slice a CSC matrix in
        O(nnz) without a CSR conversion or an index array of block length (what grows with the FOV on every rank of a sharded run)"""
        key = (idx, kind)
        t = self._lut.get(key)
        if t is None:
            pix = self.block_pix[idx] if kind == "block" else self.patch_pix[idx] if kind == "patch" else self.halo_pix(idx)
            t = np.full(self.d1 * self.d2, -1, dtype=np.int32)
            t[pix] = np.arange(pix.size, dtype=np.int32)
            t = self._lut[key] = (t, pix.size, (int(pix.min()), int(pix.max())) if pix.size else (0, -1))
        return t

    def halo_pix(self, idx):
        """block pixels outside the patch (mask==1 in update_spatial_parallel.m:84-85), cached"""
        h = self._halo.get(idx)
        if h is None:
            h = self._halo[idx] = np.setdiff1d(self.block_pix[idx], self.patch_pix[idx], assume_unique=True)
        return h

    def upload_from_full(self, Y_td, chunk=512):
        """Y_td: (T, d1*d2) host video, frame-major / pixels column-major (MATLAB d x T)."""
        for idx in self.owned:
            bp = self.block_pix[idx]
            for t0 in range(0, self.T, chunk):
                self.engine.upload_block(self.pid[idx], Y_td[t0:t0 + chunk][:, bp], t0)

    def upload_block_device(self, idx, dev_ptr):
        self.engine.upload_block_device(self.pid[idx], dev_ptr, self.T)

    # -- data plane: what distribute_data.m:56-173 + get_patch_data.m:50-93 do for the reference (blocks with their halo out of the recording).
    # The recording is read ONCE, in frame chunks; every owned block of a chunk goes up in the file's own element type (uint8/uint16/float16/
    # float32/float64: cnmfe_upload_block converts on the device) -- there is no blocked intermediate file.
    def upload_from_images(self, frames, chunk=256):
        """frames: an array (T, d1, d2) or any iterable of T images of shape (d1, d2) -- rows x columns as MATLAB's Y(:,:,t), e.g. TIFF pages.
        Pixels are re-ordered to the column-major order of the reference (pixel = (c-1)*d1 + r) chunk by chunk."""
        t0, buf = 0, []
        def flush():
            nonlocal t0, buf
            if len(buf) == 0:
                return
            arr = np.stack(buf) if not isinstance(buf, np.ndarray) else buf
            if arr.shape[1:] != (self.d1, self.d2):
                raise ValueError("frames are %s, the field of view is (%d, %d)" % (arr.shape[1:], self.d1, self.d2))
            cm = arr.transpose(0, 2, 1).reshape(arr.shape[0], self.d1 * self.d2)        # column-major pixel order
            for idx in self.owned:
                self.engine.upload_block(self.pid[idx], cm[:, self.block_pix[idx]], t0)
            t0 += arr.shape[0]; buf = []
        if isinstance(frames, np.ndarray) and frames.ndim == 3:
            for s0 in range(0, frames.shape[0], chunk):
                buf = frames[s0:s0 + chunk]; flush()
        else:
            for im in frames:
                buf.append(np.asarray(im))
                if len(buf) == chunk:
                    flush()
            flush()
        if t0 != self.T:
            raise ValueError("the recording has %d frames, the patches were created for %d" % (t0, self.T))

    def upload_from_tiff(self, path, chunk=256):
        """a multi-page TIFF (what the reference's demos ship and tif2mat.m / bigread2.m read), one page per frame"""
        from PIL import Image, ImageSequence
        with Image.open(path) as im:
            self.upload_from_images((np.asarray(page) for page in ImageSequence.Iterator(im)), chunk)

    def upload_from_raw(self, path, dtype, offset=0, order="F", chunk=256):
        """a flat binary recording of T frames (np.memmap, nothing is read twice): order='F' = MATLAB's fwrite of a d1 x d2 x T array (pixels
        column-major inside a frame), order='C' = row-major frames (rows x columns, e.g. numpy's tofile of a (T, d1, d2) array)."""
        mm = np.memmap(path, dtype=np.dtype(dtype), mode="r", offset=offset, shape=(self.T, self.d1 * self.d2))
        if order == "F":
            self.upload_from_full(mm, chunk)
        else:
            self.upload_from_images(mm.reshape(self.T, self.d1, self.d2), chunk)

    def upload_from_hdf5(self, path, dataset=None, chunk=256, frame0=0):
        """a `.h5` / `.hdf5` recording with its movie in the root group (smod_bigread2.m:338-355) or a v7.3 `.mat` recording holding `Y` or a single
        array (smod_bigread2.m:378-400).  A d1 x d2 x T MATLAB array / h5read view is the HDF5 dataset of dims (T, d2, d1) -- singleton dims anywhere, as
        in the 5-D files get_data_dimension.m:32-35 indexes with [2 3 5] -- so a hyperslab over the first axis is a run of frames already in the reference's
        pixel order.  Read once, in frame chunks; frame0 = frames to skip (obj.frame_range(1) - 1)."""
        from . import h5io
        with h5io.H5File(path) as f:
            name = dataset if dataset is not None else _pick_movie(f)
            dims = f.shape(name)
            axes = [i for i, n in enumerate(dims) if n != 1]
            if len(axes) != 3 or (dims[axes[1]], dims[axes[2]]) != (self.d2, self.d1):
                raise ValueError("dataset %r of %s has MATLAB size %s, the field of view is %d x %d" % (name, path, tuple(reversed(dims)), self.d1, self.d2))
            if frame0 < 0 or frame0 + self.T > dims[axes[0]]:
                raise ValueError("dataset %r has %d frames, frames %d..%d were asked for" % (name, dims[axes[0]], frame0 + 1, frame0 + self.T))
            f.dtype(name)                                                  # (raises for a non-numeric dataset before anything is uploaded)
            for t0 in range(0, self.T, chunk):
                n = min(chunk, self.T - t0)
                start = [0] * len(dims); count = list(dims)
                start[axes[0]], count[axes[0]] = frame0 + t0, n
                cm = f.read(name, start, count).reshape(n, self.d1 * self.d2)
                for idx in self.owned:
                    self.engine.upload_block(self.pid[idx], cm[:, self.block_pix[idx]], t0)

    def upload_from_mat_data(self, path, chunk=256, frame0=0):
        """the blocked file distribute_data.m:127-173 wrote (`mat_data`): every owned block (patch + halo) is put together from the storage blocks
        `Y_r0_r1_c0_c1` that meet it, as get_patch_data.m:50-93 does -- neighbouring storage blocks share their cut line, the shared line is the same data
        in both -- and goes up in the file's element type.  Only the parts of the file this rank's blocks cover are read."""
        from . import h5io
        with h5io.H5File(path) as f:
            info = mat_data_info(f)
            if tuple(info["dims"][:2]) != (self.d1, self.d2):
                raise ValueError("%s holds a %d x %d field of view, the patches were created for %d x %d" % ((path,) + tuple(info["dims"][:2]) + (self.d1, self.d2)))
            if frame0 < 0 or frame0 + self.T > info["dims"][2]:
                raise ValueError("%s holds %d frames, frames %d..%d were asked for" % (path, info["dims"][2], frame0 + 1, frame0 + self.T))
            stored = info["blocks"]
            for idx in self.owned:
                r0, r1, c0, c1 = [int(v) for v in self.block_pos[idx]]
                nr, nc = r1 - r0 + 1, c1 - c0 + 1
                parts = []
                for (s0, s1, q0, q1), name in stored.items():
                    a0, a1, b0, b1 = max(r0, s0), min(r1, s1), max(c0, q0), min(c1, q1)
                    if a0 <= a1 and b0 <= b1:
                        parts.append((name, (b0 - q0, a0 - s0), (b1 - b0 + 1, a1 - a0 + 1), (b0 - c0, a0 - r0)))
                cover = np.zeros((nc, nr), dtype=bool)
                for _, _, (wc, wr), (oc, orr) in parts:
                    cover[oc:oc + wc, orr:orr + wr] = True
                if not cover.all():
                    raise ValueError("%s: the stored blocks do not cover block [%d %d %d %d]" % (path, r0, r1, c0, c1))
                for t0 in range(0, self.T, chunk):
                    n = min(chunk, self.T - t0)
                    buf = np.empty((n, nc, nr), dtype=info["dtype"])
                    for name, (sc, sr), (wc, wr), (oc, orr) in parts:
                        buf[:, oc:oc + wc, orr:orr + wr] = f.read(name, (frame0 + t0, sc, sr), (n, wc, wr))
                    self.engine.upload_block(self.pid[idx], buf.reshape(n, nc * nr), t0)


def _pick_movie(f):
    """the movie of a recording: `Y` when there is one (a .mat with Y and Ysiz), else the only numeric dataset with three non-singleton dims"""
    if f.has("Y"):
        return "Y"
    cand = []
    for n in f.names():
        if f.has(n):
            try:
                f.dtype(n)
            except TypeError:
                continue
            if sum(1 for v in f.shape(n) if v != 1) == 3:
                cand.append(n)
    if len(cand) != 1:
        raise ValueError("%s: cannot tell which dataset is the movie (%s); name it" % (f.path, ", ".join(cand) or "none has three dimensions"))
    return cand[0]


def mat_data_info(f):
    """what a blocked `mat_data` file says about itself (distribute_data.m:129-158): dims, patch_dims, w_overlap, the cut lines, the element type and
    the name of every stored block keyed by its rectangle.  `f`: a path or an open h5io.H5File."""
    from . import h5io
    import re
    if not isinstance(f, h5io.H5File):
        with h5io.H5File(f) as g:
            return mat_data_info(g)
    ints = lambda name: [int(v) for v in np.asarray(f.matlab_value(name)).reshape(-1, order="F")]
    blocks = {}
    for n in f.names():
        m = re.fullmatch(r"Y_(\d+)_(\d+)_(\d+)_(\d+)", n)
        if m:
            rect = tuple(int(v) for v in m.groups())
            if f.matlab_size(n)[:2] != (rect[1] - rect[0] + 1, rect[3] - rect[2] + 1):
                raise ValueError("%s: block %s has size %s" % (f.path, n, f.matlab_size(n)))
            blocks[rect] = n
    if not blocks:
        raise ValueError("%s holds no Y_r0_r1_c0_c1 block: not a file distribute_data wrote" % f.path)
    dts = {f.dtype(n) for n in blocks.values()}
    if len(dts) != 1:
        raise ValueError("%s: the stored blocks have different element types" % f.path)
    return {"dims": tuple(ints("dims")), "patch_dims": tuple(ints("patch_dims")), "w_overlap": ints("w_overlap")[0],
            "block_idx_r": np.array(ints("block_idx_r")), "block_idx_c": np.array(ints("block_idx_c")), "dtype": dts.pop(), "blocks": blocks}


# --------------------------------------------------------------------------------------
# Sources2D
# --------------------------------------------------------------------------------------
class _LazyRow:
    """a length-K vector that arrives with a LazyHostTraces of shape (1, K) (kernel_pars, neuron_sn of deconvTemporal on the bound matrix)"""
    def __init__(self, lazy):
        self._lazy = lazy
    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._lazy)[0]
        return a.copy() if (copy or dtype is None) else a.astype(dtype)
    def __getitem__(self, k):
        return np.asarray(self._lazy)[0][k]
    def __len__(self):
        return self._lazy.shape[1]
    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(np.asarray(self._lazy)[0], name)

for _op in ("add", "sub", "mul", "truediv", "radd", "rsub", "rmul", "rtruediv", "lt", "le", "gt", "ge", "eq", "ne", "and", "rand"):
    setattr(_LazyRow, "__%s__" % _op, (lambda op: lambda self, *a: getattr(np.asarray(self._lazy)[0], "__%s__" % op)(*a))(_op))
_LazyRow.__hash__ = object.__hash__

_UNSET = object()


class Sources2D:
    """State and the three update methods of the reference's handle class (Sources2D.m:10-57):
    A (d x K sparse), A_prev, C, C_prev, C_raw (K x T), W{.}/b0{.} (resident on the GPU per patch,
    fetch with ``get_W`` / ``get_b0``), b0_new, options, P (sn, Ymean)."""

    def __init__(self, video: PatchedVideo, options: Options, A, C, sn, dist_group=None):
        if options.background_model != "ring":
            raise ValueError("only the ring background model is built (north star); got %r" % options.background_model)
        if not (1 <= int(options.bg_ssub) <= 8):
            raise ValueError("bg_ssub must be in 1..8")
        self.video = video
        self.options = options
        options.d1, options.d2 = video.d1, video.d2
        self.engine = video.engine
        d = video.d1 * video.d2
        self.A = sp.csc_matrix(A, dtype=np.float32)
        if self.A.shape[0] != d:
            raise ValueError("A must have d1*d2 rows")
        self.C = np.ascontiguousarray(C, dtype=np.float32)
        self.C_raw = self.C.copy()
        self.A_prev = self.A          # snapshots: A and C are only ever REPLACED by the update methods, never mutated in place
        self.C_prev = self.C
        self.P = {"sn": np.asarray(sn, dtype=np.float32).reshape(-1).copy(), "Ymean": {}}
        self._b0_new_val = None; self._b0_new_src = None
        self.dist = dist_group
        # test hook: take the collective (sharded) branches even with one rank, so that a single-GPU box can run the all-gather of A, the
        # device-side stitch of C_raw and the lazy all-reduces over RCCL exactly as an N-rank job does (scripts/nccl_smoke.py)
        self.force_collectives = False
        self._bind_C()
        self.ssub = int(options.bg_ssub)
        npatch = len(video.order)
        # bg_ssub > 1: every patch gets two low-resolution companions on the device (cnmfe.h, cnmfe_patch_derive)
        self.pid_fit = {idx: npatch + 2 * video.pid[idx] for idx in video.order}
        self.pid_res = {idx: npatch + 2 * video.pid[idx] + 1 for idx in video.order}
        for idx in video.owned:                                           # initComponents_parallel.m:213-236
            self.engine.ring_init(video.pid[idx], options.ring_radius, options.num_neighbors)
            if self.ssub > 1:                                             # :214,237-251: rr = ceil(r/bg_ssub), W on the low-resolution block
                rr = -(-int(options.ring_radius) // self.ssub)
                self.engine.patch_derive(video.pid[idx], self.pid_fit[idx], self.ssub, "nearest")
                self.engine.patch_derive(video.pid[idx], self.pid_res[idx], self.ssub, "bicubic")
                self.engine.ring_init(self.pid_fit[idx], rr, options.num_neighbors)
                self.engine.ring_init(self.pid_res[idx], rr, options.num_neighbors)
            if hasattr(self.engine, "fit_reserve"):                       # the fit's large buffers now, not inside the first update_background_parallel
                self.engine.fit_reserve(video.pid[idx] if self.ssub == 1 else self.pid_fit[idx])
            self.P["Ymean"][idx] = self.engine.ymean(video.pid[idx])[video.ind_patch[idx]]      # :338-339, patch part
        self._ymean_full = None

    # -- accessors ------------------------------------------------------------------
    def get_W(self, idx):
        return self.engine.ring_csr(self.video.pid[idx] if self.ssub == 1 else self.pid_fit[idx])

    def _slice(self, A, idx, kind, cols=None):
        """(ind, A(pixels of idx's block / patch / halo, ind)): the reference's `mask` selections (update_*_parallel.m) without a CSR of A"""
        t, n, span = self.video.lut(idx, kind)
        cand = None
        if cols is None and sp.isspmatrix_csc(A) and A.has_sorted_indices:
            # columns whose bounding box meets the rectangle (block for 'block' / 'halo', patch for 'patch'): with 4 x 4 patches the first / last
            # stored row test of rows_of alone passes every neuron of the patch's COLUMN band; the boxes are found once per matrix
            bb = self._bbox_of(A)
            if bb is not None:
                v = self.video
                r0, r1, c0, c1 = [int(x) - 1 for x in (v.patch_pos[idx] if kind == "patch" else v.block_pos[idx])]
                nz, rmin, rmax, cmin, cmax = bb
                cand = nz[(rmax >= r0) & (rmin <= r1) & (cmax >= c0) & (cmin <= c1)]
        return rows_of(A, t, n, cols=cols, span=span, cand=cand)

    def _slice_block_patch(self, A, idx):
        """(ind, A(block, ind), A(patch, ind)) = _slice(A, idx, "block") followed by _slice(A, idx, "patch", cols=ind), in one pass over the candidates"""
        if sp.isspmatrix_csc(A) and A.has_sorted_indices:
            bb = self._bbox_of(A)
            if bb is not None:
                v = self.video
                tb, nb, _ = v.lut(idx, "block"); tp, npx, _ = v.lut(idx, "patch")
                r0, r1, c0, c1 = [int(x) - 1 for x in v.block_pos[idx]]
                nz, rmin, rmax, cmin, cmax = bb
                got = _select_block_patch_native(A, tb, nb, tp, npx, nz[(rmax >= r0) & (rmin <= r1) & (cmax >= c0) & (cmin <= c1)])
                if got is not None:
                    return got
        ind, A_blk = self._slice(A, idx, "block")
        return ind, A_blk, (self._slice(A, idx, "patch", cols=ind)[1] if ind.size else None)

    def _bbox_of(self, A):
        """(non-empty columns, their first / last image row and column), cached for the last few matrices by identity"""
        cache = self.__dict__.setdefault("_bbox_cache", [])
        for M, bb in cache:
            if M is A:
                return bb
        bb = _bbox_native(A, self.video.d1)
        nz = np.nonzero(np.diff(A.indptr) > 0)[0] if bb is None else None
        if bb is not None:
            pass
        elif nz.size == 0:
            bb = (nz, nz, nz, nz, nz)
        else:
            d1 = self.video.d1
            rr = A.indices % d1
            starts = A.indptr[nz]
            # rows of a column are stored ascending in pixel index = image column major: first / last entry give the column range
            cmin = A.indices[starts] // d1; cmax = A.indices[A.indptr[nz + 1] - 1] // d1
            rmin = np.minimum.reduceat(rr, starts); rmax = np.maximum.reduceat(rr, starts)
            bb = (nz, rmin, rmax, cmin, cmax)
        cache.append((A, bb))
        del cache[:-4]
        return bb

    @staticmethod
    def _rows(Cm, ind):
        """Cm(ind, :) -- the matrix itself when ind is every row (so that a bound trace matrix is recognised by identity), else a
        (matrix, rows) pair that the engine resolves on the device when the matrix is the bound one"""
        return Cm if ind.size == Cm.shape[0] else BoundRows(Cm, ind)

    def _bind_C(self):
        """one upload of obj.C per iteration instead of one per engine call (cnmfe_traces_bind)"""
        if hasattr(self.engine, "bind_traces"):
            self.engine.bind_traces(self.C)

    def _residual(self, idx, A_prev_b, C_prev_b, tag=None):
        """the background-subtraction expression of update_spatial_parallel.m:162-178 / update_temporal_parallel.m:149-165.
        tag: who asked (see _temporal_residual_early); any other request replaces it"""
        self.__dict__.setdefault("_resid_tag", {})[idx] = tag
        if self.ssub == 1:
            self.engine.residual(self.video.pid[idx], A_prev_b, C_prev_b)
        else:
            self.engine.residual_ssub(self.video.pid[idx], self.pid_res[idx], self.ssub, A_prev_b, C_prev_b)

    def init_residual(self, idx):
        """@Sources2D/initComponents_residual_parallel.m:106-121,186-217 (ring model): the video greedyROI_endoscope searches for missed
        neurons in patch `idx` -- the block's neurons subtracted, then the ring background,
            Ypatch = Y(ind_patch,:) - A*C - W*(Y - A*C) - (b0 - W*mean(Y - A*C, 2))       (:199,:206; the imresize form :209-217 for bg_ssub > 1).
        The sweep over the video is cnmfe_residual[_ssub] with the block's current neurons (it leaves that residual resident, like
        the call of a temporal update would); the footprints' own A(patch,:)*C is a sparse product on the exported copy.  Returns a
        (T, d_patch) float32 array (frame-major == the reference's d x T in MATLAB order).  greedyROI_endoscope itself is host code
        outside this engine (initialisation is out of scope)."""
        self._need_data()
        v = self.video
        ind, A_blk = self._slice(self.A, idx, "block")                                               # :116-117
        C_blk = self._rows(self.C, ind) if ind.size else None                                        # :120
        pid = v.pid[idx]
        if self.ssub == 1:
            out = self.engine.residual(pid, A_blk if ind.size else None, C_blk, want=True)
        else:
            out = self.engine.residual_ssub(pid, self.pid_res[idx], self.ssub, A_blk if ind.size else None, C_blk, want=True)
        if ind.size:
            A_pp = self._slice(self.A, idx, "patch", cols=ind)[1].tocsr()
            Cb = np.asarray(C_blk, dtype=np.float32)
            for t0 in range(0, out.shape[0], 2048):                                                  # :199, on the patch rows
                out[t0:t0 + 2048] -= (A_pp @ Cb[:, t0:t0 + 2048]).T
        return out

    def get_b0(self, idx):
        return self.engine.b0(self.video.pid[idx])

    # obj.A_raw: the spatial update before post-processing.  Nothing in the iteration reads it, so the one-patch path stores the recipe (a callable) and
    # the matrix is assembled on first read
    @property
    def A_raw(self):
        v = self.__dict__.get("_A_raw")
        if callable(v):
            v = self.__dict__["_A_raw"] = v()
        return v

    @A_raw.setter
    def A_raw(self, val):
        self.__dict__["_A_raw"] = val

    # how many patches' spatial results may be outstanding before the oldest is collected: the host queues that many patches ahead of the device (each
    # result waits in a pinned buffer of its size; a lag of 1 tied the host to the device's pace and left the device idle whenever a patch's host work ran long)
    spatial_lag = 16

    def _need_data(self):
        if self.video is None:
            raise RuntimeError("No data file selected")                   # update_spatial_parallel.m:13-38

    def estimate_noise(self, frame_range=None):
        """obj.P.sn = obj.estimate_noise(frame_range, 'psd')  (Sources2D.m:328-379): GetSn per pixel on the device (every owned patch evaluates its
        block, the patch part is kept), then the storage-block bookkeeping of :361-376.  frame_range = (1, n): the first n frames."""
        self._need_data()
        v = self.video
        n = None if frame_range is None else int(frame_range[1]) - int(frame_range[0]) + 1
        if frame_range is not None and int(frame_range[0]) != 1:
            raise NotImplementedError("estimate_noise reads the frames from the first one on (the reference's default is [1, min(T, 3000)])")
        out = np.zeros(v.d1 * v.d2, dtype=np.float64)
        for idx in v.owned:
            out[v.patch_pix[idx]] = self.engine.estimate_noise(v.pid[idx], n)[v.ind_patch[idx]]
        out = self._allreduce(out).reshape(v.d1, v.d2, order="F")
        pr = np.array([int(v.patch_pos[(m, 0)][0]) for m in range(v.nr_patch)] + [v.d1])       # patch_idx_r / patch_idx_c of distribute_data.m:57-79
        pc = np.array([int(v.patch_pos[(0, j)][2]) for j in range(v.nc_patch)] + [v.d2])
        sn = estimate_noise_image(out, storage_block_index(v.d1, pr, v.w_overlap), storage_block_index(v.d2, pc, v.w_overlap))
        self.P["sn"] = sn.reshape(-1, order="F").astype(np.float32)
        return sn

    def reconstruct_b0(self):
        """Sources2D.m:1153-1190: stitch b0{m} into a d1 x d2 image (owned patches; all-reduced if sharded)."""
        v = self.video
        out = np.zeros(v.d1 * v.d2, dtype=np.float32)
        for idx in v.owned:
            out[v.patch_pix[idx]] = self.get_b0(idx)
        out = self._allreduce(out)
        return out.reshape(v.d1, v.d2, order="F")

    def ymean_full(self):
        if self._ymean_full is None:
            v = self.video
            out = np.zeros(v.d1 * v.d2, dtype=np.float64)
            for idx in v.owned:
                out[v.patch_pix[idx]] = self.P["Ymean"][idx]
            self._ymean_full = self._allreduce(out)
        return self._ymean_full

    def _allreduce(self, arr):
        if self.dist is None or (self.video.world_size == 1 and not self.force_collectives):
            return arr
        import torch
        import torch.distributed as td
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if td.get_backend(self.dist) == "nccl":
            t = t.cuda()
        td.all_reduce(t, group=self.dist)
        return t.cpu().numpy()

    def _cmean(self):
        """mean(obj.C, 2), cached per C object (K x T can be hundreds of MB when many ranks share one FOV)"""
        if getattr(self, "_cmean_of", None) is not self.C:
            self._cmean_val = self.C.mean(axis=1, dtype=np.float64)
            self._cmean_of = self.C
        return self._cmean_val

    def _update_b0_new(self):
        """obj.b0_new = Ymean - A*mean(C,2) (update_spatial_parallel.m:349).  Evaluated on first read from the
        (A, C) of this moment -- both are replaced, never mutated, so holding the references is a snapshot."""
        self.ymean_full()                                                 # (collective when sharded: keep it eager)
        self._b0_new_src = (self.A, self.C)

    @property
    def b0_new(self):
        if isinstance(self._b0_new_src, str):
            self._b0_new_val = self.reconstruct_b0(); self._b0_new_src = None
        if self._b0_new_src is not None:
            A, Cm = self._b0_new_src
            v = self.video
            cm = self._cmean() if Cm is self.C else Cm.mean(axis=1, dtype=np.float64)
            self._b0_new_val = (self.ymean_full() - np.asarray(A @ cm).ravel()).reshape(v.d1, v.d2, order="F")
            self._b0_new_src = None
        return self._b0_new_val

    @b0_new.setter
    def b0_new(self, val):
        self._b0_new_val, self._b0_new_src = val, None

    def set_components(self, A, C, C_raw=None):
        """Replace obj.A, obj.C (and obj.C_raw) between two update calls -- what the reference's merge / delete methods do to the handle object between
        the calls of demo_large_data_1p.m:203-209 (merge_neurons_dist_corr, merge_high_corr, remove_false_positives: not on this engine's path).  K may
        change; obj.A_prev / obj.C_prev stay what the last background update left (the reference's methods do not touch them either), so the next
        residual still subtracts the neurons the background was fitted against.  The new trace matrix is bound on the engine."""
        A = sp.csc_matrix(A, dtype=np.float32)
        C = np.ascontiguousarray(C, dtype=np.float32)
        if A.shape[0] != self.video.d1 * self.video.d2 or A.shape[1] != C.shape[0] or C.shape[1] != self.video.T:
            raise ValueError("A is %s, C is %s; expected (%d, K) and (K, %d)" % (A.shape, C.shape, self.video.d1 * self.video.d2, self.video.T))
        self.A, self.C = A, C
        self.C_raw = C if C_raw is None else np.ascontiguousarray(C_raw, dtype=np.float32)
        self._bind_C()
        self._update_b0_new()

    def delete(self, ind):
        """obj.delete(ind)  (@Sources2D/Sources2D.m:762-811): columns of A, rows of C / C_raw / S and of P.kernel_pars go; A_prev, C_prev stay"""
        K = self.A.shape[1]
        ind = np.atleast_1d(np.asarray(ind))
        if ind.dtype == np.bool_:                                                 # obj.delete(mask): MATLAB's usual calling style (obj.A(:, ind) = [])
            if ind.size != K:
                raise ValueError("delete: a logical index of %d entries for %d neurons" % (ind.size, K))
            ind = np.flatnonzero(ind)
        ind = ind.astype(np.int64)
        if ind.size == 0:
            return                                                                # :764-766
        if ind.min() < 0 or ind.max() >= K:
            raise IndexError("delete: neuron index out of range (0 .. %d)" % (K - 1))
        keep = np.setdiff1d(np.arange(K), ind)
        for name in ("ids", "tags"):                                              # :791-797 (the reference drops them with the neurons)
            val = getattr(self, name, None)
            if val is not None and len(val) == K:
                setattr(self, name, np.asarray(val)[keep])
        C_raw = np.asarray(self.C_raw)[keep] if self.C_raw is not None and np.asarray(self.C_raw).shape[0] == self.A.shape[1] else None
        if getattr(self, "S", None) is not None and np.asarray(self.S).shape[0] == self.A.shape[1]:
            self.S = np.asarray(self.S)[keep]                                     # :801-803
        if self.P.get("kernel_pars") is not None and len(self.P["kernel_pars"]) == self.A.shape[1]:
            self.P["kernel_pars"] = np.asarray(self.P["kernel_pars"])[keep]       # :807-809
        self.set_components(sp.csc_matrix(self.A)[:, keep], np.asarray(self.C)[keep], C_raw)   # :799-806

    def deconvTemporal(self):
        """@Sources2D/deconvTemporal.m:29-105: deconvolve every row of C_raw again (fresh time constants); sets
        C, C_raw, S, P.kernel_pars, P.neuron_sn.  Rows are sharded over ranks and all-reduced when distributed."""
        K = self.C_raw.shape[0]
        if K == 0:
            return self.C_raw.copy()
        v = self.video
        rows = np.arange(K)[v.rank::v.world_size] if (self.dist is not None and v.world_size > 1) else np.arange(K)
        if rows.size == K and hasattr(self.engine, "deconv_temporal_bound") and self.C_raw is getattr(self.engine, "_bound", None):
            # the stitched C_raw is the engine's bound matrix: deconvolved where it lies; C, C_raw - b, S come back lazily (pinned copies behind the kernels)
            C, Craw, S, kp, sn = self.engine.deconv_temporal_bound(self.options.deconv_options)
            self.C, self.C_raw, self.S = C, Craw, S
            self.P["kernel_pars"], self.P["neuron_sn"] = _LazyRow(kp), _LazyRow(sn)
            return C
        if rows.size == K:                                               # not sharded: no row gather / scatter of K x T arrays on the host
            # the ABI writes ck_raw - b into C_raw in place: only a PRIVATE plain array may be handed over -- with deconv_flag = false C_raw
            # is the same object as C / C_prev (and the engine's bound matrix), and a DeviceTraces' host cache must not diverge from its tensor
            private = (isinstance(self.C_raw, np.ndarray) and self.C_raw is not self.C and self.C_raw is not self.C_prev
                       and self.C_raw is not getattr(self.engine, "_bound", None))
            C, Craw, S, kp, sn = self.engine.deconv_temporal(self.C_raw, self.options.deconv_options, overwrite=private)
            self.C, self.C_raw, self.S = C, Craw, S
            self.P["kernel_pars"], self.P["neuron_sn"] = kp, sn
            self._bind_C()
            return C
        # rows k = rank mod world on this rank; ONE collective over the packed [C | C_raw | S | kernel_pars | sn] rows (3T + 2 floats per neuron)
        T = self.C_raw.shape[1]
        pack = np.zeros((K, 3 * T + 2), dtype=np.float32)
        if rows.size:
            c_, r_, s_, k_, n_ = self.engine.deconv_temporal(np.asarray(self.C_raw)[rows], self.options.deconv_options)
            pack[rows, :T], pack[rows, T:2 * T], pack[rows, 2 * T:3 * T], pack[rows, 3 * T], pack[rows, 3 * T + 1] = c_, r_, s_, k_, n_
        pack = self._allreduce(pack)
        C, Craw, S = (np.ascontiguousarray(pack[:, i * T:(i + 1) * T]) for i in range(3))
        kp, sn = pack[:, 3 * T].copy(), pack[:, 3 * T + 1].copy()
        self.C, self.C_raw, self.S = C, Craw, S
        self.P["kernel_pars"], self.P["neuron_sn"] = kp, sn
        self._bind_C()
        return C

    # -- background -------------------------------------------------------------------
    def update_background_parallel(self, use_parallel=True):
        """@Sources2D/update_background_parallel.m:121-146,176-230,311-317 (ring model, bg_ssub = 1)."""
        self._need_data()
        v, o = self.video, self.options
        # the block slices of the current A: prepared by the temporal update under its GPU work when A has not changed since
        cached = getattr(self, "_cur_blocks_src", None) is self.A
        infos = {}
        prefetched = False
        flag_first = self._first_run()                                     # :143, from patch 1 for EVERY patch (a collective when sharded)
        self._prev_blocks = {}                                             # (ind, A_block) per patch: what the temporal update's residual needs
        for idx in v.owned:
            if cached and idx in self._cur_blocks:
                ind_nz, A_block = self._cur_blocks[idx]
            else:
                ind_nz, A_block = self._slice(self.A, idx, "block")       # :128-129
            C_block = self._rows(self.C, ind_nz)                           # :130
            self._prev_blocks[idx] = (ind_nz, A_block)
            # "stop updating B because A&C doesn't change in this area" (:188-199) is decided by the engine's
            # first-run test on W{m}(1,:) exactly like :143; an empty A_block on a later run keeps W, b0.
            if A_block.shape[1] == 0 and not flag_first:
                continue
            getattr(self, "_resid_tag", {}).pop(idx, None)                 # W, b0 of this patch change: its resident residual goes with them
            if not prefetched:                                             # host thread under the first blocking (GIL-free) fit call
                self._prefetch_search_location(); prefetched = True
            if not np.isnan(o.thresh_outlier):                             # :131-138: sn of the block, resized for bg_ssub > 1
                sn_block = np.asarray(self.P["sn"], dtype=np.float32).ravel()[v.block_pix[idx]]
                if self.ssub == 1:
                    self.engine.set_noise(v.pid[idx], sn_block)
                else:
                    b = v.block_pos[idx]
                    nr_b, nc_b = int(b[1] - b[0] + 1), int(b[3] - b[2] + 1)
                    low = sn_block.reshape(nr_b, nc_b, order="F")[np.ix_(nearest_rows(nr_b, self.ssub), nearest_rows(nc_b, self.ssub))] * self.ssub
                    self.engine.set_noise(self.pid_fit[idx], low.reshape(-1, order="F"))
            if self.ssub == 1:
                _, infos[idx] = self.engine.fit_ring_model(v.pid[idx], A_block if A_block.shape[1] else None, C_block,
                                                           o.thresh_outlier, o.bg_acceleration, want_b0=False)   # :218
            else:                                                                                                # :219-230
                _, infos[idx] = self.engine.fit_ring_model_ssub(v.pid[idx], self.pid_fit[idx], self.pid_res[idx], self.ssub,
                                                                A_block if A_block.shape[1] else None, C_block,
                                                                o.thresh_outlier, o.bg_acceleration)
        # :315, evaluated on first read (b0 only changes in this method), so the call returns with the fit still running on the GPU.
        # Sharded: that first read is a collective (all-reduce of the stitched image) -- like every method of this class it must then be
        # made by all ranks at the same point of the program; nobody reading it costs nothing (update_spatial_parallel replaces it).
        self._b0_new_src = "b0"
        self.A_prev = self.A                                               # :316 (no copy needed: A, C are replaced, not mutated)
        self._prev_csr_src = self.A
        self.C_prev = self.C                                               # :317
        return infos

    def _temporal_residual_early(self, idx):
        """update_temporal_parallel.m:149-152 asks for the residual of (A_prev, C_prev) on the block -- both known since the background update.
        The engine only RECORDS that request while the resident Ysig still serves the spatial update (pending footprint term), so
        the host-side part of it is done here, under the spatial sweeps that were just queued; the temporal update then finds it in place."""
        indp, A_prev_b = self._prev_block_of(idx)
        self._residual(idx, A_prev_b if indp.size else None, self._rows(self.C_prev, indp) if indp.size else None,
                       tag=("temporal", self.A_prev, self.C_prev))

    def _temporal_residual_done(self, idx):
        t = getattr(self, "_resid_tag", {}).get(idx)
        return t is not None and t[0] == "temporal" and t[1] is self.A_prev and t[2] is self.C_prev

    def _prev_block_of(self, idx):
        """(ind, A_prev(block rows, ind)) for the neurons of A_prev that touch the block (update_temporal_parallel.m:90-91); cached by
        update_background_parallel when A_prev is the A it fitted against"""
        if getattr(self, "_prev_csr_src", None) is not self.A_prev:
            self._prev_csr_src = self.A_prev
            self._prev_blocks = {}
        hit = self._prev_blocks.get(idx)
        if hit is None:
            hit = self._prev_blocks[idx] = self._slice(self.A_prev, idx, "block")
        return hit

    def _first_run(self):
        """flag_first = (length(unique(W{1}(1,:)))==2)  (update_background_parallel.m:143): the value test on patch 1's W decides the
        "nothing changed here, keep W" skip (:188-199) of EVERY patch.  Sharded: the rank that owns patch 1 evaluates it, one
        all-reduce of a flag hands it to the others (fit_ring_model's own first-run test, :25, stays per patch inside the engine)."""
        v = self.video
        idx1 = v.order[0]
        flag = 0.0
        if idx1 in v.owned:
            flag = float(self.engine.ring_first_run(v.pid[idx1] if self.ssub == 1 else self.pid_fit[idx1]))
        return bool(self._allreduce(np.array([flag], dtype=np.float64))[0] > 0)

    # -- spatial ----------------------------------------------------------------------
    def update_spatial_parallel(self, use_parallel=True, update_sn=False):
        """@Sources2D/update_spatial_parallel.m:61-100,116-216,320-351."""
        self._need_data()
        v, o = self.video, self.options
        sn_new = np.zeros(v.d1 * v.d2, dtype=np.float64) if update_sn else None                  # :101-102
        if o.search_method not in ("ellipse", "dilate"):
            raise NotImplementedError("search_method must be 'ellipse' or 'dilate'")
        K = self.A.shape[1]
        rows, cols, vals = [], [], []
        whole_result = whole_pp = None
        IND = None
        se_now = o.se                                                      # :56 copies the options, :63-65 clears obj.options.se: once per update, here
        if o.search_method == "dilate":
            o.se = None

        def prev_of(idx):
            # A_prev restricted to neurons that touch the HALO only (mask==1 after the patch is set to 2, :84-85,96)
            if v.halo_pix(idx).size:
                indp, _ = self._slice(self.A_prev, idx, "halo")
            else:
                indp = np.zeros(0, dtype=np.int64)
            A_prev_b = self._slice(self.A_prev, idx, "block", cols=indp)[1] if indp.size else None   # :97
            C_prev_b = self._rows(self.C_prev, indp) if indp.size else None                         # :98
            return A_prev_b, C_prev_b

        def masks_of(idx):
            ind, IND_patch = self._slice(IND, idx, "patch")                                          # :87, :89
            if ind.size == 0:
                return ind, IND_patch, None, None
            return ind, IND_patch, self._slice(self.A, idx, "patch", cols=ind)[1], self._rows(self.C, ind)   # :88,199 / :91

        in_flight = []                                                     # (fetch, patch pixels, neurons) of deferred updates not collected yet

        def collect(fetch, pp, ind):
            coo = (fetch(compact=True) if hasattr(fetch, "start") else fetch()).tocoo()
            rows.append(pp[coo.row]); cols.append(ind[coo.col]); vals.append(coo.data)             # :324-334 (patches are disjoint)

        ahead = None                                                       # the next patch's slices, cut while this patch's sweeps run
        for i, idx in enumerate(v.owned):
            pp = v.patch_pix[idx]
            launched = IND is None
            if launched:
                # the residual sweep (:162-166) does not depend on the search mask: start it (the call returns with
                # the kernel in flight) and build IND (:66) on the host underneath it
                A_prev_b, C_prev_b = prev_of(idx)
                self._residual(idx, A_prev_b, C_prev_b)
                IND = self._search_location_csc(se_now)
                ind, IND_patch, A_patch, C_patch = masks_of(idx)
            else:
                (A_prev_b, C_prev_b), (ind, IND_patch, A_patch, C_patch) = ahead
            ahead = None
            nxt = v.owned[i + 1] if i + 1 < len(v.owned) else None
            if ind.size == 0 and not update_sn:
                if nxt is not None:
                    ahead = (prev_of(nxt), masks_of(nxt))
                continue                                                                             # :121-124
            if not launched:
                self._residual(idx, A_prev_b, C_prev_b)                                # :162-166
            sn_patch = self.P["sn"][pp]                                                             # :90,154
            if update_sn:
                sn_patch = self.engine.get_sn(v.pid[idx])                                           # :191-194  sn_patch = GetSn(Ypatch)
                sn_new[pp] = sn_patch
            if ind.size == 0:
                if nxt is not None:
                    ahead = (prev_of(nxt), masks_of(nxt))
                continue                                                                             # :196-199
            param = 20 if o.spatial_algorithm == "nnls" else 3                                      # :203,205,211
            if getattr(self.engine, "supports_lazy_traces", False):
                fetch = self.engine.update_spatial(v.pid[idx], o.spatial_algorithm, A_patch, C_patch, IND_patch,
                                                   sn_patch if o.spatial_algorithm == "hals_thresh" else None, param, defer=True)
                whole = pp.size == v.d1 * v.d2 and ind.size == K
                late = not whole and hasattr(fetch, "start")
                whole_conn = (whole and o.spatial_constraints.get("connected", True) and (self.dist is None or (v.world_size == 1 and not self.force_collectives)))
                if late:
                    # several patches: the download is queued right behind the sweeps and collected `spatial_lag` patches LATE -- a patch's values are assembled
                    # on the host while the next patch's kernels, queued first, keep the device busy (a fetch per patch drained the stream 16 times per update)
                    fetch.start()
                elif whole_conn and hasattr(fetch, "start_connected"):
                    # one patch = the field of view: the connectivity kernel and the downloads are queued NOW, so that what the temporal update's residual request
                    # starts on the device (the deferred half of the ring solve, the W*A_prev tables) runs behind them, under the host's assembly of A
                    fetch.start_connected((v.d1, v.d2))
                self._temporal_residual_early(idx)                       # host work under the sweeps
                if nxt is not None:
                    ahead = (prev_of(nxt), masks_of(nxt))                # ... and the slices of the next patch
                if whole_conn:
                    # one patch = the field of view: post_process_spatial's connectivity constraint (:341) runs on the result where it lies
                    # (the engine's fetch: A without stored zeros, rows sorted, and the raw update as a recipe -- see the early return below)
                    Anew, whole_pp = fetch(connected_fov=(v.d1, v.d2), **({"compact": True} if hasattr(fetch, "start") else {}))
                elif late:
                    in_flight.append((fetch, pp, ind))
                    while len(in_flight) > self.spatial_lag:
                        collect(*in_flight.pop(0))
                    continue
                else:
                    Anew = fetch()
            else:
                Anew = self.engine.update_spatial(v.pid[idx], o.spatial_algorithm, A_patch, C_patch, IND_patch,
                                                  sn_patch if o.spatial_algorithm == "hals_thresh" else None, param)
                if nxt is not None:
                    ahead = (prev_of(nxt), masks_of(nxt))
            if pp.size == v.d1 * v.d2 and ind.size == K:
                whole_result = Anew                                                                  # one patch over the whole FOV, every neuron: already A_
                continue
            collect(lambda Anew=Anew: Anew, pp, ind)
        late_any = bool(in_flight)
        while in_flight:
            collect(*in_flight.pop(0))
        if late_any and hasattr(self.engine, "synchronize"):
            # the late-collected results waited for their own point of the stream only (cnmfe_ticket_wait): what the kernels had to report (a ring over too many
            # footprints, an inconsistent table) must be heard BEFORE obj.A is replaced by what they computed
            self.engine.synchronize()
        d = v.d1 * v.d2
        if whole_result is not None and callable(whole_result):
            self.A_raw = whole_result                                                                # (not sharded: nothing to gather; assembled on first read)
            if update_sn:
                self.P["sn"] = self._allreduce(sn_new).astype(np.float32)
            self.A = whole_pp
            if o.spatial_constraints.get("circular", False):
                from . import hostops
                self.A = hostops.circular_constraints_columns(self.A, v.d1, v.d2)
            self._update_b0_new()
            # the NEXT spatial update's masks depend on this A only: their thread starts under the temporal update's blocking (GIL-free) sweep call, not here, where its
            # Python would share the GIL with the temporal set-up the device is waiting for (CNMFE_PREFETCH_EARLY=1: at once, as until round 6)
            if os.environ.get("CNMFE_PREFETCH_EARLY", "0") == "1":
                self._prefetch_search_location()
            else:
                self._prefetch_wanted = True
            return
        if whole_result is not None:
            A_ = whole_result
        elif rows:
            A_ = _csc_from_triplets(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), (d, K))
        else:
            A_ = sp.csc_matrix((d, K), dtype=np.float32)
        A_ = self._gather_sparse(A_)
        if update_sn:
            self.P["sn"] = self._allreduce(sn_new).astype(np.float32)                               # :336-337 (patches are disjoint)
        A_.eliminate_zeros()
        A_.sort_indices()
        self.A_raw = A_
        if whole_pp is not None:
            whole_pp.eliminate_zeros(); whole_pp.sort_indices()
            self.A = whole_pp                                                                        # :341, done with the fetch
        else:
            self.A = self._post_process(A_) if o.spatial_constraints.get("connected", True) else A_                          # :341, :24-26
        if o.spatial_constraints.get("circular", False):                                            # post_process_spatial.m:28-30
            from . import hostops
            self.A = hostops.circular_constraints_columns(self.A, v.d1, v.d2)
        self._update_b0_new()                                                                        # :347-351
        # (several patches: the mask thread is NOT started here -- its Python would share the GIL with the temporal update's set-up, on the hand-over where the device
        # waits for the host (0.4 ms of a rank's 4.3 at c4); update_temporal_parallel starts it under its sweep call, update_background_parallel otherwise)
        self._prefetch_wanted = True

    def _post_process(self, A_):
        """obj.post_process_spatial() (:341) works on whole footprints, which only exist after the gather.  Sharded, every rank holds the whole gathered A and
        runs the (deterministic) connectivity kernel over ALL of its columns: one 35-us kernel more per rank instead of a second all-gather round of the processed
        columns (rounds 2-5 split the columns k = rank mod world and gathered them again: two collectives + the host-side assembly of the whole A once more, on
        the spatial -> temporal hand-over where the device idles)"""
        v = self.video
        return self.engine.post_process_spatial(A_, v.d1, v.d2)

    def _start_wanted_prefetch(self):
        if getattr(self, "_prefetch_wanted", False):
            self._prefetch_wanted = False
            self._prefetch_search_location()

    def _prefetch_search_location(self):
        """The search mask of the NEXT spatial update depends on A only, which the background update leaves alone: build it
        on a host thread while the (blocking, GIL-free) fit_ring_model call keeps the GPU busy."""
        if self.options.search_method != "ellipse":
            return
        import concurrent.futures as cf
        if getattr(self, "_pool", None) is None:
            self._pool = cf.ThreadPoolExecutor(max_workers=1)
        A = self.A
        fut = getattr(self, "_ind_future", None)
        if fut is not None and fut[0] is A:
            return                                                         # (already on its way: requested at the end of the spatial update that made this A)
        self._ind_future = (A, self._pool.submit(lambda: self._as_mask_csc(self._search_location_owned(A))))

    @staticmethod
    def _as_mask_csc(IND):
        M = sp.csc_matrix(IND, dtype=np.float32)
        M.sort_indices()
        return M

    def _search_location_csc(self, se=_UNSET):
        fut = getattr(self, "_ind_future", None)
        self._ind_future = None
        if fut is not None and fut[0] is self.A:
            return fut[1].result()
        return self._as_mask_csc(self._search_location_owned(se=se))

    def _search_location_owned(self, A=None, se=_UNSET):
        """IND = determine_search_location(obj.A, ...) (:66), evaluated only for the neurons that can reach a patch
        this rank owns (bounding box of the footprint grown by the largest possible ellipse); all other columns are
        empty here and belong to other ranks.  Single rank: every neuron."""
        v, o = self.video, self.options
        A = self.A if A is None else A
        K = A.shape[1]
        if o.search_method == "dilate":
            # update_spatial_parallel.m:56 copies the options BEFORE :63-65 clears obj.options.se: an update uses the element it finds and the
            # next one strel('disk', bSiz, 0) (determine_search_location.m:42-44).  The element is only READ here (prefetches and measurement
            # helpers call this too); update_spatial_parallel clears it once per update.  Host work on every rank (off in every demo).
            from . import hostops
            se = o.se if se is _UNSET else se
            se = hostops.strel_disk(4) if isinstance(se, str) else hostops.strel_disk(o.bSiz) if se is None else se
            return hostops.search_location_dilate(A, v.d1, v.d2, se, o.nb, o.nrgthr)
        if v.world_size == 1:
            return determine_search_location(A, v.d1, v.d2, o.min_size, o.max_size, o.dist)
        A = A.tocsc()
        reach = int(np.ceil(o.dist * o.max_size)) + 2
        cand = np.zeros(K, dtype=bool)
        nzcols = np.nonzero(np.diff(A.indptr) > 0)[0]
        rr, cc = A.indices % v.d1, A.indices // v.d1
        starts = A.indptr[nzcols]
        rmin = np.minimum.reduceat(rr, starts); rmax = np.maximum.reduceat(rr, starts)
        cmin = np.minimum.reduceat(cc, starts); cmax = np.maximum.reduceat(cc, starts)
        for idx in v.owned:
            r0, r1, c0, c1 = [int(x) - 1 for x in v.patch_pos[idx]]
            hit = (rmax + reach >= r0) & (rmin - reach <= r1) & (cmax + reach >= c0) & (cmin - reach <= c1)
            cand[nzcols[hit]] = True
        sel = np.nonzero(cand)[0]
        sub = determine_search_location(A[:, sel], v.d1, v.d2, o.min_size, o.max_size, o.dist).tocoo()
        IND = sp.csc_matrix((sub.data, (sub.row, sel[sub.col])), shape=(v.d1 * v.d2, K))
        IND.sort_indices()
        return IND

    def _gather_sparse(self, A_):
        """all-gather of the per-rank rows of A (disjoint pixel sets, no reduction).  ONE tensor collective per call once the sizes are known: every
        rank sends a padded int32 [3, cap + 1] block (row, column, value bits; the last column carries its count) with cap = the largest count of the previous
        gather + a quarter; a rank that outgrew cap is seen by everybody in the gathered counts, and everybody repeats the round with the capacity those counts ask
        for (the first call of a run costs two rounds: it starts from cap = 0)."""
        if self.dist is None or (self.video.world_size == 1 and not self.force_collectives):
            return A_
        import torch
        import torch.distributed as td
        coo = A_.tocoo()
        nccl = td.get_backend(self.dist) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        W = td.get_world_size(self.dist)                                  # (= video.world_size in a run; scripts/rank_load.py times one rank's share behind a group of one)
        n = int(coo.nnz)
        cap = int(getattr(self, "_gather_cap", 0))
        while True:
            # the padded block is put together by NumPy, not by torch CPU kernels: a torch fill / copy of this size opens an OpenMP region over every core of
            # the host, whose threads then spin -- inside a CPU-quota'd container that throttled the whole process for ~40 ms of every 100-ms scheduler period
            # (one rank of RCCL at c4: 26 -> 50-58 ms per iteration; OMP_NUM_THREADS=1 made it vanish: profiles/r04/forced_collectives.txt)
            buf_np = np.zeros((3, cap + 1), dtype=np.int32)
            m = min(n, cap)
            if n <= cap:
                buf_np[0, :m] = coo.row
                buf_np[1, :m] = coo.col
                buf_np[2, :m] = np.ascontiguousarray(coo.data, dtype=np.float32).view(np.int32)
            buf_np[0, cap] = n
            buf = torch.from_numpy(buf_np).to(dev)
            out = torch.empty((W, 3, cap + 1), dtype=torch.int32, device=dev)
            td.all_gather_into_tensor(out, buf, group=self.dist) if nccl else td.all_gather(list(out.unbind(0)), buf, group=self.dist)
            out = out.cpu().numpy()
            sizes = [int(out[w, 0, cap]) for w in range(W)]
            nmax = max(sizes)
            if nmax <= cap:
                break
            cap = ((nmax + nmax // 4 + 255) // 256) * 256                     # (the same number on every rank: it comes from the gathered counts)
        self._gather_cap = max(cap, ((nmax + nmax // 4 + 255) // 256) * 256) if nmax > (cap * 3) // 4 else cap      # (grown ahead of a matrix that fills up)
        r = np.concatenate([out[w, 0, :m_] for w, m_ in zip(range(W), sizes)])            # int32 row / column indices: the CSC build is index-bound
        c = np.concatenate([out[w, 1, :m_] for w, m_ in zip(range(W), sizes)])
        d_ = np.concatenate([np.ascontiguousarray(out[w, 2, :m_]).view(np.float32) for w, m_ in zip(range(W), sizes)])
        return _csc_from_triplets(r, c, d_, A_.shape)

    # -- objective ----------------------------------------------------------------------
    def compute_RSS(self):
        """[RSS_total, RSS] = compute_RSS(obj)  (@Sources2D/Sources2D.m:1358-1510), ring model, all frames.
        Per patch the engine needs the residual of (A_prev, C_prev) on the block (:1427-1429, :1475) -- the one the temporal update asks for, so
        right after update_temporal_parallel it is still resident (or pending) and this costs one read of Ysig per patch."""
        self._need_data()
        v = self.video
        b0_ = self.reconstruct_b0().reshape(-1, order="F")                                           # :1398 (a collective when sharded)
        b0_new_ = np.asarray(self.b0_new, dtype=np.float64).reshape(-1, order="F")                   # :1399
        RSS = {}
        for idx in v.owned:
            pp, bp = v.patch_pix[idx], v.block_pix[idx]
            ind, _ = self._slice(self.A, idx, "block")                                              # :1423
            indp, A_prev_b = self._prev_block_of(idx)                                                # :1427-1428
            A_pp = self._slice(self.A, idx, "patch", cols=ind)[1] if ind.size else None              # A_patch(ind_patch, :)  (:1467)
            if self.ssub != 1:                                                                       # :1479-1486 ('nearest' both ways)
                self.engine.background_ssub(v.pid[idx], self.pid_fit[idx], self.ssub, A_prev_b if indp.size else None,
                                            self._rows(self.C_prev, indp) if indp.size else None, b0_[bp])
                RSS[idx] = self.engine.compute_rss_ssub(v.pid[idx], A_pp, self._rows(self.C, ind) if ind.size else None, b0_new_[pp])
                continue
            self._residual(idx, A_prev_b if indp.size else None, self._rows(self.C_prev, indp) if indp.size else None)
            RSS[idx] = self.engine.compute_rss(v.pid[idx], A_pp, self._rows(self.C, ind) if ind.size else None, b0_[bp], b0_new_[pp])
        total = float(self._allreduce(np.array([sum(RSS.values())], dtype=np.float64))[0])           # :1507-1508
        self.P["RSS"] = total                                                                        # :1509
        return total, RSS

    def reconstruct_background(self, frame_range=None):
        """Ybg = reconstruct_background(obj, frame_range)  (@Sources2D/Sources2D.m:1247-1355), ring model: d1 x d2 x T' (fp32) of the
        owned patches (zeros elsewhere when sharded, like every per-patch output).  frame_range = (first, last), 1-based inclusive as in MATLAB."""
        self._need_data()
        v = self.video
        T = self.C.shape[1]
        f0, f1 = (1, T) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
        b0_ = self.reconstruct_b0().reshape(-1, order="F")                                           # :1292
        b0_new_ = np.asarray(self.b0_new, dtype=np.float64).reshape(-1, order="F")                   # :1293
        Ybg = np.zeros((v.d1 * v.d2, f1 - f0 + 1), dtype=np.float32)                                 # :1297
        for idx in v.owned:
            pp, bp = v.patch_pix[idx], v.block_pix[idx]
            indp, A_prev_b = self._prev_block_of(idx)                                                # :1317-1320
            if self.ssub != 1:                                                                       # :1325-1334
                self.engine.background_ssub(v.pid[idx], self.pid_fit[idx], self.ssub, A_prev_b if indp.size else None,
                                            self._rows(self.C_prev, indp) if indp.size else None, b0_[bp])
            else:
                self._residual(idx, A_prev_b if indp.size else None, self._rows(self.C_prev, indp) if indp.size else None)
            for t0 in range(f0 - 1, f1, 4096):                                                       # (the ABI hands out at most 65535 frames per call)
                n = min(4096, f1 - t0)
                if self.ssub != 1:
                    Ybg[pp, t0 - (f0 - 1):t0 - (f0 - 1) + n] = self.engine.reconstruct_background_ssub(v.pid[idx], b0_new_[pp], t0, n).T
                else:
                    Ybg[pp, t0 - (f0 - 1):t0 - (f0 - 1) + n] = self.engine.reconstruct_background(v.pid[idx], b0_[bp], b0_new_[pp], t0, n).T
        return Ybg.reshape(v.d1, v.d2, -1, order="F")

    # -- temporal -----------------------------------------------------------------------
    def update_temporal_parallel(self, use_parallel=True, use_c_hat=True):
        """@Sources2D/update_temporal_parallel.m:62-94,112-186,264-295.  The per-patch traces never leave the device: every patch's
        aa .* C_raw is added to the engine's stitch accumulator (:269-278), sharded runs all-reduce that buffer in place (the
        overlap-region stitch, ONE collective), and :279-286 run on the device; one K x T copy comes back for obj.C_raw / obj.C."""
        self._need_data()
        v, o = self.video, self.options
        eng = self.engine
        K, T = self.C.shape
        launched_any = False
        sharded = self.dist is not None and (v.world_size > 1 or self.force_collectives)
        eng.stitch_begin(K, T)
        # several patches on this rank: every patch's update is set up first and the Gauss-Seidel sweeps run level by level ACROSS the patches (they are
        # independent: the reference's parfor, :112-186) -- one patch's level is a few workgroups as long as one trace's work
        batch = use_c_hat and len(v.owned) > 1 and getattr(eng, "supports_temporal_jobs", False)
        jobs = []
        for idx in v.owned:
            pp, bp = v.patch_pix[idx], v.block_pix[idx]
            indp, A_prev_b = self._prev_block_of(idx)                                                # :90-91
            C_prev_b = self._rows(self.C_prev, indp) if indp.size else None
            launched = not launched_any
            whole = bp.size == v.d1 * v.d2                                 # this block is the whole field of view (then so is the patch): no row slicing
            if launched:
                # the sweep (:149-152) only needs (A_prev, C_prev): start it, slice the current A underneath it
                if not self._temporal_residual_done(idx):
                    self._residual(idx, A_prev_b if indp.size else None, C_prev_b)
                launched_any = True
                self._cur_blocks, self._cur_blocks_src = {}, self.A
            if whole:
                A_csc = self.A if sp.isspmatrix_csc(self.A) else self.A.tocsc()
                ind = np.nonzero(np.asarray(A_csc.sum(axis=0)).ravel() > 0)[0]                       # :83
                A_pp = A_csc if ind.size == K else A_csc[:, ind]
                self._cur_blocks[idx] = (ind, A_pp)
            else:
                # (ind, A(block, ind)): also what the next background update fits against (update_background_parallel.m:128-130); A(patch, ind) in the same pass
                ind, A_blk, A_pp = self._slice_block_patch(self.A, idx)                              # :83, A_patch(ind_patch,:)
                self._cur_blocks[idx] = (ind, A_blk)
            if ind.size == 0:
                continue                                                                              # :123
            if not launched and not self._temporal_residual_done(idx):
                self._residual(idx, A_prev_b if indp.size else None, C_prev_b)          # :149-152
            C_patch = self._rows(self.C, ind)                                                        # :86
            if not use_c_hat:                                                                         # :174-175
                eng.fast_temporal(v.pid[idx], A_pp, want_raw=False)
            elif batch:
                jobs.append((eng.hals_temporal_job(v.pid[idx], A_pp, C_patch, o.maxIter, o.deconv_options if o.deconv_flag else None), ind))
                continue
            elif o.deconv_flag:                                                                       # :106-110
                self._start_wanted_prefetch()
                eng.hals_temporal_deconv(v.pid[idx], A_pp, C_patch, o.maxIter, o.deconv_options, want_all=None)
            else:
                self._start_wanted_prefetch()
                eng.hals_temporal(v.pid[idx], A_pp, C_patch, o.maxIter, want_C=False, want_raw=False)  # :180-181
            eng.stitch_add(ind)                                                                       # :274-275
        if jobs:
            self._start_wanted_prefetch()                                  # host thread under the (GIL-free) sweep / stitch / fit calls that follow
            eng.temporal_jobs_sweep()
            for job, ind in jobs:
                eng.stitch_add_job(job, ind)                                                          # :274-275
        if sharded:                                                        # the overlap-region stitch: ONE all-reduce, in place on the device
            eng.stitch_allreduce(self.dist)
        # without deconvolution nobody needs the values on the host right away: the engine keeps the matrix bound and streams a copy into pinned
        # memory behind the kernels (LazyHostTraces); the next background fit is set up while the temporal sweep is still running
        lazy = (not o.deconv_flag) and getattr(eng, "supports_lazy_traces", False)
        bound_deconv = o.deconv_flag and not sharded and hasattr(eng, "deconv_temporal_bound") and getattr(eng, "supports_lazy_traces", False)
        C_raw = eng.stitch_finish(subtract_min=not o.deconv_flag, want="lazy" if lazy else ("bound" if bound_deconv else True))       # :279-280, :285
        if o.deconv_flag:                                                                             # :282-283  obj.C = obj.deconvTemporal()
            self.C_raw = C_raw
            self.C = self.deconvTemporal()
        else:
            self.C_raw = C_raw
            self.C = self.C_raw                                                                       # :286 (already the engine's bound matrix)
        self._update_b0_new()                                                                         # :291-295
