"""CPU oracle for the CNMF-E factor-update hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a float64 NumPy/SciPy *restatement* of the reference's algorithm
(zhoupc/CNMF_E @ 2024_08_07, MATLAB) for the background -> spatial -> temporal
update loop.  Every function cites the reference file:line it follows.  It is
the checker for the HIP engine: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``cnmf_e_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED: the reference is MATLAB, there is no MATLAB/Octave in the build
container, and the reference ships no tests / golden vectors for this path
(SURVEY.md section 4, 8(c)).  The oracle is therefore pinned only by algebraic
self-checks (tests/test_oracle_*.py): normal equations vs ``numpy.linalg``,
HALS monotone objective, NNLS KKT conditions, planted-model recovery.

Conventions: everything is column-major like MATLAB.  A "frame image" of an
``nr x nc`` region is flattened with the row index fastest, so the pixel linear
index is ``c*nr + r`` (0-based).  ``Y`` is ``d x T`` (pixels x frames).
Positions ``[r0, r1, c0, c1]`` are 1-based and inclusive exactly as in
``distribute_data.m`` so that they can be compared against the reference by eye.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "get_nhood", "distribute_geometry", "build_ring_W", "fit_ring_model",
    "residual_ysig", "HALS_spatial", "HALS_spatial_thresh", "nnls", "nnls_spatial",
    "HALS_temporal", "com", "determine_search_location", "connectivity_constraint",
    "post_process_spatial", "OracleSources2D",
]


# --------------------------------------------------------------------------- #
# geometry
# --------------------------------------------------------------------------- #
def _matlab_round(x):
    """MATLAB round(): half away from zero."""
    x = np.asarray(x, dtype=np.float64)
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def get_nhood(radius, k=None):
    """Ring offsets (r_shift, c_shift).  endoscope/get_nhood.m:1-25."""
    rsub = np.arange(-radius, radius + 1)
    cind, rind = np.meshgrid(rsub, rsub)                     # get_nhood.m:9
    R = np.sqrt(cind.astype(np.float64) ** 2 + rind ** 2)    # :10
    kernel = (R >= radius) & (R < radius + 1)                # :11
    # MATLAB find() walks column-major: column index slow, row index fast (:13)
    c_idx, r_idx = np.nonzero(kernel.T)
    r_shift = r_idx - radius
    c_shift = c_idx - radius
    if k is None or k > r_shift.size:                        # :17
        return r_shift.astype(np.int64), c_shift.astype(np.int64)
    temp = np.arctan2(r_shift, c_shift)                      # :20
    ids = np.argsort(temp, kind="stable")                    # :21
    ind = _matlab_round(np.linspace(1, ids.size, k)).astype(np.int64) - 1  # :22
    return r_shift[ids[ind]].astype(np.int64), c_shift[ids[ind]].astype(np.int64)


def distribute_geometry(d1, d2, patch_dims, w_overlap):
    """patch_pos / block_pos (1-based inclusive).  endoscope/distribute_data.m:38-100,161-173."""
    min_patch_width = 2 * w_overlap + 3                      # :40
    pd = np.atleast_1d(np.asarray(patch_dims, dtype=np.float64)).copy()
    pd[pd < min_patch_width] = min_patch_width               # :47
    if pd.size == 1:
        pd = np.array([pd[0], pd[0]])
    pd = pd[:2]
    nr_patch = int(_matlab_round(d1 / pd[0]))                # :57
    nc_patch = int(_matlab_round(d2 / pd[1]))

    def _idx(n_patch, dd, fix_last):
        if n_patch <= 1:
            return np.array([1, dd], dtype=np.int64)
        idx = np.ceil(np.linspace(1, dd, n_patch + 1)).astype(np.int64)   # :62 / :74
        if fix_last:
            idx[-1] = dd                                     # :63 (rows only, as in the reference)
        if idx[1] - idx[0] < min_patch_width:                # :64 / :75
            idx = np.arange(1, dd + 1, min_patch_width, dtype=np.int64)
            idx[-1] = dd
        return idx

    pr = _idx(nr_patch, d1, True)
    pc = _idx(nc_patch, d2, False)
    nr_patch = pr.size - 1
    nc_patch = pc.size - 1
    patch_pos = np.empty((nr_patch, nc_patch), dtype=object)
    block_pos = np.empty((nr_patch, nc_patch), dtype=object)
    for m in range(nr_patch):
        for n in range(nc_patch):
            patch_pos[m, n] = np.array([
                pr[m], pr[m + 1] - (1 if m != nr_patch - 1 else 0),
                pc[n], pc[n + 1] - (1 if n != nc_patch - 1 else 0)], dtype=np.int64)    # :167
            block_pos[m, n] = np.array([
                max(1, pr[m] - w_overlap - 1), min(d1, pr[m + 1] + w_overlap),
                max(1, pc[n] - w_overlap - 1), min(d2, pc[n + 1] + w_overlap)], dtype=np.int64)  # :168-169
    return patch_pos, block_pos


def build_ring_W(patch, block, d1, d2, r_shift, c_shift):
    """Initial ring matrix W (d_patch x d_block, uniform 1/count rows).

    @Sources2D/initComponents_parallel.m:222-236 (dup. update_background_parallel.m:88-99).
    """
    r0, r1, c0, c1 = [int(v) for v in patch]
    br0, br1, bc0, bc1 = [int(v) for v in block]
    nr, nc = r1 - r0 + 1, c1 - c0 + 1
    nr_b, nc_b = br1 - br0 + 1, bc1 - bc0 + 1
    csub, rsub = np.meshgrid(np.arange(c0, c1 + 1), np.arange(r0, r1 + 1))   # :223
    csub = csub.reshape(-1, 1, order="F")
    rsub = rsub.reshape(-1, 1, order="F")
    ii = np.repeat(np.arange(nr * nc)[:, None], r_shift.size, axis=1)        # :226
    csub = csub + c_shift[None, :]
    rsub = rsub + r_shift[None, :]
    ind = (csub >= 1) & (csub <= d2) & (rsub >= 1) & (rsub <= d1)           # :229
    jj = (csub - bc0) * nr_b + (rsub - br0 + 1) - 1                          # :230 (0-based)
    temp = sp.csr_matrix((np.ones(int(ind.sum())), (ii[ind], jj[ind])),
                         shape=(nr * nc, nr_b * nc_b))                       # :232
    temp.sum_duplicates()
    temp.sort_indices()
    cnt = np.asarray(temp.sum(axis=1)).ravel()
    return sp.diags(1.0 / cnt) @ temp                                        # :233


# --------------------------------------------------------------------------- #
# imresize (MathWorks Image Processing Toolbox; NOT in /root/reference: restated from its documented algorithm --
# separable resampling with `contributions`: output sample x maps to u = x/scale + 0.5*(1 - 1/scale), the kernel is
# stretched by 1/scale when shrinking with antialiasing, taps are mirrored at the borders and each row of weights is
# normalised to sum 1.  Bicubic = Keys a=-0.5; 'nearest' = box kernel, no antialiasing.)  PARITY UNPINNED.
# Used only for bg_ssub > 1 (update_background_parallel.m:224, update_spatial_parallel.m:169-176).
# --------------------------------------------------------------------------- #
def _cubic(x):
    ax = np.abs(x); ax2 = ax * ax; ax3 = ax2 * ax
    return (1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1) + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) & (ax <= 2))


def imresize_weights(in_len, out_len, scale, method="bicubic"):
    """Dense (out_len x in_len) weight matrix of MATLAB's imresize along one dimension."""
    if method == "nearest":
        kernel, kw, antialias = (lambda x: ((x >= -0.5) & (x < 0.5)).astype(np.float64)), 1.0, False
    else:
        kernel, kw, antialias = _cubic, 4.0, True
    if scale < 1 and antialias:
        h = lambda x: scale * kernel(scale * x)
        kw = kw / scale
    else:
        h = kernel
    x = np.arange(1, out_len + 1, dtype=np.float64)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(u - kw / 2)
    P = int(np.ceil(kw)) + 2
    ind = left[:, None] + np.arange(P)[None, :]
    w = h(u[:, None] - ind)
    w = w / w.sum(axis=1, keepdims=True)
    aux = np.concatenate([np.arange(1, in_len + 1), np.arange(in_len, 0, -1)])
    ind = aux[np.mod(ind.astype(np.int64) - 1, aux.size)]                 # mirrored, 1-based
    M = np.zeros((out_len, in_len))
    for o in range(out_len):
        np.add.at(M[o], ind[o] - 1, w[o])
    return M


def imresize_scale(img, scale, method="bicubic"):
    """imresize(img, scale[, method]) on the first two dims of a 2-D/3-D array; output size ceil(in*scale)."""
    nr, nc = img.shape[:2]
    Mr = imresize_weights(nr, int(np.ceil(nr * scale)), scale, method)
    Mc = imresize_weights(nc, int(np.ceil(nc * scale)), scale, method)
    out = np.tensordot(Mr, img, axes=(1, 0))                              # rows first (equal scales: order [1 2])
    return np.moveaxis(np.tensordot(Mc, out, axes=(1, 1)), 0, 1)


def imresize_size(img, out_size, method="bicubic"):
    """imresize(img, [nr nc][, method]): per-dimension scale = out/in."""
    nr, nc = img.shape[:2]
    Mr = imresize_weights(nr, out_size[0], out_size[0] / nr, method)
    Mc = imresize_weights(nc, out_size[1], out_size[1] / nc, method)
    out = np.tensordot(Mr, img, axes=(1, 0))
    return np.moveaxis(np.tensordot(Mc, out, axes=(1, 1)), 0, 1)


def build_ring_W_ssub(nr_block, nc_block, bg_ssub, r_shift, c_shift):
    """W on the downsampled block grid (initComponents_parallel.m:237-251): d1s*d2s square, neighbours clipped at the BLOCK."""
    d1s, d2s = int(np.ceil(nr_block / bg_ssub)), int(np.ceil(nc_block / bg_ssub))
    csub, rsub = np.meshgrid(np.arange(1, d2s + 1), np.arange(1, d1s + 1))
    csub = csub.reshape(-1, 1, order="F"); rsub = rsub.reshape(-1, 1, order="F")
    ii = np.repeat(np.arange(d1s * d2s)[:, None], r_shift.size, axis=1)
    csub = csub + c_shift[None, :]; rsub = rsub + r_shift[None, :]
    jj = (csub - 1) * d1s + rsub - 1
    ind = (csub >= 1) & (csub <= d2s) & (rsub >= 1) & (rsub <= d1s)
    temp = sp.csr_matrix((np.ones(int(ind.sum())), (ii[ind], jj[ind])), shape=(d1s * d2s, d1s * d2s))
    temp.sum_duplicates(); temp.sort_indices()
    cnt = np.asarray(temp.sum(axis=1)).ravel()
    return sp.diags(1.0 / cnt) @ temp


def residual_ysig_ssub(Y_block, A_prev, C_prev, W, b0, ind_patch, nr_block, nc_block, bg_ssub):
    """update_spatial_parallel.m:167-178 (= update_temporal_parallel.m:153-165): the ring product on the downsampled block."""
    Yb = np.asarray(Y_block, dtype=np.float64)
    T = Yb.shape[1]
    tmp_Y = Yb - (A_prev @ np.asarray(C_prev, dtype=np.float64)) if (A_prev is not None and A_prev.shape[1] > 0) else Yb.copy()
    temp = (tmp_Y - tmp_Y.mean(axis=1, keepdims=True)).reshape(nr_block, nc_block, T, order="F")    # :171
    temp = imresize_scale(temp, 1.0 / bg_ssub)                                                          # :172
    d1s, d2s = temp.shape[:2]
    Bf = (sp.csr_matrix(W) @ temp.reshape(d1s * d2s, T, order="F")).reshape(d1s, d2s, T, order="F")      # :173
    Bf = imresize_size(Bf, (nr_block, nc_block)).reshape(nr_block * nc_block, T, order="F")             # :174-175
    ind_patch = np.asarray(ind_patch, dtype=bool).ravel()
    return Yb[ind_patch, :] - Bf[ind_patch, :] - np.asarray(b0, dtype=np.float64).ravel()[:, None]       # :177


def ind_patch_mask(patch, block):
    """Logical nr_b x nc_b mask of patch pixels inside the block, flattened column-major.

    update_spatial_parallel.m:140-141.
    """
    r0, r1, c0, c1 = [int(v) for v in patch]
    br0, br1, bc0, bc1 = [int(v) for v in block]
    m = np.zeros((br1 - br0 + 1, bc1 - bc0 + 1), dtype=bool)
    m[r0 - br0:r1 - br0 + 1, c0 - bc0:c1 - bc0 + 1] = True
    return m.reshape(-1, order="F")


# --------------------------------------------------------------------------- #
# background: ring model
# --------------------------------------------------------------------------- #
def fit_ring_model(Y, A, C, W_old, thresh_outlier, sn, ind_patch, with_projection=True, only_rows=None):
    """[W, b0] = fit_ring_model(...).  endoscope/fit_ring_model.m:1-127.

    Y: d_b x T, A: d_b x K (dense or sparse), C: K x T, W_old: sparse d x d_b.
    With a finite thresh_outlier the outlier branch (:50-56) and the frame selection (:62-67) run (no demo sets it).
    only_rows (test harness only, not in the reference): regress just these patch pixels -- the per-pixel regressions of :92-108 are
    independent, so a sample of rows of W at a size where all of them would take hours; the other rows keep W_old.
    """
    Y = np.asarray(Y)
    d_b, T = Y.shape
    if only_rows is not None and np.isnan(thresh_outlier) and d_b * T > (1 << 28):
        return _fit_ring_model_rows(Y, A, C, W_old, ind_patch, with_projection, only_rows)
    if A is None or A.shape[1] == 0:
        A = np.ones((d_b, 1)); C = np.zeros((1, T))          # :15-17 (isempty(A))
    if sp.issparse(A):
        A = A.toarray()                                      # :18-23
    A = np.asarray(A, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    W_old = sp.csr_matrix(W_old)
    W_old.sort_indices()
    # first run? (:25): number of distinct values in row 1, implicit zeros included
    row0 = W_old.getrow(0)
    vals = set(np.unique(row0.data).tolist())
    if row0.nnz < W_old.shape[1]:
        vals.add(0.0)
    if len(vals) == 2:
        ind_active = np.ones(W_old.shape[0], dtype=bool)     # :26
    else:
        ind_active = np.asarray(abs(W_old) @ A.sum(axis=1)).ravel() > 0     # :28
    if ind_patch is None:
        ind_patch = np.ones(d_b, dtype=bool)                 # :35
    ind_patch = np.asarray(ind_patch, dtype=bool).ravel()

    Ymean = Y.astype(np.float64).mean(axis=1)                # :42
    Cmean = C.mean(axis=1)                                   # :43
    b0 = Ymean[ind_patch] - A[ind_patch, :] @ Cmean          # :44
    Yc = Y.astype(np.float64) - Ymean[:, None]               # :45
    Cc = C - Cmean[:, None]                                  # :46
    Bf = Yc - A @ Cc                                         # :47

    outl = not np.isnan(thresh_outlier)
    if outl:                                                 # :50
        Bf_old = W_old @ Bf                                  # :51
        tmp_Bf = Bf[ind_patch, :]                            # :52
        ind_outlier = tmp_Bf > Bf_old + thresh_outlier * np.asarray(sn, dtype=np.float64).reshape(-1, 1)   # :53
        tmp_Bf[ind_outlier] = Bf_old[ind_outlier]            # :54
        Bf[ind_patch, :] = tmp_Bf                            # :55

    pmax = int(np.max(np.asarray((W_old > 0).sum(axis=1)).ravel()))          # :60
    nmax = pmax * 100                                        # :61
    if outl and nmax < T:                                    # :62
        temp = ind_outlier.sum(axis=0)                       # :63 (column sums: outliers per frame)
        ind_frames = temp <= matlab_quantile(temp, nmax / T)  # :64
        nmax = int(ind_frames.sum())                         # :65
        Bf = Bf[:, ind_frames]                               # :66
    ind_pixels = np.nonzero(ind_patch)[0]                    # :71
    d = ind_pixels.size
    W = W_old.copy().tolil()
    T = Bf.shape[1]
    if with_projection:
        nk = min(int(_matlab_round(T / 1)), nmax)            # :84
        k = T // nk                                          # :85
        if k != 1:
            Bf = Bf[:, ::k]                                  # :87
    vec_ones = np.ones((1, Bf.shape[1]))                     # :91 / :69
    indptr, indices, data = W_old.indptr, W_old.indices, W_old.data
    new_data = data.copy()
    _t_loop = __import__("time").time()
    for m in (range(d) if only_rows is None else only_rows):  # :92 / :112
        if not ind_active[m]:
            continue
        idx = ind_pixels[m]
        sl = slice(indptr[m], indptr[m + 1])
        nzmask = data[sl] != 0                               # :99 ind_ring = (W_old(m,:)~=0)
        ring = indices[sl][nzmask]
        y = Bf[idx, :]
        X = np.vstack([Bf[ring, :], vec_ones])               # :101
        XX = X @ X.T                                         # :103
        Xy = X @ y                                           # :104
        w = np.linalg.solve(XX + np.eye(XX.shape[0]) * np.trace(XX) * 1e-5, Xy)   # :106
        pos = np.nonzero(nzmask)[0] + indptr[m]
        new_data[pos] = w[:-1] + 1e-100                      # :107
    fit_ring_model.last_loop_seconds = __import__("time").time() - _t_loop      # (timing harness only: bench.py --cpu-baseline full separates the loop from the set-up)
    W = sp.csr_matrix((new_data, indices.copy(), indptr.copy()), shape=W_old.shape)
    return W, b0


def _fit_ring_model_rows(Y, A, C, W_old, ind_patch, with_projection, only_rows):
    """TEST HARNESS ONLY (not in the reference): fit_ring_model's regressions (:92-108) for a SAMPLE of patch pixels of a block too large to form the dense
    d_b x T float64 matrices of :42-47 for (512 x 512 x 10000: two of 21 GB and a 1.3 TFLOP dense A*C).  The same statements as above, evaluated on the pixels
    the sampled regressions read -- the sampled pixels and their ring neighbours -- with A kept sparse; Ymean and b0 (:42-44) for every pixel.  thresh_outlier
    = NaN only (what every demo runs)."""
    d_b, T = Y.shape
    a_empty = A is None or A.shape[1] == 0
    if a_empty:
        A = sp.csr_matrix(np.ones((d_b, 1))); C = np.zeros((1, T))          # :15-17
    A = sp.csr_matrix(A, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    W_old = sp.csr_matrix(W_old); W_old.sort_indices()
    row0 = W_old.getrow(0)
    vals = set(np.unique(row0.data).tolist())
    if row0.nnz < W_old.shape[1]:
        vals.add(0.0)
    if len(vals) == 2:
        ind_active = np.ones(W_old.shape[0], dtype=bool)     # :26
    else:
        ind_active = np.asarray(abs(W_old) @ np.asarray(A.sum(axis=1)).ravel()).ravel() > 0     # :28
    if ind_patch is None:
        ind_patch = np.ones(d_b, dtype=bool)
    ind_patch = np.asarray(ind_patch, dtype=bool).ravel()
    Ymean = np.mean(Y, axis=1, dtype=np.float64)             # :42
    Cmean = C.mean(axis=1)                                   # :43
    b0 = Ymean[ind_patch] - np.asarray(A[ind_patch] @ Cmean).ravel()        # :44
    Cc = C - Cmean[:, None]                                  # :46
    pmax = int(np.max(np.asarray((W_old > 0).sum(axis=1)).ravel()))          # :60
    nmax = pmax * 100                                        # :61
    ind_pixels = np.nonzero(ind_patch)[0]                    # :71
    indptr, indices, data = W_old.indptr, W_old.indices, W_old.data
    rows = [m for m in only_rows if ind_active[m]]
    rings = {}
    for m in rows:
        sl = slice(indptr[m], indptr[m + 1])
        rings[m] = indices[sl][data[sl] != 0]                # :99
    need = np.unique(np.concatenate([ind_pixels[np.asarray(rows, dtype=np.int64)]] + [rings[m] for m in rows])) if rows else np.zeros(0, dtype=np.int64)
    pos_of = {int(q): i for i, q in enumerate(need)}
    Bf = (np.asarray(Y[need], dtype=np.float64) - Ymean[need, None]) - np.asarray(A[need] @ Cc)     # :45-47 on the needed pixels
    if with_projection:
        nk = min(int(_matlab_round(T / 1)), nmax)            # :84
        k = T // nk                                          # :85
        if k != 1:
            Bf = Bf[:, ::k]                                  # :87
    vec_ones = np.ones((1, Bf.shape[1]))
    new_data = data.copy()
    for m in rows:                                           # :92
        sl = slice(indptr[m], indptr[m + 1])
        nzmask = data[sl] != 0
        y = Bf[pos_of[int(ind_pixels[m])], :]
        X = np.vstack([Bf[[pos_of[int(q)] for q in rings[m]], :], vec_ones])             # :101
        XX = X @ X.T                                         # :103
        Xy = X @ y                                           # :104
        w = np.linalg.solve(XX + np.eye(XX.shape[0]) * np.trace(XX) * 1e-5, Xy)          # :106
        new_data[np.nonzero(nzmask)[0] + indptr[m]] = w[:-1] + 1e-100                   # :107
    return sp.csr_matrix((new_data, indices.copy(), indptr.copy()), shape=W_old.shape), b0


def matlab_quantile(x, p):
    """quantile(x, p) of the Statistics Toolbox for a vector (a MathWorks function, not in the repository; restated from its documented
    definition: the sorted values are the 0.5/n, 1.5/n, ..., (n-0.5)/n quantiles, linear interpolation between them, the minimum /
    maximum outside).  Used at fit_ring_model.m:64.  PARITY UNPINNED."""
    x = np.sort(np.asarray(x, dtype=np.float64).ravel())
    n = x.size
    r = p * n + 0.5                                          # 1-based fractional rank
    if r <= 1:
        return x[0]
    if r >= n:
        return x[-1]
    lo = int(np.floor(r))
    return x[lo - 1] + (r - lo) * (x[lo] - x[lo - 1])


def ring_frame_stride(W_old, T, with_projection=True):
    """The subsampling stride k of fit_ring_model.m:60,84-87 (helper for tests)."""
    pmax = int(np.max(np.asarray((sp.csr_matrix(W_old) > 0).sum(axis=1)).ravel()))
    if not with_projection:
        return 1
    nk = min(T, pmax * 100)
    return T // nk


# --------------------------------------------------------------------------- #
# residual / background subtraction  (the "R1" expression)
# --------------------------------------------------------------------------- #
def residual_ysig(Y_block, A_prev, C_prev, W, b0, ind_patch, only_rows=None):
    """Ysig = Y(patch,:) - W*(Y - A_prev*C_prev) - (b0 - W*mean(...,2)).

    @Sources2D/update_spatial_parallel.m:162-166 (= update_temporal_parallel.m:149-152).
    only_rows (test harness only): just these patch pixels' rows of the result (rows of the expression are independent).
    """
    if only_rows is not None and np.asarray(Y_block).size > (1 << 28):
        # TEST HARNESS ONLY: the same expression for sampled rows of a block too large for the dense d_b x T float64 copies -- evaluated on the block pixels
        # the sampled rows of W read (the rows of the expression are independent, a row needs its own pixel and its ring neighbours)
        rows = np.asarray(only_rows)
        ipx = np.nonzero(np.asarray(ind_patch, dtype=bool).ravel())[0]
        Wr = sp.csr_matrix(W)[rows]
        need = np.unique(np.concatenate([Wr.indices, ipx[rows]]))
        pos = np.full(np.asarray(Y_block).shape[0], -1, dtype=np.int64); pos[need] = np.arange(need.size)
        tmp = np.asarray(Y_block[need], dtype=np.float64)
        Yrows = tmp[pos[ipx[rows]]].copy()
        if A_prev is not None and A_prev.shape[1] > 0:
            tmp = tmp - np.asarray(sp.csr_matrix(A_prev)[need] @ np.asarray(C_prev, dtype=np.float64))       # :163
        Wn = sp.csr_matrix((Wr.data, pos[Wr.indices], Wr.indptr), shape=(rows.size, need.size))
        b0r = np.asarray(b0, dtype=np.float64).ravel()[rows]
        return (Yrows - Wn @ tmp) - (b0r - Wn @ tmp.mean(axis=1))[:, None]                                   # :166
    Yb = np.asarray(Y_block, dtype=np.float64)
    if A_prev is not None and A_prev.shape[1] > 0:
        tmp_Y = Yb - (A_prev @ np.asarray(C_prev, dtype=np.float64))         # :163
    else:
        tmp_Y = Yb.copy()
    ind_patch = np.asarray(ind_patch, dtype=bool).ravel()
    W = sp.csr_matrix(W)
    b0 = np.asarray(b0, dtype=np.float64).ravel()
    if only_rows is not None:
        rows = np.asarray(only_rows)
        Wr = W[rows]
        return (Yb[np.nonzero(ind_patch)[0][rows], :] - Wr @ tmp_Y) - (b0[rows] - Wr @ tmp_Y.mean(axis=1))[:, None]
    return (Yb[ind_patch, :] - W @ tmp_Y) - (b0 - W @ tmp_Y.mean(axis=1))[:, None]   # :166


# --------------------------------------------------------------------------- #
# spatial
# --------------------------------------------------------------------------- #
def _prep_spatial(Y, A, C, active_pixel):
    A = np.array(A.toarray() if sp.issparse(A) else A, dtype=np.float64, copy=True)
    C = np.asarray(C, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if active_pixel is None:
        active = np.ones(A.shape, dtype=bool)
    else:
        active = np.asarray(active_pixel.toarray() if sp.issparse(active_pixel) else active_pixel).astype(bool)
    return Y, A, C, active


def HALS_spatial(Y, A, C, active_pixel=None, maxIter=1):
    """utilities/HALS_spatial.m:1-45."""
    Y, A, C, active = _prep_spatial(Y, A, C, active_pixel)
    A[~active] = 0                                           # :26
    K = A.shape[1]
    Cmean = C.mean(axis=1); Ymean = Y.mean(axis=1); T = C.shape[1]
    U = Y @ C.T - T * np.outer(Ymean, Cmean)                 # :31
    V = C @ C.T - T * np.outer(Cmean, Cmean)                 # :32
    cc = np.diag(V).copy()                                   # :33
    for _ in range(maxIter):                                 # :36
        for k in range(K):
            if cc[k] == 0:
                continue
            ti = active[:, k]
            ak = np.maximum(0, A[ti, k] + (U[ti, k] - A[ti, :] @ V[:, k]) / cc[k])   # :42
            A[ti, k] = ak
    return A


def HALS_spatial_thresh(Y, A, C, active_pixel, maxIter, sn):
    """utilities/HALS_spatial_thresh.m:1-53."""
    Y, A, C, active = _prep_spatial(Y, A, C, active_pixel)
    sn = np.asarray(sn, dtype=np.float64).reshape(-1)        # :27
    A[~active] = 0                                           # :29
    K = A.shape[1]
    Cmean = C.mean(axis=1); Ymean = Y.mean(axis=1); T = C.shape[1]
    U = Y @ C.T - T * np.outer(Ymean, Cmean)                 # :34
    V = C @ C.T - T * np.outer(Cmean, Cmean)                 # :35
    cc = np.diag(V).copy()                                   # :36
    with np.errstate(divide="ignore"):
        cc_thr = 3.0 / np.sqrt(cc)                           # :37
    for _ in range(maxIter):                                 # :40
        for k in range(K):
            if cc[k] == 0:
                continue
            ti = active[:, k]
            if ti.sum() == 0:
                A[:, k] = 0                                  # :46-48
                continue
            ak = A[ti, k] + (U[ti, k] - A[ti, :] @ V[:, k]) / cc[k]          # :50
            ak[ak < sn[ti] * cc_thr[k]] = 0                  # :51
            A[ti, k] = ak
    return A


def nnls(A, b, s=None, tol=1e-9, maxIter=None):
    """endoscope/nnls_spatial.m:41-109 (local function nnls)."""
    A = np.asarray(A, dtype=np.float64); b = np.asarray(b, dtype=np.float64).ravel()
    p = A.shape[1]
    if s is None:
        s = np.zeros(p)
    if maxIter is None:
        maxIter = p
    if (s > 0).sum() > maxIter:
        s = np.zeros(p)
    mu = np.zeros(0)
    for _ in range(maxIter):                                 # :76
        l = b - A @ s                                        # :77
        Pset = s > 0                                         # :78
        if l.size == 0 or l.max() < tol:                     # :80
            break
        temp = int(np.argmax(l))                             # :84
        Pset[temp] = True
        if Pset.sum() > maxIter:                             # :86
            break
        while Pset.any():                                    # :90
            mu = np.linalg.solve(A[np.ix_(Pset, Pset)], b[Pset])            # :93
            if np.all(mu > tol):                             # :98
                break
            sP = s[Pset]
            temp2 = sP / (sP - mu)                           # :102
            temp2 = temp2[~(mu > tol)]                       # :103
            a = temp2.min()                                  # :104
            s[Pset] = sP + a * (mu - sP)                     # :105
            Pset[s < tol] = False                            # :106
        s[Pset] = mu                                         # :109
    return s


def nnls_spatial(Y, A, C, active_pixel=None, maxN=5):
    """endoscope/nnls_spatial.m:1-38."""
    Y = np.asarray(Y, dtype=np.float64); C = np.asarray(C, dtype=np.float64)
    d = Y.shape[0]; K = C.shape[0]
    if active_pixel is None:
        active = np.ones((d, K), dtype=bool)
    else:
        active = np.asarray(active_pixel.toarray() if sp.issparse(active_pixel) else active_pixel).astype(bool)
    Yc = Y - Y.mean(axis=1, keepdims=True)                   # :26-27
    Cc = C - C.mean(axis=1, keepdims=True)                   # :28
    CC = Cc @ Cc.T                                           # :29
    YC = Cc @ Yc.T                                           # :30
    ind_fit = np.nonzero(active.sum(axis=1) > 1e-9)[0]       # :31
    Aout = np.zeros((d, K))                                  # :32
    for m in ind_fit:                                        # :35
        ind = active[m, :]
        Aout[m, ind] = nnls(CC[np.ix_(ind, ind)], YC[ind, m], None, 1e-4, maxN)   # :37
    return Aout


# --------------------------------------------------------------------------- #
# temporal
# --------------------------------------------------------------------------- #
def fast_temporal(Y, A):
    """[aa, C_raw] = fast_temporal(Y, A)  (@Sources2D/update_temporal_parallel.m:314-337): the mean fluorescence
    over the pixels at >= half the footprint's maximum."""
    A = np.asarray(A.todense()) if sp.issparse(A) else np.asarray(A, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tmpA = A * (1.0 / A.max(axis=0, keepdims=True))      # :327
        ind_max = tmpA >= 0.5                                # :328 (NaN compares false)
    tmp_A = A * ind_max                                      # :329
    aa = (tmp_A ** 2).sum(axis=0)                            # :330
    ind = aa == 0                                            # :331
    aa_div = aa.copy(); aa_div[ind] = np.inf                 # :332
    C_raw = (tmp_A.T @ Y) / aa_div[:, None]                  # :333
    aa[ind] = 0                                              # :334
    return aa, C_raw


def HALS_temporal(Y, A, C, maxIter=1, deconv_options=None):
    """utilities/HALS_temporal.m:1-119 -- no-deconvolution branch (:64-68).

    Returns (C, C_raw, cc).  The deconvolution branch (:70-104) needs OASIS and
    MATLAB toolbox semantics (pwelch, fminbnd) that are parity-unpinned; it is
    restated separately in oracle/oasis_oracle.py when present.
    """
    if deconv_options is not None:
        raise NotImplementedError("deconvolution branch not restated in this oracle")
    Y = np.asarray(Y, dtype=np.float64)
    A = np.asarray(A.toarray() if sp.issparse(A) else A, dtype=np.float64)   # :47
    C = np.array(C, dtype=np.float64, copy=True)
    K = A.shape[1]; T = Y.shape[1]
    C_old = C.copy()                                         # :43
    C_raw = np.zeros((K, T))                                 # :45
    U = A.T @ Y                                              # :48
    V = A.T @ A                                              # :49
    aa = np.diag(V).copy()                                   # :50
    ind_update = np.nonzero(aa > 0)[0]                       # :51
    for _ in range(maxIter):                                 # :59
        for k in ind_update:
            ck_raw = C[k, :] + (U[k, :] - V[k, :] @ C) / aa[k]               # :62
            ck_raw = ck_raw - ck_raw.min()                   # :66
            C[k, :] = ck_raw                                 # :67
            C_raw[k, :] = ck_raw                             # :68
    with np.errstate(divide="ignore", invalid="ignore"):
        cc = ((C * C_old).sum(axis=1) - T * C.mean(axis=1) * C_old.mean(axis=1)) / (
            C.std(axis=1, ddof=1) * C_old.std(axis=1, ddof=1) * T)           # :116
    return C, C_raw, cc


# --------------------------------------------------------------------------- #
# search location / post-processing
# --------------------------------------------------------------------------- #
def com(A, d1, d2):
    """utilities/com.m:1-28 (2-D case).  Returns K x 2 (row, col), 1-based coordinates."""
    A = sp.csc_matrix(A)
    x = np.tile(np.arange(1, d1 + 1), d2).astype(np.float64)                 # Coor.x :21
    y = np.repeat(np.arange(1, d2 + 1), d1).astype(np.float64)               # Coor.y :22
    s = np.asarray(A.sum(axis=0)).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        cm = np.vstack([A.T @ x, A.T @ y]).T / s[:, None]    # :24
    cm[cm < 0] = 0                                           # :26
    cm[cm[:, 0] > d1, 0] = d1
    cm[cm[:, 1] > d2, 1] = d2
    return cm


def determine_search_location(A, d1, d2, min_size=3, max_size=8, dist=3):
    """'ellipse' method.  utilities/determine_search_location.m:50-104."""
    A = sp.csc_matrix(A, dtype=np.float64, copy=True).tolil()
    d, nr = A.shape
    ind_empty = np.asarray(A.sum(axis=0)).ravel() == 0       # :52
    if ind_empty.any():
        for k in np.nonzero(ind_empty)[0]:
            A[0, k] = 1                                      # :54
    A = A.tocsc()
    x = np.tile(np.arange(1, d1 + 1), d2).astype(np.float64)
    y = np.repeat(np.arange(1, d2 + 1), d1).astype(np.float64)
    cm = com(A, d1, d2)                                      # :63
    IND = np.zeros((d, nr), dtype=bool)
    for i in range(nr):                                      # :71
        a = np.asarray(A[:, i].todense()).ravel()
        cor = np.stack([x - cm[i, 0], y - cm[i, 1]], axis=1)
        Vr = (cor.T * a) @ cor / a.sum()                     # :73
        D, V = np.linalg.eigh((Vr + Vr.T) / 2)               # :74 (ascending like MATLAB eig of sym.)
        d11 = min(max_size ** 2, max(min_size ** 2, D[0]))   # :81
        d22 = min(max_size ** 2, max(min_size ** 2, D[1]))   # :82
        IND[:, i] = np.sqrt((cor @ V[:, 0]) ** 2 / d11 + (cor @ V[:, 1]) ** 2 / d22) <= dist   # :84
    if ind_empty.any():
        IND[:, ind_empty] = False                            # :103
    return IND


def _imopen_square(img, sz):
    """Grey opening with a flat sz x sz square; MATLAB pads erosion with +inf and dilation with -inf."""
    from scipy.ndimage import minimum_filter, maximum_filter
    er = minimum_filter(img, size=sz, mode="constant", cval=np.inf)
    return maximum_filter(er, size=sz, mode="constant", cval=-np.inf)


def connectivity_constraint(img, thr=0.01, sz=5):
    """endoscope/connectivity_constraint.m:1-18.  img: d1 x d2."""
    from scipy.ndimage import label
    img = np.array(img, dtype=np.float64, copy=True)
    ind_max = int(np.argmax(img.reshape(-1, order="F")))     # :10 first max, column-major
    ai_open = _imopen_square(img, sz)                        # :12-13
    temp = ai_open > img.max() * thr                         # :15
    l, _ = label(temp, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])          # :16 bwlabel(.,4)
    lf = l.reshape(-1, order="F")
    keep = (lf == lf[ind_max]).reshape(img.shape, order="F")
    img[~keep] = 0                                           # :18
    return img


def post_process_spatial(A_img, connected=True, circular=False):
    """@Sources2D/post_process_spatial.m:19-32 (defaults: connected only).

    A_img: d1 x d2 x K.  Returns d x K (column-major flattening).
    """
    d1, d2, K = A_img.shape
    out = np.zeros((d1 * d2, K))
    for m in range(K):
        ai = np.asarray(A_img[:, :, m], dtype=np.float64)
        if connected:
            ai = connectivity_constraint(ai)                 # :24-26
        if circular:
            ai = circular_constraints(ai)                    # :28-30
        out[:, m] = ai.reshape(-1, order="F")
    return out


# --------------------------------------------------------------------------- #
# optional branches: search_method = 'dilate' and spatial_constraints.circular.  Written with explicit shifts and a flood fill (no
# scipy.ndimage), so that the product's scipy.ndimage versions (cnmf_e_amd/hostops.py) are checked against an independent statement of the
# toolbox rules: medfilt2 pads with zeros; imdilate pads with -Inf, imerode with +Inf; strel('disk', R, 0) is the exact disc.
# Toolbox functions are not in the repository: PARITY UNPINNED for them.
# --------------------------------------------------------------------------- #
def _shifted(img, dy, dx, fill):
    """out[y, x] = img[y + dy, x + dx], `fill` outside"""
    out = np.full(img.shape, fill, dtype=img.dtype)
    n0, n1 = img.shape
    ys, ye = max(0, -dy), min(n0, n0 - dy)
    xs, xe = max(0, -dx), min(n1, n1 - dx)
    if ys < ye and xs < xe:
        out[ys:ye, xs:xe] = img[ys + dy:ye + dy, xs + dx:xe + dx]
    return out


def _se_offsets(se):
    se = np.asarray(se, dtype=bool)
    cy, cx = se.shape[0] // 2, se.shape[1] // 2              # strel origin: floor((size + 1) / 2), 1-based
    return [(int(y) - cy, int(x) - cx) for y, x in zip(*np.nonzero(se))]


def _bw_dilate(bw, se):
    out = np.zeros(bw.shape, dtype=bool)
    for dy, dx in _se_offsets(se):
        out |= _shifted(bw, -dy, -dx, False)                 # reflection of the element; symmetric elements only matter here
    return out


def _bw_erode(bw, se):
    out = np.ones(bw.shape, dtype=bool)
    for dy, dx in _se_offsets(se):
        out &= _shifted(bw, dy, dx, True)
    return out


def _medfilt3(img):
    """medfilt2(img) / medfilt2(img, [3 3])"""
    st = np.stack([_shifted(img, dy, dx, 0.0) for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)
    return np.sort(st, axis=0)[4]


def _flood_label(bw, conn):
    """bwlabel(bw, 4) / bwlabeln(bw, 8): labels in column-major scan order of the first pixel of each component"""
    n0, n1 = bw.shape
    lab = np.zeros(bw.shape, dtype=np.int64)
    nb = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if conn == 8 else [])
    cur = 0
    for x in range(n1):
        for y in range(n0):
            if bw[y, x] and not lab[y, x]:
                cur += 1
                lab[y, x] = cur
                stack = [(y, x)]
                while stack:
                    a, b = stack.pop()
                    for dy, dx in nb:
                        u, w = a + dy, b + dx
                        if 0 <= u < n0 and 0 <= w < n1 and bw[u, w] and not lab[u, w]:
                            lab[u, w] = cur
                            stack.append((u, w))
    return lab, cur


def strel_disk(radius):
    r = int(radius)
    return np.array([[(x * x + y * y) <= radius * radius for x in range(-r, r + 1)] for y in range(-r, r + 1)])


def threshold_components(A, d1, d2, nb=1, nrgthr=0.99):
    """utilities/threshold_components.m:20-62 with medw = [3 3], clos_op = strel('square', 3) (CNMFSetParms.m defaults)."""
    A = np.asarray(A.toarray() if sp.issparse(A) else A, dtype=np.float64)
    d, nr = A.shape
    Ath = np.zeros((d, nr))
    Ath[:, nr - nb:] = A[:, nr - nb:]                        # :22
    sq = np.ones((3, 3), dtype=bool)
    for i in range(nr - nb):                                 # :25
        A_temp = _medfilt3(A[:, i].reshape(d1, d2, order="F")).reshape(-1, order="F")     # :26-30
        e = A_temp ** 2
        ind = np.argsort(e, kind="stable")                   # :31
        temp = np.cumsum(e[ind])                             # :32
        ff = np.nonzero(temp > (1 - nrgthr) * temp[-1])[0]   # :33
        BW = np.zeros(d, dtype=bool)
        if ff.size:
            BW[ind[ff[0]:]] = True                           # :35
        BW = _bw_erode(_bw_dilate(BW.reshape(d1, d2, order="F"), sq), sq)                 # :37 imclose
        L, NUM = _flood_label(BW, 8)                         # :39
        if NUM > 0:
            nrg = [float((A_temp[(L == l).reshape(-1, order="F")] ** 2).sum()) for l in range(1, NUM + 1)]   # :42-45
            sel = (L == 1 + int(np.argmax(nrg))).reshape(-1, order="F")                   # :46-47
            Ath[sel, i] = A_temp[sel]                        # :50,56
    return Ath


def determine_search_location_dilate(A, d1, d2, se, nb=1, nrgthr=0.99):
    """'dilate' method.  utilities/determine_search_location.m:51-56,89-98."""
    A = np.array(A.toarray() if sp.issparse(A) else A, dtype=np.float64)
    d, nr = A.shape
    ind_empty = A.sum(axis=0) == 0                           # :52
    A[0, ind_empty] = 1                                      # :54
    Ath = threshold_components(A, d1, d2, nb, nrgthr)        # :90
    IND = np.zeros((d, nr), dtype=bool)
    for i in range(nr):                                      # :91
        img = Ath[:, i].reshape(d1, d2, order="F")
        # grey dilation by a flat element (max over the neighbourhood, -Inf outside), then > 0
        st = np.stack([_shifted(img, -dy, -dx, -np.inf) for dy, dx in _se_offsets(se)], axis=0)
        IND[:, i] = (st.max(axis=0) > 0).reshape(-1, order="F")   # :92-93
    IND[:, ind_empty] = False                                # :98
    return IND


def circular_constraints(img):
    """endoscope/circular_constraints.m:8-55."""
    img = np.array(img, dtype=np.float64, copy=True)
    tmp1, tmp2 = np.nonzero(img)                             # :8
    if tmp1.size == 0:
        return img                                           # :9-11
    rmin, rmax, cmin, cmax = tmp1.min(), tmp1.max(), tmp2.min(), tmp2.max()
    nr, nc = img.shape
    if rmax - rmin < 1 or cmax - cmin < 1:                   # :17-19
        return img
    if not (rmin == 0 and rmax == nr - 1 and cmin == 0 and cmax == nc - 1):   # :21 / :52-54
        img[rmin:rmax + 1, cmin:cmax + 1] = circular_constraints(img[rmin:rmax + 1, cmin:cmax + 1])
        return img
    flat = img.reshape(-1, order="F")
    ind_max = int(np.argmax(flat)); vmax = flat[ind_max]     # :30
    y0, x0 = ind_max % nr + 1, ind_max // nr + 1             # :31
    x, y = np.meshgrid(np.arange(1, nc + 1), np.arange(1, nr + 1))           # :32
    fx = np.empty_like(img); fy = np.empty_like(img)         # :33 [fx, fy] = gradient(img)
    fx[:, 1:-1] = (img[:, 2:] - img[:, :-2]) / 2; fx[:, 0] = img[:, 1] - img[:, 0]; fx[:, -1] = img[:, -1] - img[:, -2]
    fy[1:-1, :] = (img[2:, :] - img[:-2, :]) / 2; fy[0, :] = img[1, :] - img[0, :]; fy[-1, :] = img[-1, :] - img[-2, :]
    ind = ((fx * (x0 - x) + fy * (y0 - y)) < 0) & (img < vmax / 3)           # :34
    img[ind] = 0                                             # :35
    l, _ = _flood_label(img != 0, 4)                         # :39
    keep = _bw_dilate(l == l[y0 - 1, x0 - 1], np.ones((3, 3), dtype=bool))   # :40
    img[~keep] = 0                                           # :41
    return _medfilt3(img)                                    # :42


# --------------------------------------------------------------------------- #
# method level: the three Sources2D update methods on a patched FOV
# --------------------------------------------------------------------------- #
class OracleSources2D:
    """Float64 restatement of the three @Sources2D update methods (ring model, bg_ssub=1).

    State mirrors Sources2D.m:10-57: A (d x K), C, C_raw (K x T), W{.}, b0{.}, b0_new,
    A_prev, C_prev, P.sn, P.Ymean.  ``Yfull`` is the whole video as d1 x d2 x T;
    blocks are cut from it the way get_patch_data.m:50-93 returns them.
    """

    def __init__(self, Yfull, d1, d2, T, patch_dims, ring_radius, A, C, sn, *,
                 spatial_algorithm="hals", maxIter=5, num_neighbors=None,
                 min_size=3, max_size=8, dist=3, bg_acceleration=True, bg_ssub=1, deconv_options=None, thresh_outlier=np.nan,
                 search_method="ellipse", se="default", bSiz=3, nb=1, nrgthr=0.99, connected=True, circular=False):
        self.bg_ssub = int(bg_ssub)
        # options of the 'dilate' search (CNMFSetParms.m: se = strel('disk', 4, 0), bSiz = 3, nb = 1, nrgthr = 0.99) and of post_process_spatial
        self.search_method = search_method
        self.se = strel_disk(4) if isinstance(se, str) else se
        self.bSiz, self.nb, self.nrgthr, self.connected, self.circular = bSiz, nb, nrgthr, connected, circular
        self.thresh_outlier = thresh_outlier
        self.deconv_options = deconv_options       # None: options.deconv_flag = false; a dict: the keyword arguments of oasis_oracle
        self.Y = Yfull
        self.d1, self.d2, self.T = d1, d2, T
        self.ring_radius = ring_radius
        self.patch_pos, self.block_pos = distribute_geometry(d1, d2, patch_dims, ring_radius)  # Sources2D.m:236
        self.A = sp.csc_matrix(A, dtype=np.float64)
        self.C = np.array(C, dtype=np.float64)
        self.C_raw = self.C.copy()
        self.A_prev = self.A.copy()
        self.C_prev = self.C.copy()
        self.sn = np.asarray(sn, dtype=np.float64).reshape(d1, d2, order="F")
        self.spatial_algorithm = spatial_algorithm
        self.maxIter = maxIter
        self.search = dict(min_size=min_size, max_size=max_size, dist=dist)
        self.bg_acceleration = bg_acceleration
        rr = int(np.ceil(ring_radius / self.bg_ssub))        # initComponents_parallel.m:214
        r_shift, c_shift = get_nhood(rr, num_neighbors)
        self.W, self.b0, self.Ymean = {}, {}, {}
        for idx in np.ndindex(self.patch_pos.shape):
            p, b = self.patch_pos[idx], self.block_pos[idx]
            if self.bg_ssub == 1:
                self.W[idx] = build_ring_W(p, b, d1, d2, r_shift, c_shift)
            else:
                self.W[idx] = build_ring_W_ssub(int(b[1] - b[0] + 1), int(b[3] - b[2] + 1), self.bg_ssub, r_shift, c_shift)
            self.b0[idx] = np.zeros((p[1] - p[0] + 1) * (p[3] - p[2] + 1))   # initComponents_parallel.m:221
            self.Ymean[idx] = self._block(p).astype(np.float64).mean(axis=1)  # P.Ymean (:338-339)
        self.b0_new = None

    def estimate_noise(self, frame_range=None):
        """sn = estimate_noise(obj, frame_range, 'psd')  (@Sources2D/Sources2D.m:328-379), literally: GetSn on every storage block of
        distribute_data.m:91-97 (neighbouring blocks share their cut line), row / column end-1 of every block but the last removed, cell2mat."""
        from oasis_oracle import GetSn
        T = self.T
        f0, f1 = (1, min(T, 3000)) if frame_range is None else frame_range                     # :333-335
        nr, nc = self.patch_pos.shape
        w = self.ring_radius
        pr = np.array([int(self.patch_pos[m, 0][0]) for m in range(nr)] + [self.d1]); pc = np.array([int(self.patch_pos[0, j][2]) for j in range(nc)] + [self.d2])
        bir = np.unique(np.clip(np.concatenate([pr - 1 - w, pr + w]), 1, self.d1))             # distribute_data.m:91-97
        bic = np.unique(np.clip(np.concatenate([pc - 1 - w, pc + w]), 1, self.d2))
        nrb, ncb = bir.size - 1, bic.size - 1
        rows = []
        for m in range(nrb):
            cols = []
            for n in range(ncb):
                r0, r1, c0, c1 = bir[m], bir[m + 1], bic[n], bic[n + 1]                        # :362-365
                Yp = np.asarray(self.Y[r0 - 1:r1, c0 - 1:c1, f0 - 1:f1], dtype=np.float64)      # :366-367
                tmp = np.array([[GetSn(Yp[i, j]) for j in range(Yp.shape[1])] for i in range(Yp.shape[0])])   # :368
                if m != nrb - 1:
                    tmp = np.delete(tmp, tmp.shape[0] - 2, axis=0)                              # :369-371
                if n != ncb - 1:
                    tmp = np.delete(tmp, tmp.shape[1] - 2, axis=1)                              # :372-374
                cols.append(tmp)
            rows.append(np.hstack(cols))
        sn = np.vstack(rows)                                                                    # :378
        assert sn.shape == (self.d1, self.d2)
        return sn

    # -- helpers ------------------------------------------------------------
    def _block(self, pos):
        r0, r1, c0, c1 = [int(v) for v in pos]
        blk = self.Y[r0 - 1:r1, c0 - 1:c1, :]
        return blk.reshape(-1, self.T, order="F")

    def _mask(self, pos):
        r0, r1, c0, c1 = [int(v) for v in pos]
        m = np.zeros((self.d1, self.d2), dtype=bool)
        m[r0 - 1:r1, c0 - 1:c1] = True
        return m.reshape(-1, order="F")

    def _residual(self, Yb, A_prev_b, C_prev_b, idx, ip, b):
        if self.bg_ssub == 1:
            return residual_ysig(Yb, A_prev_b, C_prev_b, self.W[idx], self.b0[idx], ip)
        return residual_ysig_ssub(Yb, A_prev_b, C_prev_b, self.W[idx], self.b0[idx], ip,
                                  int(b[1] - b[0] + 1), int(b[3] - b[2] + 1), self.bg_ssub)

    def _ring_background(self, idx, R, b, ip):
        """W*R on the patch rows; with bg_ssub > 1 through imresize 'nearest' both ways (Sources2D.m:1325-1334 == :1479-1486)"""
        if self.bg_ssub == 1:
            return self.W[idx] @ R
        nr_b, nc_b = int(b[1] - b[0] + 1), int(b[3] - b[2] + 1)
        T = R.shape[1]
        temp = imresize_scale(R.reshape(nr_b, nc_b, T, order="F"), 1.0 / self.bg_ssub, "nearest")     # :1327-1328
        d1s, d2s = temp.shape[:2]                                                                       # :1326
        Bf = (self.W[idx] @ temp.reshape(-1, T, order="F")).reshape(d1s, d2s, T, order="F")             # :1329
        Bf = imresize_size(Bf, [nr_b, nc_b], "nearest")                                                 # :1330
        return Bf.reshape(-1, T, order="F")[ip]                                                         # :1331-1332

    def _patches(self):
        # MATLAB linear order over the nr_patch x nc_patch cell: rows fastest
        nr, nc = self.patch_pos.shape
        return [(m, n) for n in range(nc) for m in range(nr)]

    def reconstruct_b0(self):
        out = np.zeros((self.d1, self.d2))
        for idx in self._patches():
            r0, r1, c0, c1 = [int(v) for v in self.patch_pos[idx]]
            out[r0 - 1:r1, c0 - 1:c1] = self.b0[idx].reshape(r1 - r0 + 1, c1 - c0 + 1, order="F")
        return out

    def _ymean_full(self):
        out = np.zeros((self.d1, self.d2))
        for idx in self._patches():
            r0, r1, c0, c1 = [int(v) for v in self.patch_pos[idx]]
            out[r0 - 1:r1, c0 - 1:c1] = self.Ymean[idx].reshape(r1 - r0 + 1, c1 - c0 + 1, order="F")
        return out

    # -- background -----------------------------------------------------------
    def update_background_parallel(self):
        """@Sources2D/update_background_parallel.m:121-146,176-230,311-317."""
        first = self._patches()[0]
        row0 = sp.csr_matrix(self.W[first]).getrow(0)
        vals = set(np.unique(row0.data).tolist())
        if row0.nnz < row0.shape[1]:
            vals.add(0.0)
        flag_first = len(vals) == 2                          # :143
        for idx in self._patches():
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mask = self._mask(b)
            ind = np.asarray((sp.csr_matrix(mask.astype(np.float64)) @ self.A).todense()).ravel() > 0   # :128
            A_block = self.A[mask, :][:, ind]                # :129
            C_block = self.C[ind, :]                         # :130
            if A_block.shape[1] == 0 and not flag_first:     # :188
                continue
            ip = ind_patch_mask(p, b)                        # :203-204
            Yb = self._block(b)                              # :208
            sn_patch = self.sn.reshape(-1, order="F")[mask][ip]
            if self.bg_ssub == 1:
                self.W[idx], self.b0[idx] = fit_ring_model(Yb, A_block, C_block, self.W[idx], self.thresh_outlier,
                                                           sn_patch, ip, self.bg_acceleration,
                                                           only_rows=getattr(self, "bg_only_rows", None))   # :218 (bg_only_rows: timing harness only, bench.py --cpu-baseline full)
            else:                                            # :219-230
                nr_b, nc_b = int(b[1] - b[0] + 1), int(b[3] - b[2] + 1)
                temp = np.asarray(Yb, dtype=np.float64) - (A_block @ C_block if A_block.shape[1] else 0.0)      # :221
                self.b0[idx] = temp.mean(axis=1)[ip]                                                            # :222-223
                low = imresize_scale(temp.reshape(nr_b, nc_b, self.T, order="F"), 1.0 / self.bg_ssub, "nearest")  # :224
                low = low.reshape(-1, self.T, order="F")
                sn_low = imresize_scale(self.sn.reshape(-1, order="F")[mask].reshape(nr_b, nc_b, order="F"), 1.0 / self.bg_ssub,
                                        "nearest") * self.bg_ssub                                              # :137
                self.W[idx], _ = fit_ring_model(low, None, None, self.W[idx], self.thresh_outlier, sn_low.reshape(-1, order="F"), None,
                                                self.bg_acceleration)                                          # :227
        self.b0_new = self.reconstruct_b0()                  # :315
        self.A_prev = self.A.copy()                          # :316
        self.C_prev = self.C.copy()                          # :317

    # -- objective -------------------------------------------------------------------
    def compute_RSS(self):
        """[RSS_total, RSS] = compute_RSS(obj)  (@Sources2D/Sources2D.m:1358-1510), ring model, bg_ssub = 1, all frames:
        per patch  RSS = sum((Y(patch) - A*C - (W*(Y_block - b0_block - A_prev*C_prev) + b0_new(patch))).^2)."""
        b0_ = self.reconstruct_b0()                                          # :1398
        b0_new_ = np.asarray(self.b0_new, dtype=np.float64).reshape(self.d1, self.d2)   # :1399
        RSS = {}
        for idx in self._patches():
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mask = self._mask(b)                                             # :1421-1422
            ind = np.asarray((sp.csr_matrix(mask.astype(np.float64)) @ self.A).todense()).ravel() > 0        # :1423
            A_b = self.A[mask, :][:, ind]; C_b = self.C[ind, :]               # :1424-1425
            indp = np.asarray((sp.csr_matrix(mask.astype(np.float64)) @ self.A_prev).todense()).ravel() > 0  # :1427
            A_pb = self.A_prev[mask, :][:, indp]; C_pb = self.C_prev[indp, :]  # :1428-1429
            ip = ind_patch_mask(p, b)                                        # :1440-1443
            Yb = np.asarray(self._block(b), dtype=np.float64)                 # :1447-1449
            r0, r1, c0, c1 = [int(v) for v in b]
            b0_ring = b0_[r0 - 1:r1, c0 - 1:c1].reshape(-1, order="F")        # :1453,1455
            b0_new_patch = b0_new_[r0 - 1:r1, c0 - 1:c1].reshape(-1, order="F")[ip]   # :1454,1468
            YmAC = Yb[ip, :] - (A_b[ip, :] @ C_b if A_b.shape[1] else 0.0)    # :1467
            R = Yb - b0_ring[:, None]                                         # :1471
            if A_pb.shape[1]:
                R = R - A_pb @ C_pb
            Bf = self._ring_background(idx, R, b, ip)                         # :1475 / :1479-1486
            Ybg = Bf + b0_new_patch[:, None]                                  # :1487
            RSS[idx] = float(np.sum((YmAC - Ybg) ** 2))                       # :1502
        total = float(sum(RSS.values()))                                      # :1507-1508
        self.RSS = total
        return total, RSS

    def reconstruct_background(self):
        """Ybg = reconstruct_background(obj)  (@Sources2D/Sources2D.m:1247-1355), ring model, bg_ssub = 1, all frames:
        per patch  Ybg = W*(Y_block - b0_block - A_prev*C_prev) + b0_new(patch)  ->  d1 x d2 x T."""
        b0_ = self.reconstruct_b0()                                          # :1292
        b0_new_ = np.asarray(self.b0_new, dtype=np.float64).reshape(self.d1, self.d2)   # :1293
        Ybg = np.zeros((self.d1, self.d2, self.T))                           # :1297
        for idx in self._patches():
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mask = self._mask(b)                                             # :1315-1316
            indp = np.asarray((sp.csr_matrix(mask.astype(np.float64)) @ self.A_prev).todense()).ravel() > 0  # :1317
            A_pb = self.A_prev[mask, :][:, indp]; C_pb = self.C_prev[indp, :]  # :1319-1320
            Yb = np.asarray(self._block(b), dtype=np.float64)                 # :1303-1305
            r0, r1, c0, c1 = [int(v) for v in b]
            b0_ring = b0_[r0 - 1:r1, c0 - 1:c1].reshape(-1, order="F")        # :1308-1309
            R = Yb - b0_ring[:, None]                                         # :1324
            if A_pb.shape[1]:
                R = R - A_pb @ C_pb
            Bf = self._ring_background(idx, R, b, ind_patch_mask(p, b))        # :1324 / :1325-1334
            q0, q1, s0, s1 = [int(v) for v in p]
            b0_patch = b0_new_[q0 - 1:q1, s0 - 1:s1].reshape(-1, order="F")   # :1311
            Ybg[q0 - 1:q1, s0 - 1:s1, :] = (Bf + b0_patch[:, None]).reshape(q1 - q0 + 1, s1 - s0 + 1, self.T, order="F")   # :1330
        return Ybg

    # -- spatial -----------------------------------------------------------------
    def update_spatial_parallel(self, update_sn=False):
        """@Sources2D/update_spatial_parallel.m:61-100,116-216,320-351."""
        d1, d2 = self.d1, self.d2
        if self.search_method == "dilate":
            # :56 copies the options struct BEFORE :63-65 clears obj.options.se, so this call still sees the old element; the cleared one
            # (-> strel('disk', bSiz, 0), determine_search_location.m:42-44) is what the NEXT call sees
            se = self.se
            self.se = None                                                   # :64
            IND = determine_search_location_dilate(self.A, d1, d2, strel_disk(self.bSiz) if se is None else se, self.nb, self.nrgthr)
        else:
            IND = determine_search_location(self.A, d1, d2, **self.search)     # :66
        K = self.A.shape[1]
        A_ = np.zeros((d1 * d2, K))
        Aprev_dense_any = self.A_prev
        sn_new = np.array(self.sn, dtype=np.float64).reshape(-1, order="F").copy() if update_sn else None   # :101-102
        for idx in self._patches():
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mb, mp = self._mask(b), self._mask(p)
            halo = mb & ~mp                                  # mask==1 after patch overwritten with 2 (:84-85)
            ind = np.nonzero(IND[mp, :].any(axis=0))[0]      # :87
            if ind.size == 0 and not update_sn:
                continue                                     # :121-124
            A_patch = self.A[mb, :][:, ind]                  # :88
            IND_patch = IND[mp, :][:, ind]                   # :89
            sn_patch = self.sn.reshape(-1, order="F")[mp]    # :90
            C_patch = self.C[ind, :]                         # :91
            indp = np.nonzero(np.asarray(Aprev_dense_any[halo, :].sum(axis=0)).ravel() > 0)[0]   # :96 (halo only!)
            A_prev_b = self.A_prev[mb, :][:, indp]           # :97
            C_prev_b = self.C_prev[indp, :]                  # :98
            ip = ind_patch_mask(p, b)
            Yb = self._block(b)                              # :147
            Ysig = self._residual(Yb, A_prev_b, C_prev_b, idx, ip, b)                     # :162-178
            if update_sn:                                    # :191-194  sn_patch = GetSn(Ypatch)
                from oasis_oracle import GetSn
                sn_patch = np.array([GetSn(row) for row in Ysig])
                sn_new[mp] = sn_patch
            if ind.size == 0:
                continue                                     # :196-199
            A_pp = A_patch[ip, :]                            # :199
            if self.spatial_algorithm == "hals":
                temp = HALS_spatial(Ysig, A_pp, C_patch, IND_patch, 3)      # :203
            elif self.spatial_algorithm == "hals_thresh":
                temp = HALS_spatial_thresh(Ysig, A_pp, C_patch, IND_patch, 3, sn_patch)   # :205
            else:
                temp = nnls_spatial(Ysig, A_pp, C_patch, IND_patch, 20)     # :211
            rows = np.nonzero(mp)[0]
            for j, k in enumerate(ind):                      # :324-334 (write, not accumulate)
                A_[rows, k] = temp[:, j]
        self.A_raw = A_.copy()
        if update_sn:
            self.sn = sn_new.reshape(np.shape(self.sn), order="F") if np.ndim(self.sn) == 2 else sn_new   # :336-337
        A_img = A_.reshape(d1, d2, K, order="F")
        self.A = sp.csc_matrix(post_process_spatial(A_img, self.connected, self.circular))  # :341
        self.b0_new = self._ymean_full() - np.asarray(
            self.A @ self.C.mean(axis=1)).reshape(d1, d2, order="F")        # :349

    # -- temporal ----------------------------------------------------------------
    def init_residual(self, idx):
        """@Sources2D/initComponents_residual_parallel.m:106-121,186-217 (ring model): the video greedyROI_endoscope searches for missed
        neurons in one patch -- the block's neurons subtracted, then the ring background: Ypatch = Y(ind_patch,:) - A*C - W*(Y - A*C) -
        (b0 - W*mean(Y - A*C, 2)) (bg_ssub = 1, :206), the imresize form for bg_ssub > 1 (:209-217).  Returns d_patch x T (float64)."""
        p, b = self.patch_pos[idx], self.block_pos[idx]
        mb = self._mask(b)
        ind = np.nonzero(np.asarray(self.A[mb, :].sum(axis=0)).ravel() > 0)[0]             # :116
        A_b = self.A[mb, :][:, ind]                                                         # :117
        C_b = self.C[ind, :]                                                                # :120
        ip = ind_patch_mask(p, b)
        Yb = self._block(b)
        # :199 subtracts A*C from the whole block first; _residual's Y(ind_patch,:) - W*(Y - A*C) - ... keeps the neurons in its first term
        Ysig = self._residual(Yb, A_b, C_b, idx, ip, b)
        if ind.size:
            Ysig = Ysig - np.asarray(A_b[ip, :] @ C_b)
        return Ysig

    def update_temporal_parallel(self, use_c_hat=True):
        """@Sources2D/update_temporal_parallel.m:62-94,112-186,264-295 (no deconv)."""
        K, T = self.C.shape
        C_new = np.zeros((K, T)); aa = np.zeros(K)
        for idx in self._patches():
            p, b = self.patch_pos[idx], self.block_pos[idx]
            mb = self._mask(b)
            ind = np.nonzero(np.asarray(self.A[mb, :].sum(axis=0)).ravel() > 0)[0]          # :83
            if ind.size == 0:
                continue                                     # :123
            A_b = self.A[mb, :][:, ind]                      # :84
            C_b = self.C[ind, :]                             # :86
            indp = np.nonzero(np.asarray(self.A_prev[mb, :].sum(axis=0)).ravel() > 0)[0]    # :90
            A_prev_b = self.A_prev[mb, :][:, indp]
            C_prev_b = self.C_prev[indp, :]
            ip = ind_patch_mask(p, b)
            Yb = self._block(b)
            Ysig = self._residual(Yb, A_prev_b, C_prev_b, idx, ip, b)                      # :149-165
            A_pp = A_b[ip, :]
            if not use_c_hat:
                aa_p, C_raw_p = fast_temporal(Ysig, A_pp)                                   # :174-175
            elif self.deconv_options is not None:                                          # :106-110 (deconv_flag = true)
                from oasis_oracle import HALS_temporal_deconv
                _, C_raw_p, _, _, _ = HALS_temporal_deconv(Ysig, A_pp, C_b, self.maxIter, **self.deconv_options)
                aa_p = np.asarray(A_pp.multiply(A_pp).sum(axis=0)).ravel()
            else:
                _, C_raw_p, _ = HALS_temporal(Ysig, A_pp, C_b, self.maxIter, None)         # :180
                aa_p = np.asarray(A_pp.multiply(A_pp).sum(axis=0)).ravel()                  # :181
            for j, k in enumerate(ind):                      # :269-278
                C_new[k, :] += C_raw_p[j, :] * aa_p[j]
                aa[k] += aa_p[j]
        aa[aa == 0] = 1                                      # :279
        self.C_raw = C_new / aa[:, None]                     # :280
        if self.deconv_options is not None:                  # :282-283  obj.C = obj.deconvTemporal()
            from oasis_oracle import deconvTemporal
            self.C, self.C_raw, self.S, self.kernel_pars, self.neuron_sn = deconvTemporal(self.C_raw, **self.deconv_options)
        else:
            self.C_raw = self.C_raw - self.C_raw.min(axis=1, keepdims=True)     # :285
            self.C = self.C_raw.copy()                       # :286
        self.b0_new = self._ymean_full() - np.asarray(
            self.A @ self.C.mean(axis=1)).reshape(self.d1, self.d2, order="F")              # :293
