"""Host-side image operations on single footprints: the optional branches around the spatial update.

  search_method = 'dilate'        utilities/determine_search_location.m:89-94  (threshold_components.m:1-63 + imdilate)
  spatial_constraints.circular    endoscope/circular_constraints.m:1-55

Both are off in every demo (CNMFSetParms.m: search_method 'ellipse', circular false) and work on one d1 x d2 image per neuron, so they
stay on the host next to determine_search_location's ellipse branch ("host-side prerequisite; tiny").  The Image
Processing Toolbox calls are written with scipy.ndimage under the toolbox's documented border rules: medfilt2 pads with zeros, imdilate
with -Inf, imerode with +Inf; bwlabel(., 4) / bwlabeln(., 8) are 4- / 8-connected; strel('disk', R, 0) is the exact disc x^2 + y^2 <= R^2.
"""
from __future__ import annotations

import numpy as np
import scipy.ndimage as ndi
import scipy.sparse as sp

SQUARE3 = np.ones((3, 3), dtype=bool)
CROSS4 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)


def strel_disk(radius):
    """strel('disk', R, 0): every pixel whose centre is no further than R from the origin"""
    r = int(radius)
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= radius * radius


def medfilt2(img, size=(3, 3)):
    """medfilt2(img, [m n]): median of the m x n neighbourhood, the image padded with zeros"""
    return ndi.median_filter(img, size=tuple(int(s) for s in size), mode="constant", cval=0.0)


def imclose(bw, se):
    """imclose = imerode(imdilate(bw, se), se): the dilation sees -Inf (false) outside the image, the erosion +Inf (true)"""
    return ndi.binary_erosion(ndi.binary_dilation(bw, structure=se, border_value=0), structure=se, border_value=1)


def threshold_components(A, d1, d2, nb=1, nrgthr=0.99, medw=(3, 3), clos_op=None):
    """Ath = threshold_components(A, options) (threshold_components.m:20-62): per component median filter, keep the pixels holding `nrgthr`
    of the energy, close, keep the 8-connected component with the most energy (values of the FILTERED image).  The last `nb` columns are
    copied unchanged (:22,25)."""
    A = sp.csc_matrix(A)
    d, nr = A.shape
    clos_op = SQUARE3 if clos_op is None else np.asarray(clos_op, dtype=bool)
    rows, cols, vals = [], [], []
    for i in range(nr):
        col = A.getcol(i)
        if i >= nr - nb:                                                   # :22
            rows.append(col.indices); cols.append(np.full(col.nnz, i)); vals.append(col.data.astype(np.float64))
            continue
        img = medfilt2(np.asarray(col.todense(), dtype=np.float64).reshape(d1, d2, order="F"), medw)   # :28
        a = img.reshape(-1, order="F")
        e = a * a
        order = np.argsort(e, kind="stable")                               # :31 (sort is stable)
        cs = np.cumsum(e[order])                                           # :32
        above = np.nonzero(cs > (1.0 - nrgthr) * cs[-1])[0]                # :33
        bw = np.zeros(d, dtype=bool)
        if above.size:
            bw[order[above[0]:]] = True                                    # :35
        bw = imclose(bw.reshape(d1, d2, order="F"), clos_op)               # :37
        lab, num = ndi.label(bw, structure=SQUARE3)                        # :39 (8-connected)
        if num == 0:
            continue                                                       # :52 (nothing is written)
        nrg = ndi.sum(img * img, lab, index=np.arange(1, num + 1))         # :43-45
        keep = np.nonzero((lab == 1 + int(np.argmax(nrg))).reshape(-1, order="F"))[0]   # :46-47
        rows.append(keep); cols.append(np.full(keep.size, i)); vals.append(a[keep])      # :49-50,55-56
    if not rows:
        return sp.csc_matrix((d, nr), dtype=np.float64)
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(d, nr))


def search_location_dilate(A, d1, d2, se, nb=1, nrgthr=0.99, medw=(3, 3), clos_op=None):
    """IND = determine_search_location(A, 'dilate', params) (determine_search_location.m:51-56,89-98): threshold, then grow every
    footprint by the structuring element `se`."""
    A = sp.csc_matrix(A, dtype=np.float64).copy()
    d, nr = A.shape
    ind_empty = np.asarray(A.sum(axis=0)).ravel() == 0                     # :52
    if ind_empty.any():                                                    # :53-55
        A = A.tolil(); A[0, np.nonzero(ind_empty)[0]] = 1.0; A = A.tocsc()
    Ath = threshold_components(A, d1, d2, nb, nrgthr, medw, clos_op)       # :90
    se = np.asarray(se, dtype=bool)
    rows, cols = [], []
    for i in range(nr):
        if ind_empty[i]:                                                   # :97-98
            continue
        img = np.asarray(Ath.getcol(i).todense()).reshape(d1, d2, order="F") > 0
        # imdilate of a non-negative image by a flat element is > 0 exactly where the dilated support is (:92-93)
        keep = np.nonzero(ndi.binary_dilation(img, structure=se, border_value=0).reshape(-1, order="F"))[0]
        rows.append(keep); cols.append(np.full(keep.size, i))
    if not rows:
        return sp.csc_matrix((d, nr), dtype=bool)
    r, c = np.concatenate(rows), np.concatenate(cols)
    return sp.csc_matrix((np.ones(r.size, dtype=bool), (r, c)), shape=(d, nr))


def circular_constraints(img):
    """img = circular_constraints(img) (circular_constraints.m:8-55): inside the bounding box of the non-zeros, zero the pixels below a
    third of the peak whose gradient points away from the peak, keep the 4-connected component of the peak grown by one pixel, and median
    filter."""
    img = np.array(img, dtype=np.float64)
    r, c = np.nonzero(img)
    if r.size == 0:                                                        # :9-11
        return img
    rmin, rmax, cmin, cmax = r.min(), r.max(), c.min(), c.max()
    if rmax - rmin < 1 or cmax - cmin < 1:                                 # :17-19
        return img
    sub = img[rmin:rmax + 1, cmin:cmax + 1].copy()                         # :53 (the recursive call sees the cropped image)
    nr, nc = sub.shape
    ind_max = int(np.argmax(sub.reshape(-1, order="F")))                   # :30 (first maximum, column-major)
    vmax = sub.reshape(-1, order="F")[ind_max]
    y0, x0 = ind_max % nr, ind_max // nr                                   # :31 (0-based; only differences are used)
    y, x = np.mgrid[:nr, :nc]                                              # :32
    fy, fx = np.gradient(sub)                                              # :33 gradient(): central differences, one-sided at the edges
    ind = ((fx * (x0 - x) + fy * (y0 - y)) < 0) & (sub < vmax / 3.0)       # :34
    sub[ind] = 0                                                           # :35
    lab, _ = ndi.label(sub != 0, structure=CROSS4)                         # :39
    keep = ndi.binary_dilation(lab == lab[y0, x0], structure=SQUARE3, border_value=0)   # :40
    sub[~keep] = 0                                                         # :41
    img[rmin:rmax + 1, cmin:cmax + 1] = medfilt2(sub)                      # :42,54
    return img


def circular_constraints_columns(A, d1, d2):
    """post_process_spatial.m:29-31 for every column of a sparse d x K matrix"""
    A = sp.csc_matrix(A)
    rows, cols, vals = [], [], []
    for k in range(A.shape[1]):
        col = A.getcol(k)
        if col.nnz == 0:
            continue
        out = circular_constraints(np.asarray(col.todense()).reshape(d1, d2, order="F")).reshape(-1, order="F")
        nz = np.nonzero(out)[0]
        rows.append(nz); cols.append(np.full(nz.size, k)); vals.append(out[nz].astype(A.dtype))
    if not rows:
        return sp.csc_matrix(A.shape, dtype=A.dtype)
    out = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=A.shape)
    out.sort_indices()
    return out
