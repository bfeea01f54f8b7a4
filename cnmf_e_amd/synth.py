"""Seedable synthetic 1-photon video generator.

Content: isotropic Gaussian footprints (sigma=3 px, truncated at gSiz x gSiz,
``demos/demo_large_data_1p.m:16-17``), AR(1) traces driven by Bernoulli spikes
(same family as ``OASIS_matlab/functions/gen_data.m:31-41``), a smooth
fluctuating background that a ring model can fit, unit Gaussian noise.

All *factors* (A, C, background field and time course) come from
``numpy.random.default_rng(seed)`` and are identical on every box.  The dense
video is either synthesised on the host (``make_video``) or directly in HBM
(``make_video_device``, torch generator seeded with ``seed``), so that the
512x512x10000 headline volume (10.5 GB) never has to exist in host RAM.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

__all__ = ["SynthFactors", "make_factors", "make_video", "make_video_device"]


@dataclass
class SynthFactors:
    d1: int
    d2: int
    T: int
    K: int
    A_true: sp.csc_matrix        # d x K
    C_true: np.ndarray           # K x T  float32
    A_init: sp.csc_matrix        # perturbed start (blurred)
    C_init: np.ndarray           # perturbed start
    bg_field: np.ndarray         # d  (column-major pixels) float32
    bg_time: np.ndarray          # T  float32   (1 + 0.1*slow AR(1))
    bg_const: float
    sn: np.ndarray               # d  noise std per pixel (== 1)
    seed: int

    @property
    def d(self):
        return self.d1 * self.d2


def _gauss_patch(sig, siz):
    h = siz // 2
    x = np.arange(-h, h + 1)
    g = np.exp(-(x[:, None] ** 2 + x[None, :] ** 2) / (2.0 * sig * sig))
    return g


def _place_centres(rng, d1, d2, K, min_sep, margin):
    """Uniform centres with a minimum separation (dart throwing on a coarse grid hash)."""
    pts = []
    cell = max(min_sep, 1)
    grid = {}
    tries = 0
    while len(pts) < K and tries < 200 * K:
        tries += 1
        r = rng.uniform(margin, d1 - 1 - margin)
        c = rng.uniform(margin, d2 - 1 - margin)
        gr, gc = int(r // cell), int(c // cell)
        ok = True
        for a in (gr - 1, gr, gr + 1):
            for b in (gc - 1, gc, gc + 1):
                for (pr, pc) in grid.get((a, b), ()):
                    if (pr - r) ** 2 + (pc - c) ** 2 < min_sep ** 2:
                        ok = False
        if ok:
            grid.setdefault((gr, gc), []).append((r, c))
            pts.append((r, c))
    if len(pts) < K:
        raise ValueError("could not place %d neurons with separation %g" % (K, min_sep))
    return np.asarray(pts)


def _footprints(d1, d2, centres, amp, sig, siz):
    h = siz // 2
    rows, cols, vals = [], [], []
    for k, ((r, c), a) in enumerate(zip(centres, amp)):
        ri, ci = int(round(r)), int(round(c))
        rr = np.arange(max(0, ri - h), min(d1, ri + h + 1))
        cc = np.arange(max(0, ci - h), min(d2, ci + h + 1))
        g = a * np.exp(-((rr[:, None] - r) ** 2 + (cc[None, :] - c) ** 2) / (2.0 * sig * sig))
        pix = (cc[None, :] * d1 + rr[:, None]).ravel()
        rows.append(pix); cols.append(np.full(pix.size, k)); vals.append(g.ravel())
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(d1 * d2, len(amp)))
    A.sort_indices()
    return A


def _ar1(drive, g):
    from scipy.signal import lfilter
    return lfilter([1.0], [1.0, -g], drive, axis=-1)


def make_factors(d1, d2, T, K, seed, *, gSig=3.0, gSiz=13, min_sep=5.0,
                 noise_sd=1.0, bg_const=1000.0) -> SynthFactors:
    rng = np.random.default_rng(seed)
    margin = gSiz // 2 + 1
    if K > 0:
        centres = _place_centres(rng, d1, d2, K, min_sep, margin)
        amp = rng.uniform(0.5, 1.5, K)
        A_true = _footprints(d1, d2, centres, amp, gSig, gSiz)
        A_init = _footprints(d1, d2, centres, amp * (gSig ** 2) / (gSig ** 2 + 1.0),
                             np.sqrt(gSig ** 2 + 1.0), gSiz + 2)      # A blurred with sigma=1
        spikes = (rng.random((K, T)) < 0.01) * rng.uniform(5.0, 15.0, (K, T))
        C_true = _ar1(spikes, 0.95)
        C_init = np.maximum(C_true + rng.normal(0.0, 0.5, (K, T)), 0.0)
    else:
        A_true = sp.csc_matrix((d1 * d2, 0)); A_init = A_true.copy()
        C_true = np.zeros((0, T)); C_init = C_true.copy()
    # smooth background field: 8 broad Gaussians, amplitude 100..300
    rr, cc = np.meshgrid(np.arange(d1), np.arange(d2), indexing="ij")
    field = np.zeros((d1, d2))
    sig_bg = 60.0 * max(d1, d2) / 512.0 if max(d1, d2) < 512 else 60.0
    for _ in range(8):
        r0, c0 = rng.uniform(0, d1), rng.uniform(0, d2)
        field += rng.uniform(100.0, 300.0) * np.exp(-((rr - r0) ** 2 + (cc - c0) ** 2) / (2 * sig_bg ** 2))
    slow = _ar1(rng.normal(0.0, 1.0, T), 0.999) * np.sqrt(1 - 0.999 ** 2)   # unit stationary variance
    bg_time = 1.0 + 0.1 * slow
    return SynthFactors(d1, d2, T, K, A_true, C_true.astype(np.float32), A_init,
                        C_init.astype(np.float32),
                        field.reshape(-1, order="F").astype(np.float32),
                        bg_time.astype(np.float32), float(bg_const),
                        np.full(d1 * d2, noise_sd, dtype=np.float32), seed)


def make_video(f: SynthFactors, dtype=np.float32) -> np.ndarray:
    """Host synthesis.  Returns the video as a (T, d) array: frame-major, pixels column-major
    within a frame -- byte-identical to MATLAB's d x T column-major matrix."""
    rng = np.random.default_rng(f.seed + 7919)
    Y = (f.C_true.T.astype(np.float64) @ f.A_true.T.toarray()) if f.K > 0 else np.zeros((f.T, f.d))
    Y += np.outer(f.bg_time.astype(np.float64), f.bg_field.astype(np.float64))
    Y += f.bg_const
    Y += rng.normal(0.0, 1.0, (f.T, f.d)) * f.sn[None, :]
    return np.ascontiguousarray(Y.astype(dtype))


def make_video_device(f: SynthFactors, device="cuda:0", chunk=500, pixels=None):
    """Synthesise the (T, npix) float32 video directly in HBM with torch; returns the torch tensor.

    ``pixels``: global column-major pixel indices of a rectangle (e.g. one block), default the whole FOV.
    The noise of FOV column c comes from a generator seeded with (seed, c), so two ranks that synthesise
    overlapping blocks (ring halos) see identical data in the overlap.
    """
    import torch
    dev = torch.device(device)
    pix = np.arange(f.d, dtype=np.int64) if pixels is None else np.asarray(pixels, dtype=np.int64)
    npix = pix.size
    col_of = pix // f.d1
    cols = np.unique(col_of)
    nrb = npix // cols.size
    rows = pix[:nrb] % f.d1
    assert np.array_equal(pix, (cols[None, :] * f.d1 + rows[:, None]).reshape(-1, order="F")), "pixels must form a rectangle"
    r0, r1 = int(rows[0]), int(rows[-1]) + 1
    g = torch.Generator(device=dev)
    Y = torch.empty((f.T, npix), dtype=torch.float32, device=dev)
    for j, c in enumerate(cols):
        g.manual_seed(int(f.seed) * 1000003 + 7919 + int(c))
        Y[:, j * nrb:(j + 1) * nrb] = torch.randn((f.T, f.d1), generator=g, device=dev, dtype=torch.float32)[:, r0:r1]
    A = f.A_true.tocsr()[pix].tocoo()
    At = torch.sparse_coo_tensor(np.vstack([A.row, A.col]), A.data.astype(np.float32),
                                 (npix, f.K), device=dev).coalesce() if f.K > 0 and A.nnz else None
    field = torch.from_numpy(f.bg_field[pix]).to(dev)
    bt = torch.from_numpy(f.bg_time).to(dev)
    Ct = torch.from_numpy(f.C_true).to(dev)
    sn = torch.from_numpy(f.sn[pix]).to(dev)
    for t0 in range(0, f.T, chunk):
        t1 = min(f.T, t0 + chunk)
        blk = Y[t0:t1]
        blk *= sn[None, :]
        blk += bt[t0:t1, None] * field[None, :]
        blk += f.bg_const
        if At is not None:
            blk += torch.sparse.mm(At, Ct[:, t0:t1]).t()
    return Y
