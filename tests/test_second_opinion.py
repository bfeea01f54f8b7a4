"""oracle/cnmfe_oracle.py against oracle/second_opinion.py: the reference's OTHER statements of the ring regression
(endoscope/local_background.m) and of the temporal sweep (@Sources2D/updateTemporal_endoscope.m), restated independently, on the cases they
share with fit_ring_model.m / HALS_temporal.m.  Neither side is MATLAB output -- two restatements agreeing rules out transcription slips,
not misreadings of MATLAB semantics (parity unpinned; oracle/matlab/make_fixtures.m is the route that pins it)."""
import numpy as np
import scipy.sparse as sp

import cnmfe_oracle as orc
import second_opinion as so


def test_ring_offsets_are_the_same_set_in_the_same_order():
    for rr in (3, 5, 8, 15, 18):
        r1, c1 = orc.get_nhood(rr)
        r2, c2 = so.ring_offsets(rr)
        assert np.array_equal(np.asarray(r1).ravel(), r2) and np.array_equal(np.asarray(c1).ravel(), c2), rr


def test_ring_regression_agrees_with_local_background():
    """Shared case: no footprints, no events, every row of the data with zero mean over the frames the regression uses.  Then the constant
    row of fit_ring_model.m:101 decouples (X*1' = 0 -> w0 = 0) and the two differ only by T in the ridge, 1e-5*(trace + T) against
    1e-5*trace: with unit-variance data a relative 1/p of a 1e-5 perturbation.  local_background uses frames 1..T-1 of its T (its :114
    quirk), so the video it gets is the regression's frames plus one frame that makes every pixel's mean over all T equal the mean over
    the first T - 1."""
    rng = np.random.default_rng(12)
    d1, d2, T, rr = 14, 12, 90, 4
    Z = rng.standard_normal((d1 * d2, T)) * rng.uniform(20, 40, (d1 * d2, 1))
    Z += 15 * np.sin(np.arange(T) / 7.0)[None, :] * rng.uniform(0.5, 1.5, (d1 * d2, 1))   # shared background-like component
    Z -= Z.mean(axis=1, keepdims=True)
    offset = rng.uniform(100, 300, (d1 * d2, 1))
    Yfull = np.concatenate([Z, np.zeros((d1 * d2, 1))], axis=1) + offset               # centring over T + 1 frames gives [Z, 0] back
    _, weights = so.local_background(Yfull.reshape(d1, d2, T + 1, order="F"), rr)
    patch = np.array([1, d1, 1, d2]); block = patch.copy()
    r_shift, c_shift = orc.get_nhood(rr)
    W0 = orc.build_ring_W(patch, block, d1, d2, r_shift, c_shift)
    W, b0 = orc.fit_ring_model(Z + offset, None, None, W0, np.nan, None, np.ones(d1 * d2, bool), with_projection=False)
    W = sp.csr_matrix(W); W.sort_indices()
    worst = 0.0
    for px in range(d1 * d2):
        nb, w = weights[px]
        row = W.getrow(px)
        order = np.argsort(nb)
        assert np.array_equal(row.indices, nb[order]), px
        worst = max(worst, np.abs(row.data - w[order]).max() / np.abs(w).max())
    assert worst <= 1e-6, worst
    assert np.allclose(b0, offset.ravel(), rtol=1e-12)


def test_ring_regression_event_clipping_matches_the_outlier_branch_idea():
    """local_background.m:67-72 and fit_ring_model.m:50-56 clip the same way (value > ring estimate + thresh*sn -> ring estimate); with the
    initial uniform ring matrix as W_old the ring estimate IS the in-bounds ring mean, so the clipped videos must coincide."""
    rng = np.random.default_rng(3)
    d1, d2, T, rr, thr = 12, 10, 40, 3, 1.0
    Y = rng.standard_normal((d1, d2, T)) * 5 + 50
    Y[rng.random(Y.shape) < 0.05] += 30
    sn = rng.uniform(2, 4, (d1, d2))
    Yc = Y - Y.mean(axis=2, keepdims=True)
    r_shift, c_shift = orc.get_nhood(rr)
    patch = np.array([1, d1, 1, d2])
    W0 = sp.csr_matrix(orc.build_ring_W(patch, patch, d1, d2, r_shift, c_shift))
    Bf = Yc.reshape(d1 * d2, T, order="F")
    Bf_old = W0 @ Bf
    clipped = np.where(Bf > Bf_old + thr * sn.reshape(-1, 1, order="F"), Bf_old, Bf)       # fit_ring_model.m:51-55
    # second opinion: run local_background's clipping and read the clipped video back through its estimate of a pixel from itself
    Yest, weights = so.local_background(Y, rr, sn=sn, thresh=thr)
    # reconstruct the clipped video local_background regressed on: Yest(px) = w' * Yclipped(nb), so compare through the same weights
    ref = np.stack([w @ clipped[nb] for nb, w in weights])
    assert np.abs(Yest - ref).max() <= 1e-9 * np.abs(ref).max()


def test_temporal_sweep_agrees_with_updateTemporal_endoscope():
    """Shared case: the Gauss-Seidel step temp = C(k,:) + (U(k,:) - V(k,:)*C)/aa(k) (updateTemporal_endoscope.m:48 == HALS_temporal.m:62) with
    HALS_temporal's post-step (subtract the minimum, :66) injected; one neuron has an empty footprint (aa = 0: skipped by both)."""
    rng = np.random.default_rng(8)
    d, K, T = 120, 6, 70
    A = np.abs(rng.standard_normal((d, K))) * (rng.random((d, K)) < 0.3)
    A[:, 4] = 0.0
    Ctrue = np.abs(rng.standard_normal((K, T))).cumsum(axis=1) % 5
    Y = A @ Ctrue + 0.1 * rng.standard_normal((d, T))
    C0 = np.abs(Ctrue + 0.3 * rng.standard_normal((K, T)))
    C1, Craw1, _ = orc.HALS_temporal(Y, A, C0, maxIter=4)
    C2, Craw2, _ = so.updateTemporal_endoscope(Y, A, C0, 4, post=lambda t: t - t.min())
    upd = np.diag(A.T @ A) > 0
    assert np.abs(C1[upd] - C2[upd]).max() <= 1e-10 * np.abs(C1).max()
    # (C_raw: HALS_temporal stores the shifted trace, updateTemporal_endoscope the unshifted one of the last sweep)
    assert np.abs((Craw2[upd] - Craw2[upd].min(axis=1, keepdims=True)) - Craw1[upd]).max() <= 1e-10 * np.abs(Craw1).max()
