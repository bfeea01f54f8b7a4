"""TEST INFRASTRUCTURE (like everything under oracle/): a second, independently written restatement of two OTHER functions of the reference
that solve the same sub-problems as the hot path, used to cross-check oracle/cnmfe_oracle.py on the cases the two formulations share
(tests/test_second_opinion.py).  Written with explicit loops from the .m text, sharing no code with cnmfe_oracle.py.

  local_background           endoscope/local_background.m:63-130      the stand-alone ring regression (no constant row, no footprints)
  updateTemporal_endoscope   @Sources2D/updateTemporal_endoscope.m:20-95  the older temporal HALS sweep (baseline / noise steps injectable)

Neither is on the hot path of demo_large_data_1p.m; they are here because they are the reference's own second statement of the ring
regression (fit_ring_model.m:92-108) and of the temporal sweep (HALS_temporal.m:59-68).  PARITY UNPINNED like the oracle itself.
"""
import numpy as np


def ring_offsets(rr):
    """local_background.m:64-68,87-89: offsets of the pixels with rr <= distance < rr + 1, in find() (column-major) order"""
    sub = np.arange(-rr, rr + 1)
    r_shift, c_shift = [], []
    for jc, c in enumerate(sub):                             # find() walks columns, then rows
        for jr, r in enumerate(sub):
            R = np.sqrt(float(c) ** 2 + float(r) ** 2)
            if R >= rr and R < rr + 1:
                r_shift.append(jr + 1 - rr - 1); c_shift.append(jc + 1 - rr - 1)
    return np.array(r_shift), np.array(c_shift)


def local_background(Y, rr, sn=None, thresh=np.inf):
    """[Yest, weights] = local_background(Y, 1, rr, [], sn, thresh, 1)  (local_background.m:21-130, ssub = 1, every pixel, p_cutoff = 1).
    Y: d1 x d2 x T.  thresh = Inf switches the event clipping of :69-72 off (then sn is not needed).
    Returns (Yest (d1*d2) x T, weights: list per pixel of (neighbour indices 0-based column-major, w))."""
    Y = np.array(Y, dtype=np.float64)
    d1, d2, T = Y.shape
    Y = Y - Y.mean(axis=2, keepdims=True)                    # :25-26
    r_shift, c_shift = ring_offsets(rr)
    ind_event = np.zeros((d1 * d2, T), dtype=bool)
    if np.isfinite(thresh):                                  # :67-72: ring mean (zero padded sum / in-bounds count), clip events
        Yconv = np.zeros_like(Y)
        cnt = np.zeros((d1, d2))
        for dr, dc in zip(r_shift, c_shift):
            for r in range(d1):
                for c in range(d2):
                    # imfilter is a correlation: out(r,c) = sum_k h(k) Y(r + k); the kernel is symmetric, so the sign does not matter
                    if 0 <= r + dr < d1 and 0 <= c + dc < d2:
                        Yconv[r, c, :] += Y[r + dr, c + dc, :]; cnt[r, c] += 1
        Yconv /= cnt[:, :, None]
        ev = (Y - Yconv) / np.asarray(sn, dtype=np.float64).reshape(d1, d2, 1) > thresh
        Y[ev] = Yconv[ev]
        ind_event = ev.reshape(d1 * d2, T, order="F")
    Yf = Y.reshape(d1 * d2, T, order="F")                    # :104
    Yest = np.zeros_like(Yf)
    weights = []
    for px in range(d1 * d2):                                # :108
        r, c = px % d1, px // d1
        nb = [(c + dc) * d1 + (r + dr) for dr, dc in zip(r_shift, c_shift) if 0 <= r + dr < d1 and 0 <= c + dc < d2]   # :91-100,110-111
        nb = np.array(nb, dtype=np.int64)
        # :114-116: tmp_ind = ~ind_event(px, 2:end) has T - 1 entries and indexes the columns from the FIRST one on
        tmp_ind = np.nonzero(~ind_event[px, 1:])[0]
        X = Yf[nb][:, tmp_ind]
        y = Yf[px, tmp_ind]
        XX = np.zeros((nb.size, nb.size)); Xy = np.zeros(nb.size)
        for i in range(nb.size):                             # :117-118
            Xy[i] = float(np.dot(X[i], y))
            for j in range(i, nb.size):
                XX[i, j] = XX[j, i] = float(np.dot(X[i], X[j]))
        w = np.linalg.solve(XX + np.eye(nb.size) * np.trace(XX) * 1e-5, Xy)   # :126
        Yest[px, :] = w @ Yf[nb, :]                          # :127
        weights.append((nb, w))                              # :128
    return Yest, weights


def updateTemporal_endoscope(Y, A, C, maxIter, baseline=None, noise=None, post=None):
    """The sweep of @Sources2D/updateTemporal_endoscope.m:20-95 with deconv_flag = false.  The reference's baseline / noise steps (:49-58:
    estimate_baseline_noise, GetSn) need fit_gauss1 and pwelch; they are injected here: baseline(temp) -> b, noise(temp) -> sn,
    post(temp - b) -> ck (default max(0, .), :70).  Returns (C, C_raw, sn) BEFORE the rescaling by sn of :86-88."""
    Y = np.asarray(Y, dtype=np.float64); A = np.asarray(A, dtype=np.float64)
    C = np.array(C, dtype=np.float64)
    K, T = C.shape
    baseline = baseline or (lambda t: 0.0)
    noise = noise or (lambda t: 1.0)
    post = post or (lambda t: np.maximum(0.0, t))
    C_raw = np.zeros((K, T))
    U = np.zeros((K, T)); V = np.zeros((K, K))
    for k in range(K):                                       # :27-28
        for t in range(T):
            U[k, t] = float(np.dot(A[:, k], Y[:, t]))
        for j in range(K):
            V[k, j] = float(np.dot(A[:, k], A[:, j]))
    aa = np.array([V[k, k] for k in range(K)])               # :29
    sn = np.zeros(K)
    ind_del = np.zeros(K, dtype=bool)
    for miter in range(1, maxIter + 1):                      # :36
        for k in range(K):
            if aa[k] == 0:                                   # :38-44
                C_raw[k] = 0; C[k] = 0; ind_del[k] = True
                continue
            if ind_del.all():                                # :45-47 (`if ind_del` on a vector is all())
                continue
            temp = C[k] + (U[k] - V[k] @ C) / aa[k]          # :48
            b = baseline(temp)                               # :50-58
            temp = temp - b                                  # :60
            sn[k] = noise(temp)
            ck = post(temp)                                  # :70
            C[k] = ck                                        # :73
            if ck[1:].sum() == 0:                            # :75-77
                ind_del[k] = True
            if miter == maxIter:                             # :79-84
                C_raw[k] = temp
    return C, C_raw, sn
