"""CPU emulation (float64 NumPy) of the ring solve out of a cached explicit inverse (round 6):
    M(lam) = G0 + lam I - U~ A' - A U~'        (core, ring pixels of one centre; G0 = the VIDEO's Gram, corrections = the footprints' rank-2 terms)
    K      = inv(G0 + lam0 I)                  cached per pixel
    inv(M) through Woodbury on K (exact in A, C) + a Neumann series in (lam - lam0)
against the direct solve of fit_ring_model.m:101-107.  Prints the error of w per number of Neumann terms for several lam0 / lam."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from cnmf_e_amd import synth

d1 = d2 = 64; T = 2000; K = 24; r = 15
f = synth.make_factors(d1, d2, T, K, 3)
Y = synth.make_video(f, np.float64); Y = Y if Y.shape[0] == d1 * d2 else Y.T
d = d1 * d2
Yc = Y - Y.mean(1, keepdims=True)
A = np.asarray(f.A_init.todense(), dtype=np.float64); C = f.C_init.astype(np.float64); Cc = C - C.mean(1, keepdims=True)
# ring offsets (get_nhood.m): pixels at rounded distance r
rr, cc = np.mgrid[-r - 1:r + 2, -r - 1:r + 2]
dist = np.sqrt(rr ** 2 + cc ** 2)
sel = (dist >= r) & (dist < r + 1)
dr, dc = rr[sel], cc[sel]
P = Yc @ Cc.T; Gc = Cc @ Cc.T
Ut = P - A @ Gc / 2
rng = np.random.default_rng(0)
Bf = Yc - A @ Cc
worst = {}
for trial in range(40):
    i, j = rng.integers(r + 1, d1 - r - 1), rng.integers(r + 1, d2 - r - 1)
    ring = (j + dc) * d1 + (i + dr); m = j * d1 + i          # column-major pixels
    n = ring.size
    X0 = Yc[ring]; G0 = X0 @ X0.T; g0 = X0 @ Yc[m]
    ks = np.flatnonzero((A[ring] != 0).any(0) | (A[m] != 0))
    Ur, Ar = Ut[np.ix_(ring, ks)], A[np.ix_(ring, ks)]
    uN, aN = Ut[m, ks], A[m, ks]
    # reference: direct
    X = np.vstack([Bf[ring], np.ones(T)]); y = Bf[m]
    XX = X @ X.T; Xy = X @ y
    lam = 1e-5 * np.trace(XX)
    wref = np.linalg.solve(XX + lam * np.eye(n + 1), Xy)[:-1]
    # pieces as the engine has them
    Gcorr = G0 - Ur @ Ar.T - Ar @ Ur.T
    assert np.allclose(Gcorr, XX[:n, :n], rtol=1e-9, atol=1e-6 * np.abs(XX).max())
    g = g0 - Ur @ aN - Ar @ uN
    u = XX[:n, n]; sc = Xy[n]
    s = ks.size
    for eps in (0.0, 1e-4, 1e-3, 1e-2, 1e-1, -1e-1, 0.5):
        lam0 = lam / (1 + eps); delta = lam - lam0
        Kinv = np.linalg.inv(G0 + lam0 * np.eye(n))
        # full (n+1) system: diag(P0, tau) + rank-2 border + corrections; V = [U~ A u_ e], S couples (U~,A) and (u_, e)
        tau = T + lam
        V = np.zeros((n + 1, 2 * s + 2)); V[:n, :s] = Ur; V[:n, s:2 * s] = Ar; V[:n, 2 * s] = u; V[n, 2 * s + 1] = 1.0
        S = np.zeros((2 * s + 2, 2 * s + 2)); S[:s, s:2 * s] = -np.eye(s); S[s:2 * s, :s] = -np.eye(s); S[2 * s, 2 * s + 1] = S[2 * s + 1, 2 * s] = 1.0
        Kf = np.zeros((n + 1, n + 1)); Kf[:n, :n] = Kinv; Kf[n, n] = 1.0 / tau
        rhs = np.concatenate([g, [sc]])
        Z = Kf @ V; cap = np.linalg.inv(S) + V.T @ Z
        def C0(v): return Kf @ v - Z @ np.linalg.solve(cap, Z.T @ v)
        Dex = np.ones(n + 1); Dex[n] = 0.0
        x0 = C0(rhs); x = x0.copy(); errs = [np.abs(x[:n] - wref).max() / np.abs(wref).max()]
        for it in range(6):
            x = x0 - delta * C0(Dex * x)
            errs.append(np.abs(x[:n] - wref).max() / np.abs(wref).max())
        key = eps
        worst[key] = np.maximum(worst.get(key, 0), errs)
    if trial == 0: print("n", n, "neurons on ring", s, "cond(G0+lam)", np.linalg.cond(G0 + lam * np.eye(n)), "trcorr/tr0", (np.trace(XX[:n,:n]) - np.trace(G0)) / np.trace(G0))
for k, v in worst.items():
    print("lam/lam0 - 1 = %+.0e : max rel err of w after 0..6 Neumann terms: %s" % (k, " ".join("%.1e" % e for e in v)))
