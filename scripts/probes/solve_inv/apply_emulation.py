"""Float64 emulation of k_ring_apply's arithmetic (round 6): cached K = inv(G0 + lam0 D_ex) per pixel (explicit), k_g = K g0, k_u = K u0;
per fit V = [A~ | U~] (8 + 8 columns; pairs without a ring pixel masked), H = V'KV, cap = S + H inverted by Gauss-Jordan WITHOUT pivoting (A~ block first: a Gram matrix, then the negative definite Schur complement),
border (ones row) by its Schur complement, Neumann series in delta = lam - lam0 with C0 v = K (v - V capinv V' K v).  Compared with the direct solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from cnmf_e_amd import synth

def gj_inverse_nopivot(M):
    a = M.copy(); n = a.shape[0]
    for p in range(n):
        piv = a[p, p]; inv = 1.0 / piv
        rowp = a[p] * inv; rowp[p] = inv
        col = a[:, p].copy()
        a -= np.outer(col, rowp); a[:, p] = -col * inv
        a[p] = rowp
    return a

d1 = d2 = 96; T = 3000; K = 20; r = 15
f = synth.make_factors(d1, d2, T, K, 5)
Y = synth.make_video(f, np.float64); Y = Y if Y.shape[0] == d1 * d2 else Y.T
kstride = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Yc = Y - Y.mean(1, keepdims=True)
A = np.asarray(f.A_init.todense(), dtype=np.float64); C = f.C_init.astype(np.float64); Cc = C - C.mean(1, keepdims=True)
Yc = Yc[:, ::kstride]; Cc = Cc[:, ::kstride]; Tp = Yc.shape[1]
rr, cc = np.mgrid[-r - 1:r + 2, -r - 1:r + 2]
dist = np.sqrt(rr ** 2 + cc ** 2); sel = (dist >= r) & (dist < r + 1); dr, dc = rr[sel], cc[sel]
P = Yc @ Cc.T; Gc = Cc @ Cc.T; Ut = P - A @ Gc / 2
Bf = Yc - A @ Cc
csum = Cc.sum(1)
rng = np.random.default_rng(0)
stats = {}
nskip = 0
for trial in range(200):
    i, j = rng.integers(0, d1), rng.integers(0, d2)
    ri, cj = i + dr, j + dc
    ex = (ri >= 0) & (ri < d1) & (cj >= 0) & (cj < d2)
    ring = np.where(ex, cj * d1 + ri, 0); m = j * d1 + i; n = ring.size
    Dex = ex.astype(np.float64)
    X0 = Yc[ring] * Dex[:, None]; G0 = X0 @ X0.T + np.diag(1.0 - Dex)            # missing rows: unit rows (fill 1), no ridge
    g0 = X0 @ Yc[m]; u0 = X0.sum(1)
    ks = np.flatnonzero(((A[ring] != 0) & ex[:, None]).any(0) | (A[m] != 0))
    if ks.size > 8: nskip += 1; continue
    s = ks.size
    Ur = Ut[np.ix_(ring, ks)] * Dex[:, None]; Ar = A[np.ix_(ring, ks)] * Dex[:, None]
    uN, aN = Ut[m, ks], A[m, ks]
    # reference
    Xr = np.vstack([Bf[ring[ex]], np.ones(Tp)]); y = Bf[m]
    XX = Xr @ Xr.T; Xy = Xr @ y
    lam = 1e-5 * np.trace(XX)
    wref = np.zeros(n); wref[ex] = np.linalg.solve(XX + lam * np.eye(XX.shape[0]), Xy)[:-1]
    sc = y.sum()
    for eps in (0.0, 1e-3, 1e-2, 5e-2):
        lam0 = lam / (1 + eps); delta = lam - lam0
        Kinv = np.linalg.inv(G0 + lam0 * np.diag(Dex))
        k_g = Kinv @ g0; k_u = Kinv @ u0
        # V = [A~ | U~]: -u~ a~' - a~ u~' = V S V', S = [[0, -I], [-I, 0]] = inv(S).  A neuron without a pixel on the ring (under the centre only: the usual case)
        # corrects g alone: its pair is masked out of cap (its u~ column stays in V for the cached-vector bookkeeping)
        V = np.zeros((n, 16)); cg = np.zeros(16); cu = np.zeros(16); live = np.zeros(16, bool)
        for q in range(s):
            V[:, q] = Ar[:, q]; V[:, 8 + q] = Ur[:, q]
            cg[q] = uN[q]; cg[8 + q] = aN[q]; cu[q] = csum[ks[q]]
            live[q] = live[8 + q] = np.any(Ar[:, q] != 0)
        S = np.zeros((16, 16))
        for q in range(8): S[q, 8 + q] = S[8 + q, q] = -1.0
        H = V.T @ (Kinv @ V)
        cap = S + H
        cap[~live] = 0; cap[:, ~live] = 0; cap[~live, ~live] = 1.0
        capinv = gj_inverse_nopivot(cap); capinv[~live] = 0; capinv[:, ~live] = 0
        hg = V.T @ k_g; hu = V.T @ k_u
        g = g0 - V @ cg; u = u0 - V @ cu
        tau = Tp + lam
        def C0(v):                                   # core inverse at lam0 (Woodbury), no Z
            t = Kinv @ v
            return Kinv @ (v - V @ (capinv @ (V.T @ t)))
        # scalars through the cached vectors
        VKg = hg - H @ cg; VKu = hu - H @ cu
        uKg = u0 @ k_g - cu @ hg - hu @ cg + cu @ H @ cg
        uKu = u0 @ k_u - 2 * cu @ hu + cu @ H @ cu
        yg = capinv @ VKg; yu = capinv @ VKu
        uCg = uKg - VKu @ yg; uCu = uKu - VKu @ yu
        Cu = Kinv @ (u - V @ yu)
        w0 = (sc - uCg) / (tau - uCu)
        x0 = Kinv @ (g - w0 * u - V @ (yg - w0 * yu))
        x = x0.copy(); errs = [np.abs(x - wref).max() / np.abs(wref).max()]
        for it in range(4):
            v = Dex * x
            Cv = C0(v)
            w0p = -(u @ Cv) / (tau - uCu)
            x = x0 - delta * (Cv - w0p * Cu)
            errs.append(np.abs(x - wref).max() / np.abs(wref).max())
        stats[eps] = np.maximum(stats.get(eps, 0), errs)
print("kstride", kstride, "pixels skipped (> 8 neurons):", nskip, "of 200")
for k, v in stats.items():
    print("lam/lam0 - 1 = %+.0e : max rel err of w after 0..4 Neumann terms: %s" % (k, " ".join("%.1e" % e for e in v)))
