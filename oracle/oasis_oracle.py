"""CPU oracle for the deconvolution branch of the CNMF-E temporal update  --  TEST INFRASTRUCTURE ONLY.

Float64 restatement of the AR(1) FOOPSI path the target demo selects
(`demos/demo_large_data_1p.m:38-43`: type 'ar1', method 'foopsi', smin=-5, optimize_pars, optimize_b,
max_tau=100), following (paths under /root/reference/OASIS_matlab/ unless stated):
  deconvolveCa.m:61-123,199-206      packages/oasis/foopsi_oasisAR1.m:36-180
  packages/oasis/oasisAR1.m:30-109   functions/GetSn.m:19-46   functions/estimate_time_constant.m:21-66
  ca_source_extraction/utilities/HALS_temporal.m:70-104        ca_source_extraction/@Sources2D/deconvTemporal.m:29-105

PARITY UNPINNED: the MATLAB toolbox functions on this path are not in the reference
tree.  They are restated from their published definitions and flagged here:
  * pwelch(x,[],[],[],1): Hamming window of length floor(N/4.5), 50 % overlap (8 segments), nfft =
    max(256, 2^nextpow2(L)), one-sided density with fs=1, no detrending  (MathWorks documentation).
  * fminbnd: Brent's golden-section / parabolic-interpolation minimiser (Forsythe, Malcolm & Moler `fmin`),
    TolX = 1e-4, as documented for fminbnd.  The reference also depends on the LAST point fminbnd
    evaluated (foopsi_oasisAR1.m:156 reuses `h` from the last rss_g call); this restatement exposes it.
  * quantile(y, .15): linear interpolation at (i-0.5)/n.   median: mean of the two middle values.
  * estimate_time_constant adds 0.001*randn jitter to out-of-range roots (:62-64) -- non-deterministic even
    in MATLAB; the jitter is dropped here (g<0 -> 0.15).  An unstable AR(1) estimate (|g|>1) recurses to
    higher orders and deconvolveCa then returns zeros (:84-89); restated as: zeros, pars = 0.
"""
from __future__ import annotations

import numpy as np

__all__ = ["pwelch_psd", "GetSn", "estimate_time_constant_ar1", "oasisAR1", "fminbnd", "foopsi_oasisAR1",
           "deconvolveCa_ar1_foopsi", "HALS_temporal_deconv", "deconvTemporal", "matlab_quantile"]


def pwelch_psd(x):
    """[psdx, ff] = pwelch(x, [], [], [], 1) for a real vector (MathWorks-documented defaults)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    N = x.size
    L = int(np.floor(N / 4.5))
    nov = int(np.floor(L / 2))
    nfft = max(256, 1 << int(np.ceil(np.log2(L))))
    n = np.arange(L)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * n / (L - 1))                       # hamming(L), symmetric
    step = L - nov
    nseg = (N - nov) // step
    acc = np.zeros(nfft // 2 + 1)
    for k in range(nseg):
        seg = x[k * step:k * step + L] * w
        X = np.fft.rfft(seg, nfft)
        acc += (X.real ** 2 + X.imag ** 2)
    psd = acc / (nseg * (w ** 2).sum())                                    # fs = 1
    psd[1:-1] *= 2.0                                                       # one-sided (nfft even)
    return psd, np.arange(nfft // 2 + 1) / nfft


def GetSn(y, range_ff=(0.25, 0.5)):
    """functions/GetSn.m:19-46, method 'logmexp'."""
    psd, ff = pwelch_psd(y)
    ind = (ff >= range_ff[0]) & (ff <= range_ff[1])                        # :34
    return float(np.sqrt(np.exp(np.mean(np.log(psd[ind] / 2.0)))))         # :41


def estimate_time_constant_ar1(y, sn):
    """functions/estimate_time_constant.m:21-66 with p=1, lags=5.  Returns g, or None if the AR(1) estimate is
    unstable (|g|>1; the reference then recurses to p=2.. and deconvolveCa.m:84-89 gives up)."""
    y = np.asarray(y, dtype=np.float64).ravel()
    lags = 5 + 1                                                           # :36
    yn = y - y.mean()
    xc = np.array([yn[k:] @ yn[:y.size - k] for k in range(lags + 1)]) / y.size        # :40-45 (biased)
    A = xc[:lags].copy()                                                   # toeplitz(xc(lags+(1:lags)), xc(lags+1)) is the column c0..c5  (:48)
    A[0] -= sn ** 2
    g = float(A @ xc[1:lags + 1] / (A @ A))                                # pinv(A)*xc(lags+2:end)  (:49)
    if abs(g) > 1:                                                         # :51
        return None
    if g < 0:                                                              # :64 (jitter dropped)
        g = 0.15
    return g


def matlab_quantile(y, q):
    s = np.sort(np.asarray(y, dtype=np.float64).ravel())
    n = s.size
    pos = q * n - 0.5                                                      # 0-based position of the (i-0.5)/n rule
    if pos <= 0:
        return float(s[0])
    if pos >= n - 1:
        return float(s[-1])
    i = int(np.floor(pos))
    return float(s[i] + (pos - i) * (s[i + 1] - s[i]))


def oasisAR1(y, g, lam=0.0, smin=0.0, active_set=None):
    """packages/oasis/oasisAR1.m:30-109.  active_set rows: (v, w, t, l) with 1-based t.
    Returns (c, s, active_set).  Restated as the equivalent left-to-right pool stack (same merges, same order,
    same arithmetic as the linked-list walk :60-96)."""
    y = np.asarray(y, dtype=np.float64).ravel()
    T = y.size
    if active_set is None or len(active_set) == 0:
        v = y - lam * (1 - g); v[-1] = y[-1] - lam                        # :49-50
        pools_in = [(v[i], 1.0, i + 1, 1) for i in range(T)]
    else:
        pools_in = [tuple(r[:4]) for r in active_set]
    st = []                                                                # processed pools
    cur = list(pools_in[0])
    for nxt in pools_in[1:]:
        nxt = list(nxt)
        if nxt[0] / nxt[1] >= cur[0] / cur[1] * g ** cur[3] + smin:        # :64-65 no violation: advance
            st.append(cur); cur = nxt
            continue
        cur[0] += nxt[0] * g ** cur[3]; cur[1] += nxt[1] * g ** (2 * cur[3]); cur[3] += nxt[3]     # :74-76 merge
        while st and cur[0] / cur[1] < max(0.0, st[-1][0] / st[-1][1] * g ** st[-1][3]) + smin:   # :83-84 backtrack
            prv = st.pop()
            prv[0] += cur[0] * g ** prv[3]; prv[1] += cur[1] * g ** (2 * prv[3]); prv[3] += cur[3]  # :87-89
            cur = prv
    st.append(cur)
    c = np.zeros(T); s = np.zeros(T)
    for (vv, ww, t0, l) in st:                                             # :103-107
        t0 = int(t0); l = int(l)
        c[t0 - 1:t0 - 1 + l] = max(0.0, vv / ww) * g ** np.arange(l)
    for (vv, ww, t0, l) in st[1:]:                                         # :109
        t0 = int(t0)
        s[t0 - 1] = c[t0 - 1] - g * c[t0 - 2]
    return c, s, [list(p) for p in st]


def fminbnd(f, ax, bx, tol=1e-4, maxiter=500):
    """Brent's method as documented for MATLAB fminbnd.  Returns (xmin, last_x_evaluated)."""
    seps = np.sqrt(np.finfo(float).eps)
    cgold = 0.5 * (3.0 - np.sqrt(5.0))
    a, b = ax, bx
    v = a + cgold * (b - a); w = v; xf = v
    d = 0.0; e = 0.0
    x = xf; fx = f(x); last = x
    fv = fx; fw = fx
    xm = 0.5 * (a + b)
    tol1 = seps * abs(xf) + tol / 3.0
    tol2 = 2.0 * tol1
    it = 0
    while abs(xf - xm) > (tol2 - 0.5 * (b - a)) and it < maxiter:
        it += 1
        gs = True
        if abs(e) > tol1:                                                  # parabolic fit
            gs = False
            r = (xf - w) * (fx - fv)
            q = (xf - v) * (fx - fw)
            p = (xf - v) * q - (xf - w) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = abs(q)
            r = e
            e = d
            if (abs(p) < abs(0.5 * q * r)) and (p > q * (a - xf)) and (p < q * (b - xf)):
                d = p / q
                x = xf + d
                if ((x - a) < tol2) or ((b - x) < tol2):
                    si = np.sign(xm - xf) + ((xm - xf) == 0)
                    d = tol1 * si
            else:
                gs = True
        if gs:                                                             # golden-section step
            e = (a - xf) if xf >= xm else (b - xf)
            d = cgold * e
        si = np.sign(d) + (d == 0)
        x = xf + si * max(abs(d), tol1)
        fu = f(x); last = x
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            v, fv = w, fw
            w, fw = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if (fu <= fw) or (w == xf):
                v, fv = w, fw
                w, fw = x, fu
            elif (fu <= fv) or (v == xf) or (v == w):
                v, fv = x, fu
        xm = 0.5 * (a + b)
        tol1 = seps * abs(xf) + tol / 3.0
        tol2 = 2.0 * tol1
    return xf, last


def _update_g(y, active_set, lam, smin, g_range=(0.0, 1.0)):
    """foopsi_oasisAR1.m:122-180."""
    y = np.asarray(y, dtype=np.float64).ravel()
    pools = [list(p) for p in active_set]
    maxl = int(max(p[3] for p in pools))                                   # :147
    c = np.zeros_like(y)

    def rss_g(g):                                                          # :166-179
        h = np.exp(np.log(g) * np.arange(maxl + 1))
        hh = np.cumsum(h * h)
        yp = y - lam * (1 - g)
        for (_, _, ti, li) in pools:
            ti = int(ti); li = int(li)
            seg = yp[ti - 1:ti - 1 + li]
            tmp_v = max(seg @ h[:li] / hh[li - 1], 0.0)
            c[ti - 1:ti - 1 + li] = tmp_v * h[:li]
        res = y - c
        return float(res @ res)

    g, g_last = fminbnd(rss_g, g_range[0], g_range[1])                     # :152
    yp = y - lam * (1 - g)                                                 # :153
    tmp_h = np.exp(np.log(g) * np.arange(maxl + 1))                        # :155
    h_last = np.exp(np.log(g_last) * np.arange(maxl + 1))                  # `h` left over from the last rss_g call
    tmp_hh = np.cumsum(h_last * h_last)                                    # :156 (sic)
    for p in pools:                                                        # :154-162
        ti = int(p[2]); li = int(p[3])
        p[0] = float(yp[ti - 1:ti - 1 + li] @ tmp_h[:li])
        p[1] = float(tmp_hh[li - 1])
    c, s, pools = oasisAR1(y, g, lam, smin, pools)                         # :163
    return c, pools, g, s


def foopsi_oasisAR1(y, g, lam, smin, optimize_b, optimize_g, maxIter, gmax, sn_for_restart=None):
    """foopsi_oasisAR1.m:36-120 (decimate unused, tau_range = [] -> g_range = [0,1])."""
    y = np.asarray(y, dtype=np.float64).ravel()
    if not optimize_b:                                                     # :82-90
        b = 0.0
        sol, spks, aset = oasisAR1(y, g, lam, smin)
        if optimize_g:
            sol, aset, g, spks = _update_g(y, aset, lam, smin)
        return sol, spks, b, g
    b = matlab_quantile(y, 0.15)                                           # :93
    sol, spks, aset = oasisAR1(y - b, g, lam, smin)                        # :94
    for _ in range(maxIter):                                               # :97
        b = float(np.mean(y - sol))                                        # :98
        if optimize_g:
            if len(aset) == 0:
                break
            g0 = g
            if g > gmax:                                                   # :104-108
                g = estimate_time_constant_ar1(y, GetSn(y))
                if g is None:
                    g = g0
                sol, spks, aset = oasisAR1(y - b, g, lam, smin)
                break
            sol, aset, g, spks = _update_g(y - b, aset, lam, smin)         # :109
            if abs(g - g0) / g0 < 1e-3:                                    # :110
                optimize_g = False
        else:
            break
    return sol, spks, b, g


def deconvolveCa_ar1_foopsi(y, sn, pars=None, maxIter=10, smin=-5.0, optimize_b=True, optimize_pars=True, max_tau=100.0, lam=0.0):
    """deconvolveCa.m:61-123,199-206 for type='ar1', method='foopsi'.  Returns (c, s, b, pars)."""
    y = np.asarray(y, dtype=np.float64).ravel()
    if pars is None or pars == 0:                                          # :73
        pars = estimate_time_constant_ar1(y, sn)                           # :77
        if pars is None:                                                   # :84-89
            return np.zeros_like(y), np.zeros_like(y), 0.0, 0.0
    if smin < 0:
        smin = abs(smin) * sn                                              # :116-118
    gmax = np.exp(-1.0 / max_tau)                                          # :119
    c, s, b, g = foopsi_oasisAR1(y, pars, lam, smin, optimize_b, optimize_pars, maxIter, gmax)     # :120-122 (b0 = 0)
    c = np.where(np.isfinite(c), c, 0.0)                                   # :206
    return c, s, b, g


def HALS_temporal_deconv(Y, A, C, maxIter, **opt):
    """utilities/HALS_temporal.m:47-117 with deconv_options given (the :70-104 branch).
    Returns (C, C_raw, S, sn, kernel_pars)."""
    import scipy.sparse as sp
    Y = np.asarray(Y, dtype=np.float64)
    A = np.asarray(A.toarray() if sp.issparse(A) else A, dtype=np.float64)
    C = np.array(C, dtype=np.float64, copy=True)
    K, T = C.shape
    C_raw = np.zeros((K, T)); S = np.zeros((K, T)); sn = np.zeros(K); kp = [None] * K
    U = A.T @ Y; V = A.T @ A; aa = np.diag(V).copy()
    for it in range(maxIter):
        for k in np.nonzero(aa > 0)[0]:
            ck_raw = C[k] + (U[k] - V[k] @ C) / aa[k]                      # :62
            srt = np.sort(ck_raw)
            med = 0.5 * (srt[(T - 1) // 2] + srt[T // 2])
            b = ck_raw[ck_raw < med].mean()                                # :78
            sn_psd = GetSn(ck_raw)                                         # :79
            ck_raw = ck_raw - b                                            # :88
            sn[k] = sn_psd
            ck, sk, bb, g = deconvolveCa_ar1_foopsi(ck_raw, sn_psd, kp[k], maxIter=20, **opt)      # :92
            kp[k] = g
            ck_raw = ck_raw - bb                                           # :94
            if np.abs(ck).sum() == 0:
                ck = ck_raw                                                # :95-97
            C[k] = ck
            if it == maxIter - 1:
                S[k] = sk; C_raw[k] = ck_raw                               # :100-103
    return C, C_raw, S, sn, kp


def deconvTemporal(C_raw, **opt):
    """@Sources2D/deconvTemporal.m:29-105 (method_noise='psd').  Returns (C, C_raw, S, kernel_pars, sn)."""
    C_raw = np.array(C_raw, dtype=np.float64, copy=True)
    K, T = C_raw.shape
    C = np.zeros((K, T)); S = np.zeros((K, T)); kp = np.zeros(K); sn = np.zeros(K)
    for k in range(K):
        ck_raw = C_raw[k]
        if np.isnan(ck_raw).any():                                         # :37-40
            C_raw[k] = 0
            continue
        sn[k] = GetSn(ck_raw)                                              # :45
        ck, sk, b, g = deconvolveCa_ar1_foopsi(ck_raw, sn[k], None, maxIter=10, **opt)             # :51
        if np.abs(ck).sum() == 0:
            ck = ck_raw                                                    # :53-55
        C[k] = ck; S[k] = sk; kp[k] = g
        C_raw[k] = ck_raw - b                                              # :59
    return C, C_raw, S, kp, sn
