"""Self-checks pinning the deconvolution oracle (parity unpinned vs MATLAB; see oracle/oasis_oracle.py header)."""
import numpy as np
import pytest

import oasis_oracle as oo


def _trace(T=1500, g=0.95, sn=0.3, seed=13, rate=0.01, amp=1.0):
    # same family as OASIS_matlab/functions/gen_data.m:31-41
    rng = np.random.default_rng(seed)
    s = (rng.random(T) < rate).astype(float) * amp
    c = np.zeros(T)
    for t in range(T):
        c[t] = (g * c[t - 1] if t else 0.0) + s[t]
    return c + 0.5 + sn * rng.standard_normal(T), c, s


def test_pwelch_matches_scipy_welch():
    from scipy.signal import welch, get_window
    y, _, _ = _trace(3000)
    psd, ff = oo.pwelch_psd(y)
    L = int(np.floor(y.size / 4.5))
    f2, p2 = welch(y, fs=1.0, window=get_window("hamming", L, fftbins=False), noverlap=L // 2,
                   nfft=max(256, 1 << int(np.ceil(np.log2(L)))), detrend=False, scaling="density")
    assert np.allclose(ff, f2) and np.allclose(psd, p2, rtol=1e-10)
    # white noise of std 0.3 -> sn ~ 0.3
    rng = np.random.default_rng(0)
    assert abs(oo.GetSn(0.3 * rng.standard_normal(20000)) - 0.3) < 0.01


def test_oasis_solves_the_nonnegative_deconvolution_problem():
    from scipy.optimize import nnls
    y, _, _ = _trace(120, seed=3, rate=0.05)
    y = y - 0.5
    g = 0.9
    c, s, pools = oo.oasisAR1(y, g, 0.0, 0.0)
    # min |K s - y|^2, s >= 0 with K the AR(1) impulse-response matrix (first sample free-signed in OASIS: c_1 >= 0)
    T = y.size
    Kmat = np.tril(g ** (np.arange(T)[:, None] - np.arange(T)[None, :]).clip(min=0))
    s_ref, _ = nnls(Kmat, y)
    c_ref = Kmat @ s_ref
    assert np.allclose(c, c_ref, atol=1e-8)
    assert np.all(c >= 0) and np.all(s >= -1e-12)
    assert sum(int(p[3]) for p in pools) == T
    # warm start from the final pools reproduces the solution
    c2, s2, _ = oo.oasisAR1(y, g, 0.0, 0.0, pools)
    assert np.allclose(c2, c)


def test_oasis_smin_gives_sparser_spikes():
    y, _, _ = _trace(800, seed=5)
    c0, s0, _ = oo.oasisAR1(y - 0.5, 0.95, 0.0, 0.0)
    c1, s1, _ = oo.oasisAR1(y - 0.5, 0.95, 0.0, 0.6)
    assert (s1 > 0).sum() < (s0 > 0).sum()
    assert np.all(s1[s1 > 0] >= 0.6 - 1e-9)


def test_fminbnd_matches_scipy():
    from scipy.optimize import fminbound
    f = lambda x: (x - 0.7312) ** 2 + 0.1 * np.sin(9 * x)
    x, last = oo.fminbnd(f, 0.0, 1.0, tol=1e-5)
    x2 = fminbound(f, 0.0, 1.0, xtol=1e-5)
    assert abs(x - x2) < 2e-5
    assert 0 <= last <= 1


def test_foopsi_recovers_time_constant_and_baseline():
    y, c_true, s_true = _trace(3000, g=0.95, sn=0.3, seed=13, amp=4.0)      # events well above smin = 5*sn
    sn = oo.GetSn(y)
    assert 0.27 < sn < 0.45          # spike onsets leak some power into the 0.25-0.5 band
    c, s, b, g = oo.deconvolveCa_ar1_foopsi(y, sn, None, maxIter=10)
    assert abs(g - 0.95) < 0.02
    assert abs(b - 0.5) < 0.1
    assert np.corrcoef(c, c_true)[0, 1] > 0.95
    # detected events line up with the true spikes (within one frame)
    det = np.nonzero(s > 0)[0]
    tru = np.nonzero(s_true > 0)[0]
    hits = sum(np.min(np.abs(tru - d)) <= 1 for d in det)
    assert hits >= 0.9 * len(det) and len(det) >= 0.7 * len(tru)


def test_estimate_time_constant_edge_cases():
    rng = np.random.default_rng(1)
    y = rng.standard_normal(2000)                  # white noise: g ~ 0 or slightly negative -> 0.15 rule or small g
    g = oo.estimate_time_constant_ar1(y, 0.0)
    assert g is not None and 0 <= g <= 0.2


def test_fminbnd_second_step_is_golden_not_parabolic():
    """MATLAB's fminbnd (Forsythe-Malcolm-Moler) enters the parabolic branch on its second step with v == w and fv == fw: the fit is degenerate,
    p = q = 0 EXACTLY, `abs(p) < abs(0.5*q*r)` is false and the step falls through to golden section.  The engine's copy of the loop must be
    compiled without FMA contraction for the same to hold (csrc/deconv.hip; with contraction it took an arbitrary 'parabolic' step there and
    gamma came out up to 2e-3 away from the reference's) -- this pins the evaluation sequence the GPU tests compare against."""
    import oasis_oracle as oo
    xs = []
    def f(x):
        xs.append(x); return (x - 0.93) ** 2 + 0.1 * np.sin(40 * x)
    oo.fminbnd(f, 0.0, 1.0)
    g = 0.5 * (3.0 - np.sqrt(5.0))
    assert abs(xs[0] - g) < 1e-15 and abs(xs[1] - (g + g * (1 - g))) < 1e-15
    assert abs(xs[2] - (xs[1] + g * (1 - xs[1]))) < 1e-15            # golden section of [x1, 1], not a parabola through two coincident points
