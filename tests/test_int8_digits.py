"""CPU check of the arithmetic behind cnmf_e_amd/csrc/gram_i8.hpp / win_proj_i8.hpp (no GPU, no library): 32-bit fixed-point scaling, balanced base-256 digits, the
digit-pair weight classes that are kept (p + r >= 2) -- the same integer formulation the kernels execute on v_mfma_i32_16x16x64_i8, in NumPy int64."""
import numpy as np


def digits_of(X):
    """rows of X -> (four digit planes d0..d3 in [-128, 127], per-row scale): X ~ s * (d0 + 256 d1 + 256^2 d2 + 256^3 d3)"""
    m = np.abs(X).max(axis=1, keepdims=True)
    s = np.where(m > 0, m / (2.0 ** 31 - 2.0 ** 24), 1.0)
    q = np.rint(X / s).astype(np.int64)
    planes = []
    for _ in range(4):
        d = ((q + 128) & 255) - 128
        q = (q - d) >> 8
        planes.append(d)
    assert np.all(q == 0)
    return planes, s


def test_digits_reconstruct_the_fixed_point_value_and_fit_int8():
    rng = np.random.default_rng(0)
    X = (rng.standard_normal((50, 333)) * rng.uniform(1e-3, 1e3, (50, 1))).astype(np.float32).astype(np.float64)
    X[7] = 0.0                                                   # an all-zero row: scale 1, digits 0
    X[8, 5] = -X[8].max() * 3                                    # the row maximum is negative
    D, s = digits_of(X)
    for d in D:
        assert d.min() >= -128 and d.max() <= 127
    q = sum((256 ** p) * D[p] for p in range(4))
    assert np.all(np.abs(q) <= 2 ** 31 - 2 ** 24)
    assert np.array_equal(q, np.rint(X / s).astype(np.int64))
    assert np.abs(q * s - X).max() <= 0.5 * s.max() * (1 + 1e-12)      # the quantisation step, nothing else


def test_kept_weight_classes_reproduce_the_gram_to_the_quantisation():
    """G = X X' from the digit-pair sums of classes p + r = 2 .. 6 (13 pairs, 5 accumulators) against float64, and the int32 range of a 64-frame step"""
    rng = np.random.default_rng(1)
    T = 640
    base = rng.standard_normal((1, T)) * 20.0                    # a shared slow component: the strongly correlated rows of a background video
    X = (base + rng.standard_normal((40, T))).astype(np.float32).astype(np.float64)
    X -= X.mean(axis=1, keepdims=True)
    X = X.astype(np.float32).astype(np.float64)
    D, s = digits_of(X)
    acc = {c: np.zeros((40, 40), dtype=np.int64) for c in range(2, 7)}
    worst = 0
    for t0 in range(0, T, 64):                                   # one MFMA step = 64 frames
        for p in range(4):
            for r in range(4):
                if p + r >= 2:
                    part = D[p][:, t0:t0 + 64] @ D[r][:, t0:t0 + 64].T
                    acc[p + r] += part
        worst = max(worst, max(int(np.abs(a).max()) for a in acc.values()))
    assert worst < 2 ** 31                                        # exact in int32 at this length (4 pairs x 64 frames x 2^14 per step: <= 24576 frames in general)
    G = sum((256.0 ** c) * acc[c].astype(np.float64) for c in acc) * (s @ s.T)
    Gref = X @ X.T
    assert np.abs(G - Gref).max() <= 3e-9 * np.abs(np.diag(Gref)).max()
    # the ridge system of fit_ring_model.m:103-106 from both: W to ~1e-8
    n = 39
    def solve(Gm):
        A = Gm[:n, :n]; g = Gm[:n, n]
        return np.linalg.solve(A + 1e-5 * np.trace(A) * np.eye(n), g)
    w, wref = solve(G), solve(Gref)
    assert np.abs(w - wref).max() <= 1e-6 * np.abs(wref).max()
