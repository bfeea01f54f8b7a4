"""Regenerate tests/golden/*.npz:  python tests/golden/make_golden.py

These are NOT outputs of the MATLAB reference (no MATLAB/Octave in the build image, no golden vectors in the reference's
own tests -- SURVEY.md section 8(c)); they are inputs + outputs of the float64 restatement in oracle/ on small seeded
cases, frozen so that (a) the oracle itself cannot drift unnoticed (tests/test_golden.py, CPU) and (b) the HIP engine is
compared against committed numbers, not only against whatever the oracle computes today (tests/test_gpu_parity.py, GPU).
Parity with MATLAB remains unpinned and is stated as such in DESIGN.md."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cnmfe_oracle as orc
import oasis_oracle as oo
from cnmf_e_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def iteration_case():
    d1, d2, T, K, r = 36, 32, 160, 5, 5
    f = synth.make_factors(d1, d2, T, K, 101, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [18, 16], r, f.A_init.astype(np.float32), f.C_init, f.sn,
                            spatial_algorithm="hals", maxIter=3)
    out = dict(d1=d1, d2=d2, T=T, K=K, r=r, patch=np.array([18, 16]), Y=Y, A_init=f.A_init.toarray().astype(np.float32),
               C_init=f.C_init.astype(np.float32), sn=np.asarray(f.sn, np.float32))
    o.update_background_parallel()
    for idx in sorted(o.W):
        out["W_%d_%d" % idx] = np.asarray(o.W[idx].todense() if hasattr(o.W[idx], "todense") else o.W[idx], dtype=np.float64)
        out["b0_%d_%d" % idx] = np.asarray(o.b0[idx], dtype=np.float64)
    o.update_spatial_parallel()
    out["A_after_spatial"] = o.A.toarray()
    o.update_temporal_parallel()
    out["C_after_temporal"] = o.C
    out["b0_new"] = o.b0_new
    np.savez_compressed(os.path.join(HERE, "iteration_2x2.npz"), **out)


def oasis_case():
    rng = np.random.default_rng(77)
    T, g = 400, 0.93
    s = (rng.random((3, T)) < 0.02) * (1 + rng.random((3, T)))
    c = np.zeros((3, T))
    for t in range(T):
        c[:, t] = (g * c[:, t - 1] if t else 0) + s[:, t]
    y = c * 4 + 0.3 * rng.standard_normal((3, T)) + np.array([[0.5], [1.5], [0.0]])
    out = dict(y=y.astype(np.float32))
    sn = np.array([oo.GetSn(row) for row in y.astype(np.float32).astype(np.float64)])
    out["sn"] = sn
    res = [oo.deconvolveCa_ar1_foopsi(row.astype(np.float32).astype(np.float64), sn_, smin=-5.0, optimize_pars=True, optimize_b=True)
           for row, sn_ in zip(y, sn)]
    out["c"] = np.array([r_[0] for r_ in res]); out["s"] = np.array([r_[1] for r_ in res])
    out["b"] = np.array([r_[2] for r_ in res]); out["g"] = np.array([r_[3] for r_ in res])
    np.savez_compressed(os.path.join(HERE, "oasis_ar1.npz"), **out)


if __name__ == "__main__":
    iteration_case()
    oasis_case()
    print("written", os.listdir(HERE))
