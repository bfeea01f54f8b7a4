"""TEST INFRASTRUCTURE: float64 oracle trajectories for the GPU parity tests at non-toy sizes.

The oracle needs ~13 s per iteration at 128 x 128 x 3000, so the variants (spatial algorithm, bg_ssub, deconvolution, crops of the BASELINE
configurations) run concurrently in spawned worker processes while the GPU tests go on (tests/conftest.py: `oracle_jobs`); a test asks for
its trajectory by name and compares the engine's run on the SAME seeded inputs with it.  Nothing here is imported by the product."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import numpy as np
import scipy.sparse as sp

# name -> configuration.  `crop` = (r0, c0, size): a window of the full FOV of the named BASELINE configuration (the same video, cropped).
JOBS = {
    # 128 x 128 x 3000, K = 31, r = 15: the size bench.py times the oracle on (one iteration = 13 s)
    "m128_hals": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="hals", iters=2),
    "m128_hals_thresh": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="hals_thresh", iters=2),
    "m128_nnls": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="nnls", iters=2),
    "m128_ssub2": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="hals", bg_ssub=2, iters=2),
    "m128_deconv": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="hals", deconv=True, iters=2),
    "m128_2x2": dict(d1=128, d2=128, T=3000, K=31, r=15, seed=9, alg="hals", patch=[64, 64], iters=2),
    "m96_r18": dict(d1=96, d2=96, T=1500, K=16, r=18, seed=11, alg="hals", iters=2),
    # outlier branch of fit_ring_model (:50-67; dead in every demo): with frame selection (24 ring offsets -> nmax = 2400 < T), without
    # (nmax >= T), and through the low-resolution fit of bg_ssub = 2.  The first two meet a few fp32-flipped `>` decisions (see the test): one
    # iteration only, a second one starts from weights that already differ by ~thresh/T and decides differently in many more places
    "m64_outlier": dict(d1=64, d2=64, T=3000, K=8, r=4, seed=13, alg="hals", thresh_outlier=3.0, iters=1),
    "m64_outlier_all": dict(d1=64, d2=64, T=1200, K=8, r=7, seed=14, alg="hals", thresh_outlier=2.5, iters=1),
    "m64_outlier_ssub2": dict(d1=64, d2=64, T=3000, K=8, r=8, seed=15, alg="hals", bg_ssub=2, thresh_outlier=3.0, iters=2),
    # the shipped demo's parameter set as ONE trajectory (demos/demo_large_data_1p.m:16-55,142-201): ring_radius = 18 with bg_ssub = 2 (a low-resolution ring of
    # radius 9), spatial_algorithm = 'hals_thresh' on the P.sn that update_sn = true just re-estimated, deconv_flag = true, two temporal updates, then the
    # switch to 'nnls' and a second background / spatial / temporal round
    "m96_demo": dict(d1=96, d2=96, T=1500, K=16, r=18, seed=11, alg="hals_thresh", bg_ssub=2, deconv=True, demo=True, iters=2),
    # BASELINE configs[1] (C2: 256 x 256 x 3000, K = 200, seed 1): a 64 x 64 window of the same video
    "c2_crop64": dict(d1=256, d2=256, T=3000, K=200, r=15, seed=1, alg="hals", crop=(96, 96, 64), iters=2),
}
NROWS = 384          # sampled patch pixels whose rows of the background-subtracted video are compared


def make_inputs(cfg):
    """(Y (T, d) float32, A_init csc float32, C_init float32, sn, d1, d2) -- identical on the engine side and in the worker"""
    from cnmf_e_amd import synth
    f = synth.make_factors(cfg["d1"], cfg["d2"], cfg["T"], cfg["K"], cfg["seed"])
    Y = synth.make_video(f, np.float32)
    A, C, sn, d1, d2 = sp.csc_matrix(f.A_init.astype(np.float32)), np.ascontiguousarray(f.C_init, dtype=np.float32), np.asarray(f.sn, dtype=np.float32), cfg["d1"], cfg["d2"]
    if cfg.get("crop"):
        r0, c0, n = cfg["crop"]
        pix = (np.arange(c0, c0 + n)[None, :] * d1 + np.arange(r0, r0 + n)[:, None]).reshape(-1, order="F")
        Y = np.ascontiguousarray(Y[:, pix])
        A = sp.csc_matrix(A.tocsr()[pix])
        keep = np.nonzero(np.asarray(A.sum(axis=0)).ravel() > 0)[0]
        A, C, sn, d1, d2 = sp.csc_matrix(A[:, keep]), np.ascontiguousarray(C[keep]), sn[pix], n, n
    return Y, A, C, sn, d1, d2


def sample_rows(cfg, npix):
    return np.sort(np.random.default_rng(1234).choice(npix, size=min(NROWS, npix), replace=False))


def trajectory(name):
    """run the oracle for JOBS[name]; everything the GPU test compares, per iteration"""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(16)
    except Exception:
        pass
    import cnmfe_oracle as orc
    cfg = JOBS[name]
    Y, A, C, sn, d1, d2 = make_inputs(cfg)
    T = cfg["T"]
    patch = cfg.get("patch") or [d1, d2]
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, patch, cfg["r"], A, C, sn, spatial_algorithm=cfg["alg"], maxIter=5,
                            bg_ssub=cfg.get("bg_ssub", 1), deconv_options={} if cfg.get("deconv") else None,
                            thresh_outlier=cfg.get("thresh_outlier", np.nan))
    out = {}
    if cfg.get("demo"):
        return demo_sequence(o, out, oracle=True)
    first = o._patches()[0]
    p0 = o.patch_pos[first]
    rows = sample_rows(cfg, int((p0[1] - p0[0] + 1) * (p0[3] - p0[2] + 1)))
    for it in range(cfg["iters"]):
        o.update_background_parallel()
        for idx in o._patches():
            Wc = sp.csr_matrix(o.W[idx]); Wc.sort_indices()
            out["W_%d_%d_%d" % (it, idx[0], idx[1])] = Wc.data.copy()
            out["b0_%d_%d_%d" % (it, idx[0], idx[1])] = np.asarray(o.b0[idx]).copy()
        out["resid_%d" % it] = o.init_residual(first)[rows]              # Y - A C - ring background on sampled pixels of the first patch (R1)
        out["C_before_spatial_%d" % it] = o.C.copy()
        o.update_spatial_parallel()
        out["A_raw_%d" % it] = sp.csc_matrix(o.A_raw)
        out["A_%d" % it] = sp.csc_matrix(o.A)
        out["b0new_s_%d" % it] = np.asarray(o.b0_new).copy()
        o.update_temporal_parallel()
        out["C_%d" % it] = o.C.copy(); out["C_raw_%d" % it] = o.C_raw.copy()
        out["b0new_t_%d" % it] = np.asarray(o.b0_new).copy()
        if cfg.get("deconv"):
            out["S_%d" % it] = o.S.copy(); out["kp_%d" % it] = np.asarray(o.kernel_pars, dtype=np.float64)
    return out


def demo_sequence(o, out, oracle):
    """demos/demo_large_data_1p.m:142,161-182,189,199-201 on either side (the oracle's OracleSources2D or the engine's Sources2D -- the same method names):
    background; spatial with update_sn; temporal x 2; spatial_algorithm = 'nnls'; background; spatial; temporal.  Records what the test compares."""
    def snap(tag, what):
        for w in what:
            if w == "W":
                if oracle:
                    Wc = sp.csr_matrix(o.W[(0, 0)]); Wc.sort_indices(); out["W_" + tag] = Wc.data.copy(); out["b0_" + tag] = np.asarray(o.b0[(0, 0)]).copy()
                else:
                    out["W_" + tag] = o.get_W((0, 0)).data.copy(); out["b0_" + tag] = o.get_b0((0, 0)).copy()
            elif w == "A":
                out["A_raw_" + tag] = sp.csc_matrix(o.A_raw).copy(); out["A_" + tag] = sp.csc_matrix(o.A).copy()
            elif w == "sn":
                out["sn_" + tag] = np.asarray(o.sn if oracle else o.P["sn"], dtype=np.float64).reshape(-1, order="F").copy()
            elif w == "C":
                out["C_" + tag] = np.asarray(o.C).copy(); out["C_raw_" + tag] = np.asarray(o.C_raw).copy(); out["S_" + tag] = np.asarray(o.S).copy()
                out["kp_" + tag] = np.asarray(o.kernel_pars if oracle else o.P["kernel_pars"], dtype=np.float64).copy()
    o.update_background_parallel(); snap("bg0", ["W"])                            # :142
    out["C_before_spatial_0"] = np.asarray(o.C).copy()
    o.update_spatial_parallel(update_sn=True); snap("sp0", ["A", "sn"])           # :161-163
    o.update_temporal_parallel(); snap("t0a", ["C"])                              # :172 (m = 1)
    o.update_temporal_parallel(); snap("t0b", ["C"])                              # :172 (m = 2)
    if oracle:
        o.spatial_algorithm = "nnls"                                              # :182
    else:
        o.options.spatial_algorithm = "nnls"
    o.update_background_parallel(); snap("bg1", ["W"])                            # :199
    o.update_spatial_parallel(); snap("sp1", ["A"])                               # :200
    o.update_temporal_parallel(); snap("t1", ["C"])                               # :201
    return out


if __name__ == "__main__":
    import time
    for n in sys.argv[1:] or ["m128_hals"]:
        t0 = time.time(); r = trajectory(n); print(n, "%.1f s" % (time.time() - t0), sorted(r)[:6])
