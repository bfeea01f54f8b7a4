"""GPU parity at non-toy sizes and on the BASELINE configurations.

  * 128 x 128 x 3000, K = 31, r = 15 -- two FULL iterations of Sources2D.update_{background,spatial,temporal}_parallel against the float64
    oracle's method-level restatement, for spatial_algorithm hals / hals_thresh / nnls, bg_ssub = 2, deconv_flag = true, 2 x 2 patches, and
    ring radius 18 (block-pair table with displacement 3).  The second iteration is the non-first-run fit (ind_active, kept table).
  * C2 (256 x 256 x 3000, K = 200): oracle on a 64 x 64 window of the same video; properties at full size.
The oracle trajectories come from worker processes (tests/oracle_jobs.py, conftest.oracle_jobs).  File name: runs after the other GPU tests."""
import numpy as np
import scipy.sparse as sp
import pytest

import oracle_jobs as oj

pytestmark = pytest.mark.gpu


from parity_util import rel


@pytest.fixture(scope="module")
def eng():
    from cnmf_e_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _support_report(Ag, Ar, thr=None):
    """(entries present in exactly one of the two, how many of them sit off the decision threshold, rel error on the common support)"""
    Ag, Ar = np.asarray(Ag.todense()), np.asarray(Ar.todense())
    ng, nr = Ag != 0, Ar != 0
    mism = ng != nr
    val = np.where(ng, Ag, Ar)[mism]                                     # the surviving value of a mismatched entry
    scale = np.maximum(np.abs(Ar).max(axis=0, keepdims=True), 1e-30) * np.ones_like(Ar)
    if thr is None:
        off = np.abs(val) > 1e-5 * scale[mism]                           # hals / nnls decide at 0: a mismatch must be a value that is ~0
    else:
        off = np.abs(val - thr[mism]) > 1e-4 * np.abs(thr[mism])         # hals_thresh decides at sn * 3 / sqrt(cc)
    both = ng & nr
    return int(mism.sum()), int(off.sum()), rel(Ag[both], Ar[both]), int(nr.sum())


@pytest.mark.parametrize("name", ["m128_hals", "m128_hals_thresh", "m128_nnls", "m128_ssub2", "m128_deconv", "m128_2x2", "m96_r18", "c2_crop64"])
def test_two_iterations_against_oracle(eng, oracle_jobs, observed, name):
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    cfg = oj.JOBS[name]
    Y, A, C, sn, d1, d2 = oj.make_inputs(cfg)
    T, r = cfg["T"], cfg["r"]
    video = PatchedVideo(d1, d2, T, cfg.get("patch") or [d1, d2], r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=cfg["alg"], maxIter=5, bg_ssub=cfg.get("bg_ssub", 1), deconv_flag=bool(cfg.get("deconv"))), A, C, sn)
    first = video.order[0]
    rows = oj.sample_rows(cfg, video.patch_pix[first].size)
    got = {}
    for it in range(cfg["iters"]):
        s.update_background_parallel()
        for idx in video.order:
            got["W_%d_%d_%d" % (it, idx[0], idx[1])] = s.get_W(idx).data.copy()
            got["b0_%d_%d_%d" % (it, idx[0], idx[1])] = s.get_b0(idx).copy()
        got["resid_%d" % it] = s.init_residual(first)[:, rows].T.copy()
        got["C_before_spatial_%d" % it] = np.asarray(s.C).copy()
        s.update_spatial_parallel()
        got["A_raw_%d" % it] = s.A_raw.copy(); got["A_%d" % it] = s.A.copy(); got["b0new_s_%d" % it] = np.asarray(s.b0_new).copy()
        s.update_temporal_parallel()
        got["C_%d" % it] = np.asarray(s.C).copy(); got["C_raw_%d" % it] = np.asarray(s.C_raw).copy(); got["b0new_t_%d" % it] = np.asarray(s.b0_new).copy()
        if cfg.get("deconv"):
            got["S_%d" % it] = np.asarray(s.S).copy(); got["kp_%d" % it] = np.asarray(s.P["kernel_pars"], dtype=np.float64)
    ref = oracle_jobs[name].result(timeout=900)
    obs = observed.setdefault(name, {})
    deconv = bool(cfg.get("deconv"))
    # Tolerances = <= 10x the errors observed on MI355X (gpurun_out/parity_observed.json), far inside SURVEY 8(c)'s
    # 1e-3 (W) / 1e-4 (Ysig, U, C).  With deconv_flag the first iteration's traces already differ by the discrete OASIS decisions (worst trace
    # 1.6e-3, median 5e-5), and everything the second iteration computes inherits that: `loose` scales its tolerances.
    for it in range(cfg["iters"]):
        loose = 1e4 if (deconv and it) else 1.0
        # ---- background ----
        for idx in video.order:
            k = "W_%d_%d_%d" % (it, idx[0], idx[1])
            e = rel(got[k], ref[k]); obs[k] = e
            assert got[k].shape == ref[k].shape and e <= 2e-6 * loose, (name, k, e)
            k = "b0_%d_%d_%d" % (it, idx[0], idx[1])
            e = float(np.abs(got[k] - ref[k]).max()); obs[k] = e
            assert e <= 5e-4 * min(loose, 100), (name, k, e)              # b0 ~ 1e3 stored in fp32: 6e-5 is one ulp
        # ---- R1 on sampled pixels of the first patch ----
        k = "resid_%d" % it
        e = rel(got[k], ref[k]); obs[k] = e
        assert e <= 5e-5 * min(loose, 100), (name, k, e)
        # ---- spatial: support exact off the decision threshold, values on the common support ----
        thr = None
        if cfg["alg"] == "hals_thresh":
            Cb = ref["C_before_spatial_%d" % it]
            cc = (Cb * Cb).sum(axis=1) - Cb.shape[1] * Cb.mean(axis=1) ** 2
            thr = np.asarray(s.P["sn"], dtype=np.float64)[:, None] * (3.0 / np.sqrt(cc))[None, :]
        n_m, n_off, e, nnz = _support_report(got["A_raw_%d" % it], ref["A_raw_%d" % it], thr)
        obs["A_raw_%d" % it] = dict(mismatch=n_m, off_threshold=n_off, rel=e, nnz=nnz)
        assert n_off == 0 and (n_m == 0 or loose > 1) and e <= 1e-6 * loose, (name, it, n_m, n_off, e)
        n_m, n_off, e, nnz = _support_report(got["A_%d" % it], ref["A_%d" % it])
        obs["A_%d" % it] = dict(mismatch=n_m, rel=e, nnz=nnz)
        assert (n_m == 0 or (loose > 1 and n_m <= 0.01 * nnz)) and e <= 1e-6 * loose, (name, it, n_m, e)
        e = float(np.abs(got["b0new_s_%d" % it] - ref["b0new_s_%d" % it]).max()); obs["b0new_s_%d" % it] = e
        assert e <= 1e-4 * loose, (name, it, e)
        # ---- temporal ----
        if not deconv:
            for k in ("C_%d" % it, "C_raw_%d" % it):
                e = rel(got[k], ref[k]); obs[k] = e
                assert e <= 5e-6, (name, k, e)
        else:
            # OASIS is a discrete active-set method: a pool boundary may move by a frame when a comparison falls within fp32 rounding
            er = [rel(got["C_raw_%d" % it][j], ref["C_raw_%d" % it][j]) for j in range(C.shape[0])]
            ec = [rel(got["C_%d" % it][j], ref["C_%d" % it][j]) for j in range(C.shape[0])]
            ns = [(int((got["S_%d" % it][j] > 0).sum()), int((ref["S_%d" % it][j] > 0).sum())) for j in range(C.shape[0])]
            obs["C_raw_%d" % it] = dict(max=max(er), median=float(np.median(er))); obs["C_%d" % it] = dict(max=max(ec), median=float(np.median(ec)))
            obs["spikes_%d" % it] = dict(max_count_diff=max(abs(a - b) for a, b in ns), total=(sum(a for a, _ in ns), sum(b for _, b in ns)))
            obs["kp_%d" % it] = float(np.abs(got["kp_%d" % it] - ref["kp_%d" % it]).max())
            tl = 10.0 if it else 1.0
            assert max(er) <= 1e-2 * tl and max(ec) <= 1.5e-2 * tl and np.median(ec) <= 5e-4 * tl, (name, it, max(er), max(ec), float(np.median(ec)))
            assert all(abs(a - b) <= (2 if not it else max(2, 0.05 * b)) for a, b in ns), (name, it, ns)
            assert obs["kp_%d" % it] <= 4e-4 * tl
        e = float(np.abs(got["b0new_t_%d" % it] - ref["b0new_t_%d" % it]).max()); obs["b0new_t_%d" % it] = e
        assert e <= (0.1 if deconv else 1e-4), (name, it, e)
