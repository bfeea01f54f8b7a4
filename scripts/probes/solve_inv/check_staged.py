"""k_ring_solve8 (ring_solve_staged.hpp: the footprints sampled from per-neuron windows) against k_ring_solve6 on the same sequence of fits: the weights must be
bit-identical.   python scripts/probes/solve_inv/check_staged.py --cfg small|edge|c2|c3 [--radius r]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="small"); ap.add_argument("--radius", type=int, default=0); ap.add_argument("--probe", type=int, default=0)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2), "small": (128, 128, 1000, 30, 15, 4), "edge": (96, 80, 600, 12, 15, 7)}
d1, d2, T, K, r, seed = CFG[a.cfg]
r = a.radius or r
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); Yh = Yd.cpu().numpy().astype(np.float64) if a.probe & 4096 else None; del Yd; torch.cuda.empty_cache()
eng.profile(True)
A0, A1 = f.A_init.astype(np.float32), f.A_true.astype(np.float32)
seq = [(A0, f.C_init), ((0.7 * A0 + 0.3 * A1).tocsc().astype(np.float32), (0.7 * f.C_init + 0.3 * f.C_true).astype(np.float32)), (A1, f.C_true)]
Ws = {}
eng.set_option("solve_inv", 0); eng.set_option("solve_probe", a.probe)
for mode in [int(x) for x in os.environ.get("STAGED_MODES", "0,1").split(",")]:
    eng.set_option("solve_staged", mode)
    eng.ring_init(0, r)
    Ws[mode] = []
    for i, (A, C) in enumerate(seq):
        eng.profile_reset()
        _, info = eng.fit_ring_model(0, A, C)
        eng.synchronize()
        tab = eng.profile_table()
        ts = {k: round(v["total_ms"] / v["calls"], 3) for k, v in tab.items() if (k.startswith("bg_ring") or k.startswith("bg_neuron")) and v["calls"]}
        Wc = eng.ring_csr(0); Ws[mode].append(Wc.data.copy())
        print("solve_staged %d fit %d: %s  active %d" % (mode, i, ts, info["n_active"]), flush=True)
MA = max(Ws)
for i in range(len(seq)):
    d = Ws[MA][i].view(np.uint32) != Ws[0][i].view(np.uint32)
    print("fit %d: weights that differ in any bit: %d of %d;  max |dW| / max |W| = %.2e" % (i, int(d.sum()), d.size, np.abs(Ws[MA][i] - Ws[0][i]).max() / np.abs(Ws[0][i]).max()), flush=True)
    nan = np.isnan(Ws[MA][i]); dd = np.abs(Ws[MA][i] - Ws[0][i]); dd[nan] = 0
    rows = np.repeat(np.arange(Wc.shape[0]), np.diff(Wc.indptr))
    print("   NaN weights %d in %d pixels; without them max |dW| / max |W| = %.2e; pixels with a difference > 1e-6: %d; first NaN pixels %s" % (
        int(nan.sum()), np.unique(rows[nan]).size, dd.max() / np.abs(Ws[0][i]).max(), np.unique(rows[dd > 1e-6 * np.abs(Ws[0][i]).max()]).size, np.unique(rows[nan])[:8].tolist()), flush=True)

if a.probe & 2048:                                          # the staged A against the footprints' row sums on every ring
    import scipy.sparse as sp
    Wd = Wc.tocsr(); A = seq[-1][0].tocsr(); rs_ = np.asarray(A.sum(1)).ravel()
    bad = 0
    for m in range(0, Wd.shape[0], 97):
        cols = Wd.indices[Wd.indptr[m]:Wd.indptr[m + 1]]; vals = Ws[1][-1][Wd.indptr[m]:Wd.indptr[m + 1]]
        if cols.size != 96: continue
        exp = rs_[cols[:-1]]
        nlive = np.unique(np.concatenate([A[cols].indices, A[m].indices])).size
        code = int(round(vals[-1]))
        if code // 100 != nlive:
            bad += 1
            if bad <= 5: print("pixel", m, "live slots", code // 100, "candidates", code % 100, "neurons on ring or centre", nlive)
        if not np.allclose(vals[:-1], exp, rtol=1e-5, atol=1e-7):
            bad += 1
            if bad <= 3: print("pixel", m, "staged", vals[:6], "expected", exp[:6], "code", vals[-1])
    print("pixels checked with a wrong staged sum:", bad)

if a.probe & 4096:                                          # the staged U~ of both kernels against a float64 U~ = Yc Cc' - A (Cc Cc') / 2 of the last fit
    Yh = Yh if Yh.shape[0] == d1 * d2 else Yh.T
    A = seq[-1][0].tocsr(); C = seq[-1][1].astype(np.float64)
    Yc = Yh - Yh.mean(1, keepdims=True); Cc = C - C.mean(1, keepdims=True)
    Ut = Yc @ Cc.T - (A @ (Cc @ Cc.T)) / 2
    Wd = Wc.tocsr()
    for m in list(range(1000, Wd.shape[0], 1511))[:8]:
        cols = Wd.indices[Wd.indptr[m]:Wd.indptr[m + 1]]
        ks_ = np.unique(np.concatenate([A[cols].indices, A[m].indices]))
        exp = Ut[np.ix_(cols, ks_)].sum(1) if ks_.size else np.zeros(cols.size)
        v0 = Ws[0][-1][Wd.indptr[m]:Wd.indptr[m + 1]]; v1 = Ws[1][-1][Wd.indptr[m]:Wd.indptr[m + 1]]
        sc_ = max(1.0, np.abs(exp).max())
        print("pixel %d (%d neurons): solve6 err %.2e, solve8 err %.2e of %.2e" % (m, ks_.size, np.abs(v0 - exp).max() / sc_, np.abs(v1 - exp).max() / sc_, sc_))
