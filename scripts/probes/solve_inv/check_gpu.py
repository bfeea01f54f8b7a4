"""The ring solve out of cached inverses (option solve_inv, ring_solve_inv.hpp) against the factorising kernel on the same sequence of fits:
    python scripts/probes/solve_inv/check_gpu.py --cfg small|c2|c3 [--probe 512]
Three fits per mode (footprints / traces moving from the perturbed start to the truth, so that the ridge drifts between them); prints max |dW| / max |W| per fit,
the kernels' times and the fast path's statistics (pixels left over, ridge-series terms)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="small"); ap.add_argument("--probe", type=int, default=512); ap.add_argument("--radius", type=int, default=0); ap.add_argument("--modes", default="0,2,1")
ap.add_argument("--terms", type=int, default=5)
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
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.profile(True)
A0, A1 = f.A_init.astype(np.float32), f.A_true.astype(np.float32)
C0, C1 = f.C_init, f.C_true
seq = [(A0, C0), ((0.7 * A0 + 0.3 * A1).tocsc().astype(np.float32), (0.7 * C0 + 0.3 * C1).astype(np.float32)), ((0.65 * A0 + 0.35 * A1).tocsc().astype(np.float32), (0.65 * C0 + 0.35 * C1).astype(np.float32)), (A1, C1)]
Ws = {}
for mode in [int(x) for x in a.modes.split(",")]:
    eng.set_option("solve_inv", mode); eng.set_option("solve_probe", a.probe if mode else 0); eng.set_option("solve_inv_terms", a.terms)
    eng.ring_init(0, r)
    Ws[mode] = []
    for i, (A, C) in enumerate(seq):
        eng.profile_reset()
        _, info = eng.fit_ring_model(0, A, C)
        eng.synchronize()
        tab = eng.profile_table()
        ts = {k: round(v["total_ms"] / v["calls"], 3) for k, v in tab.items() if k.startswith("bg_ring") and v["calls"]}
        W = eng.ring_csr(0)
        Ws[mode].append(W.data.copy()); Wlast = W
        st = eng.ring_solve_stats(0) if mode else None
        print("mode %d fit %d: %s  active %d  nan %d  %s" % (mode, i, ts, info["n_active"], int(np.isnan(W.data).sum()), st), flush=True)
ks = list(Ws)
for k in ks[1:]:
    for i in range(len(seq)):
        dW = np.abs(Ws[k][i] - Ws[ks[0]][i]); sc = np.abs(Ws[ks[0]][i]).max()
        print("fit %d: max |W_mode%d - W_mode%d| / max|W| = %.3e   (rms %.3e, nan %d)" % (i, k, ks[0], np.nanmax(dW) / sc, np.sqrt(np.nanmean(dW ** 2)) / sc, int(np.isnan(Ws[k][i]).sum())), flush=True)
        if np.nanmax(dW) / sc > 1e-5:                      # the worst pixels: what is around their rings
            rowmax = np.maximum.reduceat(dW, Wlast.indptr[:-1][np.diff(Wlast.indptr) > 0])
            rows = np.flatnonzero(np.diff(Wlast.indptr) > 0)
            A = seq[i][0].tocsr()
            for j in np.argsort(-rowmax)[:4]:
                m = rows[j]; ring = Wlast.indices[Wlast.indptr[m]:Wlast.indptr[m + 1]]
                sub = A[ring]; ks_ = np.unique(sub.indices)
                cen = A[m].indices
                print("   pixel %d (r %d, c %d): err %.2e; neurons on the ring %s with %s ring pixels each (values %s); under the centre %s" % (
                    m, m % d1, m // d1, rowmax[j] / sc, ks_.tolist(), [int((sub[:, k] != 0).sum()) for k in ks_], [np.round(sub[:, k].data, 6).tolist()[:3] for k in ks_], cen.tolist()), flush=True)
