"""Phase probes of the ring-regression solve kernel on one patch: python scripts/solve_ab.py --cfg c3 [--probes 0,1,3,7]
(solve_probe bits: 1 no table loads, 2 no factorisation, 4 return before the substitutions, 8 no footprint corrections, 16 no refinement (fp32 solve only)).
Every run fits the same first-run problem (ring re-initialised before each fit).  --modes = values of the option solve_f32 (0: k_ring_solve6, fp64 tiles; 1:
k_ring_solve7, fp32 factorisation + fp64 refinement: round 6's experiment, scripts/probes/solve_r6/ -- without the patch the runs only repeat).  The weights of every mode are compared with the first's.  --refit: a second fit with the footprints in
(not a first run: only the active pixels are solved, the corrections apply)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--modes", default="0"); ap.add_argument("--reps", type=int, default=1); ap.add_argument("--probes", default="0"); ap.add_argument("--radius", type=int, default=0); ap.add_argument("--refit", type=int, default=0)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2), "small": (128, 128, 1000, 30, 15, 4)}
d1, d2, T, K, r, seed = CFG[a.cfg]
r = a.radius or r
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.profile(True)
Ws = {}
for mode in [int(x) for x in a.modes.split(",")]:
    for probe in [int(x) for x in a.probes.split(",")]:
        eng.ring_init(0, r)
        eng.set_option("solve_probe", probe)
        try: eng.set_option("solve_f32", mode)              # (only builds with scripts/probes/solve_r6/f32_refine.patch applied know the option)
        except Exception: print("(this build has no option solve_f32: mode %d only repeats the shipped kernel)" % mode, flush=True)
        ts = []
        for rep in range(a.reps):
            eng.ring_init(0, r); eng.profile_reset()
            _, info = eng.fit_ring_model(0, f.A_init.astype(np.float32), f.C_init)
            eng.synchronize()
            tab = eng.profile_table()
            ts.append(tab["bg_ring_solve"]["total_ms"] / tab["bg_ring_solve"]["calls"])
        print("mode %d probe %d: bg_ring_solve %s ms   (%s)" % (mode, probe, " ".join("%.3f" % t for t in ts), info), flush=True)
        if probe == 0:
            Ws[mode] = eng.ring_csr(0).data.copy()
        if probe & 256:                                     # k_ring_solve7's diagnostic rows: W(0, :) = refinement steps taken, W(1, :) = |d_1| / |x|
            Wd = eng.ring_csr(0)
            full = np.flatnonzero(np.diff(Wd.indptr) == info["pmax"])
            nit = Wd.data[Wd.indptr[full]]; eta = Wd.data[Wd.indptr[full] + 1]
            print("  refinement steps (pixels with a whole ring: %d): %s;  |d_1|/|x| quantiles 10/50/90/99/100%%: %s" % (
                full.size, dict(zip(*np.unique(nit, return_counts=True))), " ".join("%.2e" % q for q in np.quantile(eta, [0.1, 0.5, 0.9, 0.99, 1.0]))), flush=True)
ks = list(Ws)
for k in ks[1:]:
    dW = np.abs(Ws[k] - Ws[ks[0]])
    print("max |W_%d - W_%d| / max|W| = %.3e   (rms %.3e, nan %d, entries that differ in any bit %d of %d)" % (k, ks[0], np.nanmax(dW) / np.abs(Ws[ks[0]]).max(), np.sqrt(np.nanmean(dW ** 2)), int(np.isnan(Ws[k]).sum()), int((Ws[k].view(np.uint32) != Ws[ks[0]].view(np.uint32)).sum()), Ws[k].size))
