"""bg_ssub > 1, sweep-free (option ssub_virtual = 2: forced; 1, the default, decides by patch size; vproj_*_ssub) against the swept residual (0) on one engine build: A, C, W after full iterations, and which kernels ran.
    python scripts/ssub_virtual_check.py [--cfg small|mid|c3] [--ssub 2] [--iters 2] [--pdims 64,64]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="small"); ap.add_argument("--ssub", type=int, default=2); ap.add_argument("--iters", type=int, default=2); ap.add_argument("--pdims", default="")
ap.add_argument("--alg", default="hals")
a = ap.parse_args()
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
CFG = {"small": (96, 80, 400, 12, 10, 3), "mid": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2), "odd": (75, 66, 203, 9, 18, 5)}
d1, d2, T, K, r, seed = CFG[a.cfg]
f = synth.make_factors(d1, d2, T, K, seed)
Y = synth.make_video(f, np.float32)
pd = [int(x) for x in a.pdims.split(",")] if a.pdims else [d1, d2]
res = {}
for virt in (1, 0):
    eng = Engine(0)
    eng.set_option("ssub_virtual", 2 if virt else 0)
    video = PatchedVideo(d1, d2, T, pd, r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=a.alg, maxIter=3, bg_ssub=a.ssub), f.A_init, f.C_init, f.sn)
    eng.profile(True)
    ts = []
    for it in range(a.iters):
        eng.synchronize(); t0 = time.perf_counter()
        s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    tab = eng.profile_table()
    names = sorted((k for k, v in tab.items() if v["calls"]), key=lambda k: -tab[k]["total_ms"])
    print("ssub_virtual %d: iterations %s ms" % (virt, " ".join("%.2f" % t for t in ts)))
    print("   " + ", ".join("%s %.2f" % (k, tab[k]["total_ms"]) for k in names[:14]))
    res[virt] = (s.A.toarray().astype(np.float64), np.asarray(s.C, dtype=np.float64).copy())
    eng.close()
rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))
print("A: rel %.3e (same support: %s)   C: rel %.3e" % (rel(res[1][0], res[0][0]), np.array_equal(res[1][0] != 0, res[0][0] != 0), rel(res[1][1], res[0][1])))
