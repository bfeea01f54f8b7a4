"""The bench's iteration (background + spatial + temporal, bench.py) with the ring solve out of cached inverses: per step the solve kernels' times and the
fast path's statistics -- how many ridge-series terms the steady-state iteration needs.   python scripts/probes/solve_inv/bench_loop.py --cfg c3 --steps 14 --mode 1"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--steps", type=int, default=14); ap.add_argument("--mode", type=int, default=1); ap.add_argument("--probe", type=int, default=512)
ap.add_argument("--terms", type=int, default=5)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2), "small": (128, 128, 1000, 30, 15, 4)}
d1, d2, T, K, r, seed = CFG[a.cfg]
f = synth.make_factors(d1, d2, T, K, seed)
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
eng.set_option("solve_inv", a.mode); eng.set_option("solve_probe", a.probe if a.mode else 0); eng.set_option("solve_inv_terms", a.terms)
eng.profile(True)
pid = video.pid[(0, 0)] if hasattr(video, "pid") else 0
for it in range(a.steps):
    eng.profile_reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
    eng.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    tab = eng.profile_table()
    ts = {k: round(v["total_ms"] / v["calls"], 3) for k, v in tab.items() if k.startswith("bg_ring") and v["calls"]}
    st = eng.ring_solve_stats(pid) if a.mode else None
    print("step %2d: %.2f ms  %s  %s" % (it, ms, ts, st), flush=True)
