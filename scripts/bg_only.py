"""background-fit-only driver for PMC passes: python scripts/bg_only.py --cfg c3 [--mode 2]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--mode", type=int, default=2); ap.add_argument("--reps", type=int, default=1); ap.add_argument("--kernels", default="3,2"); ap.add_argument("--probe", type=int, default=0); ap.add_argument("--sprobes", default="0"); ap.add_argument("--smode", type=int, default=2)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2)}
d1, d2, T, K, r, seed = CFG[a.cfg]
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.ring_init(0, r)
eng.set_option("solve_mode", a.smode); eng.set_option("gram_mode", a.mode); eng.set_option("gram_probe", a.probe)
eng.profile(True)
Ws = {}
for sp_ in [int(x) for x in a.sprobes.split(",")]:
  for gk in [int(x) for x in a.kernels.split(",")]:
    eng.set_option("gram_kernel", gk); eng.set_option("solve_probe", sp_); eng.profile_reset()
    for _ in range(a.reps):
        eng.fit_ring_model(0, f.A_init.astype(np.float32), f.C_init)
    tab = eng.profile_table()
    print("solve_probe %d gram_kernel %d:" % (sp_, gk), {k: round(v["total_ms"] / v["calls"], 3) for k, v in tab.items() if k.startswith("bg_") and v["calls"]}, flush=True)
    Ws[gk] = eng.ring_csr(0).data.copy()
    if sp_: eng.ring_init(0, r)
ks = list(Ws)
for k in ks[1:]:
    print("max |W_%d - W_%d| / max|W| = %.3e" % (k, ks[0], np.abs(Ws[k] - Ws[ks[0]]).max() / np.abs(Ws[ks[0]]).max()))
