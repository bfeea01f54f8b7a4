"""R1-only driver for PMC passes: python scripts/r1_only.py --cfg c3 --variant 2 [--ac]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--variant", type=int, default=2); ap.add_argument("--ac", action="store_true")
ap.add_argument("--reps", type=int, default=2); ap.add_argument("--order", type=int, default=1); ap.add_argument("--delta", action="store_true", help="time the incremental path: alternate no footprints / all footprints"); ap.add_argument("--probe", type=int, default=0)
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
eng.set_option("r1_variant", a.variant); eng.set_option("r1_delta", 1 if a.delta else 0); eng.set_option("r1_probe", a.probe)
eng.profile(True)
A_b = f.A_init.astype(np.float32) if a.ac else None
if a.delta:
    A_all = f.A_init.astype(np.float32)
    eng.residual(0, None, None)
    for _ in range(a.reps):
        eng.residual(0, A_all, f.C_init); eng.residual(0, None, None)
for _ in range(0 if a.delta else a.reps):
    eng.residual(0, A_b, f.C_init if a.ac else None)
print(eng.profile_table())
