"""R1 timing sweep over (variant, tile_order): python scripts/r1_sweep.py --variants 2,5,7 --orders 0,1 [--ac]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="c3"); ap.add_argument("--variants", default="2,5,7"); ap.add_argument("--orders", default="0,1")
ap.add_argument("--ac", action="store_true"); ap.add_argument("--reps", type=int, default=3)
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
A_b = f.A_init.astype(np.float32)
for v in [int(x) for x in a.variants.split(",")]:
    for o in [int(x) for x in a.orders.split(",")]:
        eng.set_option("r1_variant", v); eng.set_option("tile_order", o)
        for ac in ([False, True] if a.ac else [False]):
            eng.residual(0, A_b if ac else None, f.C_init if ac else None)
            eng.profile(True); eng.profile_reset(); eng.set_option("r1_delta", 0)   # variant timings: always the full ring sweep
            for _ in range(a.reps):
                eng.residual(0, A_b if ac else None, f.C_init if ac else None)
            tab = eng.profile_table()
            print("variant %d order %d ac %d : %s" % (v, o, ac, ["%s %.3f ms" % (n, e["total_ms"] / e["calls"]) for n, e in tab.items() if n.startswith("residual_r1")]), flush=True)
            eng.profile(False)
