"""Times the two video passes of the iteration (bg_win_proj, temporal_proj_B) on the headline problem with whatever library CNMFE_LIB names -- for probe builds
whose results are garbage (scripts/build_variant.py with -DCNMFE_PROBE_NOMFMA: the loads stay, the fp64 MFMAs go).  python scripts/video_pass_probe.py [--cfg c3]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", default="c3"); a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2)}
d1, d2, T, K, r, seed = CFG[a.cfg]
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
eng.profile(True)
for it in range(3):
    for name in ("update_background_parallel", "update_temporal_parallel"):
        try:
            getattr(s, name)()
            eng.synchronize()
        except Exception as e:
            print("(%s raised %s: expected of a probe build)" % (name, type(e).__name__))
tab = eng.profile_table()
for k in ("bg_win_proj", "temporal_proj_B", "bg_ring_solve"):
    if k in tab and tab[k]["calls"]:
        print("%-18s %.3f ms per call (%d calls)" % (k, tab[k]["total_ms"] / tab[k]["calls"], tab[k]["calls"]))
