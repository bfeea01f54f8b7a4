"""Per-pixel GetSn of the resident residual at the headline size (cnmfe_get_sn, update_spatial_parallel.m:191-194).  python scripts/sn_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
s = Sources2D(video, Options(ring_radius=r), f.A_init, f.C_init, f.sn)
s.update_background_parallel(); s.update_spatial_parallel()   # (leaves the residual of (A_prev, C_prev) resident)
eng.profile(True)
for it in range(2):
    torch.cuda.synchronize(); eng.profile_reset(); t0 = time.perf_counter()
    sn = eng.get_sn(video.pid[video.owned[0]])
    dt = time.perf_counter() - t0
    tab = eng.profile_table()
    print("get_sn: %.1f ms wall  %s  median sn %.4f" % (dt * 1e3, {k: round(v["total_ms"], 2) for k, v in tab.items() if v["total_ms"] > 0.05}, float(np.median(sn))), flush=True)
