"""Where the time of ONE trace of a deconvolution level goes at the headline size (verdict r5 #3): a temporal update with deconv_flag = true on the c3 video,
option deconv_trace = 1000000 + k + 1 -- thread 0 of that trace's workgroup prints its phase times from the kernel (`DTT` lines: 100 MHz wall clock), for a few traces.
    python scripts/deconv_phases.py [--traces 1,100,300]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--traces", default="1,100,300"); ap.add_argument("--cfg", default="c3")
a = ap.parse_args()
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
s = Sources2D(video, Options(ring_radius=r, deconv_flag=True), f.A_init, f.C_init, f.sn)
for it in range(2):                                         # (the second iteration is the steady state: time constants known, pools warm)
    s.update_background_parallel(); s.update_spatial_parallel()
    if it == 1:
        eng.profile(True); eng.profile_reset()
    s.update_temporal_parallel(); eng.synchronize()
tab = eng.profile_table()
print({k: (round(v["total_ms"], 3), v["calls"]) for k, v in tab.items() if "deconv" in k}, flush=True)
for k in [int(x) for x in a.traces.split(",")]:
    print("---- trace %d ----" % k, flush=True)
    eng.set_option("deconv_trace", 1000000 + k + 1)
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel(); eng.synchronize()
eng.set_option("deconv_trace", 0)
print("(the DTT lines above: trace, phase id, microseconds -- 0 row update + load, 1 quantile + median, 2 GetSn, 3 time constant, 4 cold OASIS pass + tasks, "
      "5 b + Brent over g, 6 warm-started pass + tasks, 7 solution + outputs, 20 number of evaluations of Brent's objective (not a time), 99 total)")
