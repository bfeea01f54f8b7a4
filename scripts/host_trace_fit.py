"""Host-side phases of the fit / spatial / temporal calls at the headline size (option host_trace): python scripts/host_trace_fit.py [--patch 128]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--patch", type=int, default=512)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [a.patch, a.patch], r, eng)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
s = Sources2D(video, Options(ring_radius=r), f.A_init, f.C_init, f.sn)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize(); eng.set_option("host_trace", 1)
    s.update_background_parallel()
    s.update_spatial_parallel(); s.update_temporal_parallel()
    if it == 2:
        eng.set_option("host_trace", 0)
torch.cuda.synchronize()
