#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-200
import sys, time, os, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0); eng.set_option("lanes", 3)
video = PatchedVideo(d1, d2, T, [128, 128], r, eng)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
for it in range(7):
    s.update_background_parallel()
    pr = cProfile.Profile(); pr.enable()
    s.update_spatial_parallel()
    pr.disable()
    s.update_temporal_parallel(); torch.cuda.synchronize()
    if it in (2, 5):
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(12)
        print("==== iteration", it); print("\n".join(l for l in st.getvalue().splitlines() if l.strip())[:3000])
PY
