#!/bin/bash
# where the host's 20 ms of the FIRST temporal update (and 6 ms of the first spatial update) after an upload go: cProfile by internal time
cd $GRAFT_REPO_ROOT
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-190
import sys, time, os, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [512, 512], r, eng)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
torch.cuda.empty_cache()
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
torch.cuda.synchronize()
for nm in ["update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"]:
    pr = cProfile.Profile(); pr.enable(); getattr(s, nm)(); pr.disable(); torch.cuda.synchronize()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(9)
    print("====", nm); print("\n".join(l for l in st.getvalue().splitlines() if l.strip())[:2200])
PY
