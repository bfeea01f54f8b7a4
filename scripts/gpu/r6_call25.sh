#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-300
import sys, time, os
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
log = []
orig = eng._pinned_take
def take(n):
    t0 = time.perf_counter(); p = orig(n); dt = 1e3 * (time.perf_counter() - t0)
    if dt > 0.2: log.append((n, round(dt, 2)))
    return p
eng._pinned_take = take
names = ["update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"]
for it in range(8):
    t0 = time.perf_counter(); parts = []
    for nm in names:
        t = time.perf_counter(); getattr(s, nm)(); parts.append(round(1e3 * (time.perf_counter() - t), 1))
    torch.cuda.synchronize()
    print("iteration %d: %.1f ms, host time in calls %s, slow pinned takes (bytes, ms): %s" % (it, 1e3 * (time.perf_counter() - t0), parts, log)); log.clear()
PY
