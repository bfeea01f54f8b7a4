#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-900
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0); eng.set_option("lanes", int(os.environ.get("LANES", "3")))
video = PatchedVideo(d1, d2, T, [128, 128], r, eng)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
eng.profile(True)
prev = {}
for it in range(7):
    eng.profile_reset()
    t0 = time.perf_counter()
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    tab = {k: (v["calls"], round(v["total_ms"], 2)) for k, v in eng.profile_table().items() if v["calls"]}
    if it >= 1:
        diff = {k: (prev.get(k), v) for k, v in tab.items() if prev.get(k, (0, 0))[0] != v[0]}
        diff.update({k: (v, None) for k, v in prev.items() if k not in tab})
        print("iteration %d: %.1f ms; kernels whose call count changed against the previous iteration: %s" % (it, dt, diff))
    prev = tab
PY
