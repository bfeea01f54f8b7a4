#!/bin/bash
# the first iteration after an upload at the headline size: host-side phases of every call (host_trace = 1) + wall time per method
cd $GRAFT_REPO_ROOT
CNMFE_OPTS=host_trace=1 timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-200
import sys, time, os
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
    t = time.perf_counter(); getattr(s, nm)(); t1 = time.perf_counter(); torch.cuda.synchronize()
    sys.stderr.write("==== %s: host %.2f ms, drained after %.2f ms\n" % (nm, 1e3 * (t1 - t), 1e3 * (time.perf_counter() - t)))
PY
