#!/bin/bash
cd $GRAFT_REPO_ROOT
CNMFE_BENCH_R1=0 CNMFE_BENCH_LANES=3 timeout 280 python bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('c4 lanes 3 warmup 2, CNMFE_BENCH_R1=0:', round(d['ms_per_step'],2), [round(x,1) for x in d['first_iteration']['warmup_steps_ms']])"
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-200
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
for prof in (0, 1):
    eng = Engine(0); eng.set_option("lanes", 3)
    video = PatchedVideo(d1, d2, T, [128, 128], r, eng)
    for idx in video.owned:
        Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
        video.upload_block_device(idx, Yb.data_ptr()); del Yb
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
    if prof: eng.profile(True)
    ts = []
    for it in range(9):
        t0 = time.perf_counter()
        s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
        torch.cuda.synchronize()
        ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    print("own loop, one fence per iteration, events around every launch = %d:" % prof, ts)
    eng.close(); del s, video, eng
PY
