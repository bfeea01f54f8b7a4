#!/bin/bash
# which buffers are re-allocated in which iteration of c4 on three lanes (CNMFE_TRACE_ALLOC): the fenced iterations 2-4 take 31-36 ms, from the 5th on 21
cd $GRAFT_REPO_ROOT
CNMFE_TRACE_ALLOC=1 timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-200 > gpurun_out/alloc_trace.txt
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
for it in range(8):
    t0 = time.perf_counter()
    sys.stderr.write("== iteration %d bg\n" % it); s.update_background_parallel(); torch.cuda.synchronize(); t1 = time.perf_counter()
    sys.stderr.write("== iteration %d spatial\n" % it); s.update_spatial_parallel(); torch.cuda.synchronize(); t2 = time.perf_counter()
    sys.stderr.write("== iteration %d temporal\n" % it); s.update_temporal_parallel(); torch.cuda.synchronize(); t3 = time.perf_counter()
    sys.stderr.write("== iteration %d: bg %.2f spatial %.2f temporal %.2f ms\n" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
PY
grep -c alloc gpurun_out/alloc_trace.txt; grep "==.*ms" gpurun_out/alloc_trace.txt
