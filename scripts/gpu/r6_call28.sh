#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-400
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from cnmf_e_amd import synth, _lib as L
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0); eng.set_option("lanes", int(os.environ.get("LANES", "3")))
video = PatchedVideo(d1, d2, T, [128, 128], r, eng, rank=0, world_size=int(os.environ.get("WORLD", "1")))
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn)
acc = {"take": 0.0, "ntake_new": 0, "give": 0}
ot, og = eng._pinned_take, eng._pinned_give
def take(n):
    pool = eng.__dict__.get("_pinned_pool", {}); hit = bool(pool.get(n))
    t0 = time.perf_counter(); p = ot(n); acc["take"] += 1e3 * (time.perf_counter() - t0); acc["ntake_new"] += 0 if hit else 1
    return p
def give(p, n):
    acc["give"] += 1; return og(p, n)
eng._pinned_take = take; eng._pinned_give = give
real = L.lib.cnmfe_update_spatial_fetch_async
tf = [0.0]
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); tf[0] += 1e3 * (time.perf_counter() - t0); return r
L.lib.__dict__["cnmfe_update_spatial_fetch_async"] = Wrap()
for it in range(int(os.environ.get("ITERS", "7"))):
    acc.update(take=0.0, ntake_new=0, give=0); tf[0] = 0.0
    t0 = time.perf_counter()
    s.update_background_parallel(); t1 = time.perf_counter(); s.update_spatial_parallel(); t2 = time.perf_counter(); s.update_temporal_parallel()
    torch.cuda.synchronize()
    print("iteration %d: %.1f ms (spatial host %.1f); pinned takes %.2f ms, %d of them new allocations, %d gives; native fetch_async calls %.2f ms; pool: %s" % (
        it, 1e3 * (time.perf_counter() - t0), 1e3 * (t2 - t1), acc["take"], acc["ntake_new"], acc["give"], tf[0], {k: len(v) for k, v in eng.__dict__.get("_pinned_pool", {}).items()}))
PY
