"""Host-side timeline of ONE iteration: every engine call of the three update methods with its entry / exit time (ms from the iteration's start), on the GPU:
   python scripts/host_calls.py [--world 8 --rank 0 --patch 128 --lanes 2]      (default: c3, one patch)
Read beside a rocprofv3 --kernel-trace of the same run (scripts/gpu/r6_call52.sh / r6_call53.sh); scripts/host_timeline.py is the older, per-call-summary form: what the host is doing while the device idles."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0); ap.add_argument("--patch", type=int, default=512)
ap.add_argument("--lanes", type=int, default=1); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--force-collectives", action="store_true"); ap.add_argument("--c5", action="store_true", help="configs[4]: 1024 x 1024 x 20000, K = 2000 (use with --world 8 --patch 128: 8 of 64 patches)")
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = (1024, 1024, 20000, 2000, 15) if a.c5 else (512, 512, 10000, 500, 15)
f = synth.make_factors(d1, d2, T, K, 5 if a.c5 else 2)
eng = Engine(0)
if a.lanes > 1:
    eng.set_option("lanes", a.lanes)
video = PatchedVideo(d1, d2, T, [a.patch, a.patch], r, eng, rank=a.rank, world_size=a.world)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
group = None
if a.force_collectives:
    import torch.distributed as td
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    group = td.group.WORLD
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn, dist_group=group)
s.force_collectives = group is not None
log = []; depth = [0]
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*x, **k):
        t0 = time.perf_counter(); depth[0] += 1
        try:
            return fn(*x, **k)
        finally:
            depth[0] -= 1; log.append((t0, time.perf_counter(), depth[0], name))
    setattr(obj, name, w)
for name in dir(eng):
    if not name.startswith("__") and callable(getattr(eng, name)) and name not in ("close",):
        wrap(eng, name)
for name in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel", "_slice", "_post_process", "_search_location_csc", "_prefetch_search_location", "_update_b0_new",
             "_gather_sparse", "_rows", "_residual", "_allreduce", "_first_run", "ymean_full", "_csc_of_patches"):
    if hasattr(s, name):
        wrap(s, name)
def step():
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
log.clear(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); step(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
print("two iterations: host %.3f + %.3f ms, drained after %.3f ms more" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
for e0, e1, d, name in sorted(log):
    if e0 >= t1:
        print("%8.3f -> %8.3f  (%6.3f)  %s%s" % (1e3 * (e0 - t1), 1e3 * (e1 - t1), 1e3 * (e1 - e0), "  " * d, name))
