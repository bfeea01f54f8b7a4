"""Host-side timeline of one iteration at the headline size: time inside every Engine call and the Python time between calls.
python scripts/host_timeline.py [--iters 3]"""
import argparse, os, sys, time, functools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=3); ap.add_argument("--bg-ssub", type=int, default=1); ap.add_argument("--cprofile", action="store_true"); ap.add_argument("--deconv", action="store_true"); ap.add_argument("--patch", type=int, default=512, help="patch side (128: the 4 x 4 patches of BASELINE configs[3] on the 512 x 512 FOV)"); ap.add_argument("--npatch", type=int, default=1, help="patches side by side on this one rank (FOV 512 x 512*npatch, K = 500*npatch)"); ap.add_argument("--force-collectives", action="store_true", help="one-rank nccl group + force_collectives: the collective branches' host side"); ap.add_argument("--as-rank-of", type=int, default=0, help="N: only rank 0's patches of an N-rank run, no collectives (with --patch 128: one rank's share of c4, as scripts/rank_load.py)")
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
group = None
if world > 1:                                   # sharded run on ONE device (gloo): host-side costs of rank 0; kernel times are inflated by the shared GPU
    import torch.distributed as td
    td.init_process_group(backend="gloo"); group = td.group.WORLD
    if a.patch == 512:
        a.npatch = world                        # weak: the FOV grows with the ranks; with --patch 128 the 4 x 4 patches of the fixed FOV are sharded (c4)
if a.as_rank_of:
    world, rank, group = a.as_rank_of, 0, None
if a.force_collectives:
    import torch.distributed as td
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    print("affinity before init:", len(os.sched_getaffinity(0)), "cpus")
    td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0)); group = td.group.WORLD
    _t = torch.ones(1, device="cuda"); td.all_reduce(_t); torch.cuda.synchronize()
    print("affinity after init + first collective:", len(os.sched_getaffinity(0)), "cpus", sorted(os.sched_getaffinity(0))[:8])
    import threading
    print("threads:", threading.active_count(), "os threads:", len(os.listdir("/proc/self/task")))
    if os.environ.get("RESET_AFFINITY") == "1":
        os.sched_setaffinity(0, range(os.cpu_count())); print("affinity reset:", len(os.sched_getaffinity(0)))
d1, d2, T, K, r, seed = 512, 512 * a.npatch, 10000, 500 * a.npatch, 15, 2
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [a.patch, a.patch], r, eng, rank=rank, world_size=world)
if a.npatch == 1 and a.patch == 512:
    video.upload_block_device((0, 0), Yd.data_ptr())
else:
    del Yd
    for idx in video.owned:
        Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
        video.upload_block_device(idx, Yb.data_ptr()); del Yb
Yd = None; torch.cuda.empty_cache()
log = []
import threading
main = threading.get_ident()
for name in dir(Engine):
    fn = getattr(Engine, name)
    if name.startswith("_") or not callable(fn): continue
    def wrap(fn, name):
        @functools.wraps(fn)
        def w(*x, **k):
            t0 = time.perf_counter(); r_ = fn(*x, **k); t1 = time.perf_counter()
            log.append((name, t0, t1, threading.get_ident() == main)); return r_
        return w
    setattr(Engine, name, wrap(fn, name))
if os.environ.get("TRACE_SLOW_PY") == "1":               # which Python-side helper a long gap between two engine calls sits in (printed when > 3 ms)
    import cnmf_e_amd.sources2d as _s2
    def _wrap_slow(obj, name):
        fn = getattr(obj, name)
        static = isinstance(getattr(obj, "__dict__", {}).get(name), staticmethod)
        @functools.wraps(fn)
        def w(*x, **k):
            t0 = time.perf_counter(); r_ = fn(*x, **k); dt_ = time.perf_counter() - t0
            if dt_ > 3e-3: print("      [slow python] %s %.1f ms" % (name, dt_ * 1e3), flush=True)
            return r_
        setattr(obj, name, staticmethod(w) if static else w)
    for nm in ("_slice", "_rows", "_prev_block_of", "_residual", "_temporal_residual_done", "_temporal_residual_early", "_gather_sparse", "_allreduce", "_post_process", "_first_run",
               "_search_location_csc", "_update_b0_new", "ymean_full", "_prefetch_search_location"):
        if hasattr(Sources2D, nm): _wrap_slow(Sources2D, nm)
    for nm in ("rows_of", "_select_rows_native", "_csc_from_triplets", "determine_search_location"):
        if hasattr(_s2, nm): _wrap_slow(_s2, nm)
s = Sources2D(video, Options(ring_radius=r, bg_ssub=a.bg_ssub, deconv_flag=a.deconv), f.A_init, f.C_init, f.sn, dist_group=group)
if a.force_collectives:
    s.force_collectives = True
if rank != 0:
    sys.stdout = open(os.devnull, 'w')
marks = []
import cProfile, pstats
prof = cProfile.Profile() if a.cprofile else None
for it in range(a.iters):
    torch.cuda.synchronize(); del log[:]; t0 = time.perf_counter()
    if prof and it == a.iters - 1: prof.enable()
    s.update_background_parallel(); tb = time.perf_counter()
    s.update_spatial_parallel(); ts = time.perf_counter()
    s.update_temporal_parallel(); tt = time.perf_counter()
    if prof and it == a.iters - 1: prof.disable()
    torch.cuda.synchronize(); te = time.perf_counter()
    print("iteration %d: %.1f ms" % (it, (te - t0) * 1e3), flush=True)
print("last iteration: bg %.1f  spatial %.1f  temporal %.1f  drain %.1f  total %.1f ms" % ((tb - t0) * 1e3, (ts - tb) * 1e3, (tt - ts) * 1e3, (te - tt) * 1e3, (te - t0) * 1e3))
prev = t0
for name, a0, a1, on_main in log:
    if not on_main:
        print("      [thread] %-22s %.2f ms (at %.1f)" % (name, (a1 - a0) * 1e3, (a0 - t0) * 1e3)); continue
    print("%7.2f ms python | %-22s %7.2f ms in call (at %.1f)" % ((a0 - prev) * 1e3, name, (a1 - a0) * 1e3, (a0 - t0) * 1e3)); prev = a1
print("%7.2f ms python tail" % ((tt - prev) * 1e3))

if prof:
    st = pstats.Stats(prof); st.sort_stats("tottime").print_stats(28)
