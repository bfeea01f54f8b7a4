"""What ONE rank of an N-rank c4 run computes per iteration, without the collectives (single process, dist_group = None, only this rank's patches):
   python scripts/rank_load.py --world 8 [--rank 0]
An estimate of the per-rank floor of `bench.py --gpus N` (the collectives and the other ranks' stragglers come on top)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0); ap.add_argument("--lanes", type=int, default=1); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--prof", default=""); ap.add_argument("--force-collectives", action="store_true", help="every collective branch of the three methods behind an RCCL group of ONE rank (the host side of the collectives, no second GPU)")
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 512, 512, 10000, 500, 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0)
if a.lanes > 1:
    eng.set_option("lanes", a.lanes)
video = PatchedVideo(d1, d2, T, [128, 128], r, eng, rank=a.rank, world_size=a.world)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
group = None
if a.force_collectives or os.environ.get("CNMFE_BENCH_FORCE_COLLECTIVES", "0") == "1":
    import torch.distributed as td
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    group = td.group.WORLD
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5), f.A_init, f.C_init, f.sn, dist_group=group)
s.force_collectives = group is not None
def step():
    t0 = time.perf_counter(); s.update_background_parallel(); t1 = time.perf_counter(); s.update_spatial_parallel(); t2 = time.perf_counter(); s.update_temporal_parallel()
    return t1 - t0, t2 - t1, time.perf_counter() - t2
for _ in range(2):
    step()
torch.cuda.synchronize()
# timed WITHOUT per-launch events (a pair of events costs host and device time at every one of ~170 launches: what a run pays is the loop below); the kernel sums
# come from extra steps with the events on
if a.prof:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); parts = np.zeros(3)
for _ in range(a.steps):
    parts += step()
if a.prof:
    pr.disable()
    with open(a.prof, "w") as fh:
        pstats.Stats(pr, stream=fh).sort_stats("cumulative").print_stats(90); pstats.Stats(pr, stream=fh).sort_stats("tottime").print_stats(60)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
eng.profile(True); eng.profile_reset()
nprof = 6
t1 = time.perf_counter()
for _ in range(nprof):
    step()
torch.cuda.synchronize()
dtp = (time.perf_counter() - t1) / nprof
tab = eng.profile_table()
print("rank %d of %d (%d patches, %d lane(s))%s: %.2f ms / iteration (%.2f with events around every launch); host time in calls: bg %.2f spatial %.2f temporal %.2f ms; kernel sum %.2f ms" % (
    a.rank, a.world, len(video.owned), a.lanes, " with the collectives of a group of one" if group is not None else "", 1e3 * dt, 1e3 * dtp, *(1e3 * parts / a.steps), sum(v["total_ms"] for v in tab.values()) / nprof))
print({k: round(v["total_ms"] / nprof, 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["total_ms"])[:14]})
