"""Where the HOST spends the time between the spatial update's download and the temporal update's first launch at configuration c3 (one patch):
   python scripts/host_gap_probe.py
wall-clock per call of the Python-side steps (engine fetch, its compaction, stitch_begin, hals_temporal up to its return), averaged over iterations."""
import argparse, os, sys, time, collections
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--bg-ssub", type=int, default=1); ap.add_argument("--deconv", action="store_true"); ap.add_argument("--patch", type=int, default=512, help="128: the 4 x 4 patches of configs[3]")
a_ = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cnmf_e_amd import synth, _lib as L
from cnmf_e_amd.engine import Engine
from cnmf_e_amd import sources2d as S2
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1 = d2 = 512; T = 10000; K = 500; r = 15
f = synth.make_factors(d1, d2, T, K, 2)
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [a_.patch, a_.patch], r, eng)
for idx in video.owned:
    Yb = synth.make_video_device(f, "cuda:0", pixels=video.block_pix[idx]); torch.cuda.synchronize()
    video.upload_block_device(idx, Yb.data_ptr()); del Yb
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=5, bg_ssub=a_.bg_ssub, deconv_flag=a_.deconv), f.A_init, f.C_init, f.sn)
acc = collections.defaultdict(float); cnt = collections.Counter(); marks = []
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); marks.append((label or name, "in", t))
        try:
            return fn(*a, **k)
        finally:
            t2 = time.perf_counter(); acc[label or name] += t2 - t; cnt[label or name] += 1; marks.append((label or name, "out", t2))
    setattr(obj, name, w)
for n in ("stitch_begin", "hals_temporal", "hals_temporal_deconv", "deconv_temporal_bound", "stitch_add", "stitch_finish", "update_spatial", "residual", "residual_ssub",
          "fit_ring_model", "fit_ring_model_ssub", "ring_first_run", "bind_traces", "hals_temporal_job", "temporal_jobs_sweep", "stitch_add_job", "post_process_spatial"):
    timed(eng, n)
for n in ("_update_b0_new", "_temporal_residual_early", "_search_location_csc", "_prev_block_of", "deconvTemporal", "update_background_parallel", "update_spatial_parallel", "update_temporal_parallel",
          "_slice", "_gather_sparse", "_post_process"):
    timed(s, n)
L.lib.cnmfe_version()                                                     # (loads the library)
for n in ("cnmfe_update_spatial_fetch_connected", "cnmfe_csc_drop_zeros", "cnmfe_hals_temporal", "cnmfe_stitch_wait", "cnmfe_synchronize"):
    timed(L._Lib._dll, n)
def step():
    t0 = time.perf_counter(); s.update_background_parallel(); t1 = time.perf_counter(); s.update_spatial_parallel(); t2 = time.perf_counter(); s.update_temporal_parallel(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3):
    step()
acc.clear(); cnt.clear()
N = 8; tot = np.zeros(3)
for _ in range(N):
    marks.clear()
    tot += step()
eng.synchronize()
print("per iteration: background %.3f ms, spatial %.3f ms, temporal %.3f ms (host wall of the three calls)" % tuple(1e3 * tot / N))
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print("%-28s %8.3f ms per iteration in %d calls" % (k, 1e3 * v / N, cnt[k] // N))
if a_.patch == 512:
    print("last iteration, order of events (ms from the first):")
    t0 = marks[0][2]
    for name, io, t in marks:
        print("  %8.3f  %-4s %s" % (1e3 * (t - t0), io, name))
