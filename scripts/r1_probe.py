"""Timing ablations of the default R1 kernel (k_residual_arc_dma, option r1_probe): python scripts/r1_probe.py [--probes 0,1,2,4,8,...]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--probes", default="0,1,2,3,4,6,7,8,15"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--variant", type=int, default=11); ap.add_argument("--d1", type=int, default=512); ap.add_argument("--d2", type=int, default=512); ap.add_argument("--nseg", type=int, default=0); ap.add_argument("--arcd", type=int, default=4)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
d1, d2, T, K, r, seed = a.d1, a.d2, 10000, max(10, 500 * a.d1 * a.d2 // (512 * 512)), 15, 2
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.ring_init(0, r)
eng.set_option("r1_delta", 0); eng.set_option("r1_variant", a.variant); eng.set_option("r1_nseg", a.nseg); eng.set_option("r1_arc_d", a.arcd)
eng.residual(0, None, None)
eng.profile(True)
for pr in [int(x) for x in a.probes.split(",")]:
    eng.set_option("r1_probe", pr); eng.profile_reset()
    for _ in range(a.reps):
        eng.residual(0, None, None)
    eng.synchronize()
    tab = eng.profile_table()
    print("r1_probe %2d (%s): %.3f ms" % (pr, "+".join(n for b, n in ((1, "noDMA"), (2, "noRing"), (4, "noExchange"), (8, "noStore")) if pr & b) or "full",
                                        tab["residual_r1"]["total_ms"] / tab["residual_r1"]["calls"]), flush=True)
