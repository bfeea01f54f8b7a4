"""Round 3: the duo-role R1 kernel (r1_variant 14, resid_duo.hpp) against the one-pixel kernel (10; the first runs compared with variant 13, since removed) at the headline size:
   python scripts/r1_duo.py [--nsegs 0,1,2,4,8] [--probes 0,1,2,3,4,8] [--reps 5] [--small]
Numerics: max |Ysig(14) - Ysig(13)| / max |Ysig| under a FITTED W (a fresh ring has one value everywhere, any permutation of offsets would pass)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--nsegs", default="0,1,2,4,8"); ap.add_argument("--probes", default="0,1,2,3,4,8"); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--small", action="store_true"); ap.add_argument("--variants", default="10,14"); ap.add_argument("--ords", default="0"); ap.add_argument("--arcd", type=int, default=0)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
d1, d2, T, K, r, seed = (100, 75, 402, 20, 15, 2) if a.small else (512, 512, 10000, 500, 15, 2)
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.ring_init(0, r)
eng.fit_ring_model(0, None, None)
eng.set_option("r1_delta", 0)
if a.arcd:
    eng.set_option("r1_arc_d", a.arcd)


def timed(reps):
    eng.profile_reset()
    for _ in range(reps):
        eng.residual(0, None, None)
    eng.synchronize()
    tab = eng.profile_table()
    return tab["residual_r1"]["total_ms"] / tab["residual_r1"]["calls"]


eng.set_option("r1_variant", 10)          # the one-pixel-per-thread LDS-DMA kernel as the reference (variant 13 of the first runs was removed)
ref = eng.residual(0, None, None, want=True)
eng.profile(True)
for v in [int(x) for x in a.variants.split(",")]:
    eng.set_option("r1_variant", v)
    for od in [int(x) for x in a.ords.split(",")]:
        eng.set_option("r1_duo_ord", od)
        for ns in [int(x) for x in a.nsegs.split(",")]:
            eng.set_option("r1_nseg", ns)
            out = eng.residual(0, None, None, want=True)
            err = float(np.abs(out - ref).max()) / max(1e-30, float(np.abs(ref).max()))
            print("variant %d ord %d nseg %d: %.3f ms   max |Ysig - Ysig(13)| / max |Ysig| = %.2e" % (v, od, ns, timed(a.reps), err), flush=True)
            if v != 14:
                break
        if v != 14:
            break
eng.set_option("r1_nseg", 0)
for v in [int(x) for x in a.variants.split(",")]:
    eng.set_option("r1_variant", v)
    for pr in [int(x) for x in a.probes.split(",")]:
        eng.set_option("r1_probe", pr)
        print("variant %d r1_probe %2d (%s): %.3f ms" % (v, pr, "+".join(n for b, n in ((1, "noDMA"), (2, "noRing"), (4, "noExchange"), (8, "noStore")) if pr & b) or "full",
                                                       timed(max(2, a.reps // 2))), flush=True)
    eng.set_option("r1_probe", 0)
