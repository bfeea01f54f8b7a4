"""Round-3 experiment: the one-barrier R1 kernel with UNEQUAL arcs (k_residual_arc_dma1b, option r1_arc_bias = 2, 3, 4, 5) against the default (equal arcs):
   python scripts/r1_arc_bias.py [--bias 0,2,3,4,5] [--reps 5]
For every bias: max |Ysig - Ysig(bias 0)| over the exported patch (a different summation order only: ~1e-6 relative) and the kernel's mean time at H.
The hypothesis: roles 1 and 3 share SIMDs 2 and 3 and role 3 also finishes the previous chunk, so handing their corner offsets
to roles 2 and 0 shortens the wait at the chunk barrier.  Not measured yet (written at the end of round 2 with no GPU time left)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--bias", default="0,2,3,4,5"); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--small", action="store_true", help="128 x 96 x 400 instead of the headline size (numerics only)")
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
d1, d2, T, K, r, seed = (128, 96, 400, 20, 15, 2) if a.small else (512, 512, 10000, 500, 15, 2)
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.ring_init(0, r)
# a fitted W makes the weights non-uniform (a fresh ring has one value everywhere: any permutation of the offsets would pass)
eng.fit_ring_model(0, None, None)
eng.set_option("r1_delta", 0); eng.set_option("r1_variant", 13)
ref = None
eng.profile(True)
for b in [int(x) for x in a.bias.split(",")]:
    eng.set_option("r1_arc_bias", b)
    out = eng.residual(0, None, None, want=True)
    if ref is None:
        ref = out
    err = float(np.abs(out - ref).max()) / max(1e-30, float(np.abs(ref).max()))
    eng.profile_reset()
    for _ in range(a.reps):
        eng.residual(0, None, None)
    eng.synchronize()
    tab = eng.profile_table()
    print("r1_arc_bias %d: %.3f ms   max |Ysig - Ysig(bias 0)| / max |Ysig| = %.2e" % (b, tab["residual_r1"]["total_ms"] / tab["residual_r1"]["calls"], err), flush=True)
eng.set_option("r1_arc_bias", 0)
