"""Gram precision modes at the headline size: time + W agreement with the fp64-MFMA mode.  python scripts/gram_modes.py [--modes 1,2,3]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--modes", default="1,2,3"); ap.add_argument("--probe", type=int, default=0); ap.add_argument("--flushes", default="4"); ap.add_argument("--kernel", type=int, default=4); ap.add_argument("--incr", type=int, default=0)
a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
d1, d2, T, K, r, seed = 512, 512, 10000, 500, 15, 2
f = synth.make_factors(d1, d2, T, K, seed)
Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
eng.profile(True)
A = f.A_init.astype(np.float32)
Ws = {}
for mode, fl in [(int(x), int(y)) for x in a.modes.split(",") for y in a.flushes.split(",")]:
    eng.ring_init(0, r); eng.set_option("gram_incremental", a.incr); eng.set_option("gram_kernel", a.kernel); eng.set_option("gram_mode", mode); eng.set_option("gram_probe", a.probe); eng.set_option("gram_flush", fl)
    eng.fit_ring_model(0, A, f.C_init); eng.profile_reset()
    eng.ring_init(0, r)
    eng.fit_ring_model(0, A, f.C_init)
    tab = eng.profile_table()
    Ws[(mode, fl) if mode != 1 else 1] = eng.ring_csr(0).data.astype(np.float64)
    print("gram_mode %d flush %d:" % (mode, fl), {k: round(v["total_ms"] / v["calls"], 2) for k, v in tab.items() if k.startswith("bg_") and v["calls"] and v["total_ms"] > 0.05}, flush=True)
ref = Ws.get(1)
if ref is not None:
    for m, w in Ws.items():
        if m != 1:
            print("mode %s vs fp64: rel %.3e  max abs %.3e" % (m, np.linalg.norm(w - ref) / np.linalg.norm(ref), np.abs(w - ref).max()))
