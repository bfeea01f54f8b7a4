"""The video's covariance table on the int8 matrix pipe (option gram_i8, gram_i8.hpp) against the fp64 MFMA table: first-run fit of the same patch with both,
the weights compared, the kernels' times printed.  python scripts/gram_i8_check.py --cfg c2|c3|small"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--cfg", default="c2"); a = ap.parse_args()
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
CFG = {"c2": (256, 256, 3000, 200, 15, 1), "c3": (512, 512, 10000, 500, 15, 2), "small": (96, 80, 1000, 20, 15, 4), "r18": (128, 128, 2000, 30, 18, 5)}
d1, d2, T, K, r, seed = CFG[a.cfg]
f = synth.make_factors(d1, d2, T, K, seed)
Ws = {}
for mode in (0, 1):
    Yd = synth.make_video_device(f, "cuda:0"); torch.cuda.synchronize()
    eng = Engine(0)
    eng.set_option("gram_i8", mode)
    video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
    video.upload_block_device((0, 0), Yd.data_ptr()); del Yd; torch.cuda.empty_cache()
    eng.profile(True)
    eng.ring_init(0, r)
    _, info = eng.fit_ring_model(0, f.A_init.astype(np.float32), f.C_init)
    eng.synchronize()
    tab = eng.profile_table()
    print("gram_i8 = %d: %s   (%s)" % (mode, "  ".join("%s %.2f ms" % (k, v["total_ms"]) for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["total_ms"])[:7]), info), flush=True)
    Ws[mode] = eng.ring_csr(0).data.copy()
    eng.close(); del video, eng
dW = np.abs(Ws[1] - Ws[0])
print("max |W_i8 - W_f64| / max|W| = %.3e   (rms %.3e, nan %d, entries that differ in any bit %d of %d)" % (
    np.nanmax(dW) / np.abs(Ws[0]).max(), np.sqrt(np.nanmean(dW ** 2)), int(np.isnan(Ws[1]).sum()), int((Ws[1].view(np.uint32) != Ws[0].view(np.uint32)).sum()), Ws[0].size))
