"""W accuracy vs the float64 oracle and Gram time for gram_mode / gram_flush (run on a GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cnmfe_oracle as orc
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo

d1, d2, T, K, r = 64, 64, 4000, 8, 15
f = synth.make_factors(d1, d2, T, K, 3)
Y = synth.make_video(f, np.float32)
eng = Engine(0)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng)
video.upload_from_full(Y)
rs, cs = orc.get_nhood(r)
W0 = orc.build_ring_W([1, d1, 1, d2], [1, d1, 1, d2], d1, d2, rs, cs)
t0 = time.time()
Wref, _ = orc.fit_ring_model(Y.T.astype(np.float64), f.A_init.astype(np.float32).astype(np.float64), f.C_init, W0, np.nan, None, None, True)
print("oracle fit %.1fs" % (time.time() - t0), flush=True)
Wref = Wref.tocsr(); Wref.sort_indices()
R = Y.T.astype(np.float64) - f.A_init.astype(np.float32) @ f.C_init
Rc = R - R.mean(1, keepdims=True)
Bref = Wref @ Rc
eng.profile(True)
for mode, flush in ((1, 0), (2, 1), (2, 2), (2, 4), (2, 8), (2, 16), (2, 100000)):
    eng.ring_init(0, r)
    eng.set_option("gram_mode", mode); eng.set_option("gram_flush", max(1, flush))
    eng.profile_reset()
    eng.fit_ring_model(0, f.A_init.astype(np.float32), f.C_init)
    W = eng.ring_csr(0)
    tab = eng.profile_table()
    ms = [v["total_ms"] for k, v in tab.items() if k.startswith("bg_gram")][0]
    eW = np.linalg.norm(W.data - Wref.data) / np.linalg.norm(Wref.data)
    eB = np.linalg.norm(W.astype(np.float64) @ Rc - Bref) / np.linalg.norm(Bref)
    print("gram_mode %d flush %6d: rel err W %.2e   rel err W*(R-Rbar) %.2e   gram %.3f ms" % (mode, flush, eW, eB, ms), flush=True)
