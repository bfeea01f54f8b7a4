import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
import cnmfe_oracle as orc
rel = lambda a, b: np.linalg.norm(np.asarray(a, float) - np.asarray(b, float)) / max(np.linalg.norm(b), 1e-30)
eng = Engine(0)
d1, d2, T, K, r = 46, 42, 300, 6, 6
f = synth.make_factors(d1, d2, T, K, 23, gSig=1.5, gSiz=7, min_sep=5)
Y = synth.make_video(f, np.float32)
video = PatchedVideo(d1, d2, T, [23, 21], r, eng); video.upload_from_full(Y)
s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=3, bg_ssub=2), f.A_init, f.C_init, f.sn)
o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [23, 21], r, f.A_init.astype(np.float32), f.C_init, f.sn, spatial_algorithm="hals", maxIter=3, bg_ssub=2)
s.update_background_parallel(); o.update_background_parallel()
s.update_spatial_parallel(); o.update_spatial_parallel()
s.update_temporal_parallel(); o.update_temporal_parallel()
print("after it0: A rel %.2e C rel %.2e" % (rel(s.A.toarray(), o.A.toarray()), rel(s.C, o.C)))
for idx in video.owned:
    Wg = s.get_W(idx).toarray(); Wr = np.asarray(sp.csr_matrix(o.W[idx]).todense())
    zr = np.nonzero(np.abs(Wg).sum(axis=1) == 0)[0]
    print("before 2nd bg", idx, "zero rows engine", zr.size, zr[:12], "oracle", int((np.abs(Wr).sum(axis=1) == 0).sum()), "rel", rel(Wg, Wr))
# feed the oracle's state into the engine-side object
s.A = sp.csc_matrix(o.A.astype(np.float32)); s.C = o.C.astype(np.float32)
infos = s.update_background_parallel(); o.update_background_parallel()
print(infos)
for idx in video.owned:
    Wg = s.get_W(idx).toarray(); Wr = np.asarray(sp.csr_matrix(o.W[idx]).todense())
    e = np.abs(Wg - Wr).max(axis=1)
    bad = np.argsort(e)[-5:]
    print(idx, "W rel %.3e" % rel(Wg, Wr), "worst rows", bad, e[bad], "row norms", np.abs(Wr[bad]).max(axis=1), "shape", Wg.shape)

# which ind_active convention does the engine match?  refit patch (1,0) with the oracle from the same W_old
