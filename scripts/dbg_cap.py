import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cnmfe_oracle as orc
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
eng = Engine(0)
d1, d2, T, r = 48, 48, 64, 5
f = synth.make_factors(d1, d2, T, 3, 7, gSig=1.5, gSiz=7, min_sep=5)
Y = synth.make_video(f, np.float32)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng); video.upload_from_full(Y)
eng.ring_init(0, r); eng.fit_ring_model(0, None, None)
K = 44
rs, cs = orc.get_nhood(r); rs, cs = np.ravel(rs), np.ravel(cs)
rows = [(24 + int(cs[i % rs.size]) + (i // rs.size)) * d1 + 24 + int(rs[i % rs.size]) for i in range(K)]
A = sp.csc_matrix((np.ones(K, np.float32), (rows, np.arange(K))), shape=(d1 * d2, K))
Cm = np.random.default_rng(0).random((K, T)).astype(np.float32)
W = eng.ring_csr(0).astype(np.float64)
cnt = np.diff((abs(W) @ abs(A)).tocsr().indptr)
b0f = Y.astype(np.float64).mean(axis=0).astype(np.float32); eng.set_b0(0, b0f)
Yb = Y.T.astype(np.float64)
print("max |Y|", np.abs(Y).max(), "max count", cnt.max())
for Ksel in (K, 20, 32, 33):
    out = eng.residual(0, A[:, :Ksel], Cm[:Ksel], want=True)
    ref = orc.residual_ysig(Yb, A[:, :Ksel].astype(np.float64), Cm[:Ksel], W, b0f.astype(np.float64), np.ones(d1 * d2, dtype=bool))
    e = np.abs(out.T - ref).max(axis=1)
    cK = np.diff((abs(W) @ abs(A[:, :Ksel])).tocsr().indptr)
    worst = np.argsort(-e)[:6]
    print("K", Ksel, "max err", e.max(), "worst pixels", [(int(w), float(e[w]), int(cK[w])) for w in worst], "median", np.median(e))
