import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo
import cnmfe_oracle as orc
eng = Engine(0)
d1, d2, T, r = 80, 72, 24, 15
f = synth.make_factors(d1, d2, T, 4, 3, gSig=1.5, gSiz=7, min_sep=4)
Y = synth.make_video(f, np.float32)
video = PatchedVideo(d1, d2, T, [d1, d2], r, eng); video.upload_from_full(Y); eng.ring_init(0, r)
rng = np.random.default_rng(7)
rs, cs = orc.get_nhood(r)
W0 = orc.build_ring_W(video.patch_pos[(0, 0)], video.block_pos[(0, 0)], d1, d2, rs, cs).tocsr(); W0.sort_indices()
eng.ring_set_values(0, (W0.data * (1 + 0.5 * rng.standard_normal(W0.nnz))).astype(np.float32))
eng.set_b0(0, np.full(W0.shape[0], 990.0, dtype=np.float32))
for K in (10, 40, 80, 160):
    rows, cols, vals = [], [], []
    for k in range(K):
        r0, c0 = rng.integers(0, 78), rng.integers(0, 70)
        for dr in range(3):
            for dc in range(3):
                rows.append((c0 + dc) * 80 + r0 + dr); cols.append(k); vals.append(rng.random() + 0.1)
    A_b = sp.csc_matrix((np.array(vals, np.float32), (rows, cols)), shape=(d1 * d2, K)); A_b.sum_duplicates()
    Cm = rng.random((K, T)).astype(np.float32) * 5
    out = {}
    for v in (-1, 2, 10):
        eng.set_option("r1_variant", v); out[v] = eng.residual(0, A_b, Cm, want=True).astype(np.float64)
    WA = (W0.multiply(0) + sp.csr_matrix((eng.ring_csr(0).data, W0.indices, W0.indptr), shape=W0.shape)) @ A_b
    nwa = np.diff(sp.csr_matrix(WA).indptr)
    for v in (2, 10):
        e = np.abs(out[v] - out[-1]).max(axis=0)
        bad = np.argsort(e)[-3:]
        print("K=%d variant %d: rel %.2e  max abs err %.3e at px %s nwa there %s ; nwa max %d mean %.1f" % (
            K, v, np.linalg.norm(out[v] - out[-1]) / np.linalg.norm(out[-1]), e.max(), bad, nwa[bad], nwa.max(), nwa.mean()))
