"""Seeded INPUTS for oracle/matlab/make_fixtures.m:  python tests/golden/make_matlab_inputs.py  ->  tests/golden/matlab_inputs.mat

The build image has neither MATLAB nor Octave, so the oracle is not pinned against the reference's own output ("parity
unpinned").  This file is one half of the route that pins it: a maintainer with MATLAB runs oracle/matlab/make_fixtures.m on a checkout of
zhoupc/CNMF_E, which feeds these inputs to the reference's functions and writes tests/golden/matlab_outputs.mat;
tests/test_matlab_fixtures.py then compares the oracle with those outputs (and is skipped while the file is absent).
Only data is stored here (inputs made by cnmf_e_amd.synth with fixed seeds)."""
import os, sys
import numpy as np
import scipy.io as sio
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cnmfe_oracle as orc
from cnmf_e_amd import synth
HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    d1, d2, T, K, r = 30, 26, 240, 5, 4
    f = synth.make_factors(d1, d2, T, K, 211, gSig=1.5, gSiz=7, min_sep=4)
    Y = synth.make_video(f, np.float32).T.astype(np.float64)                 # d x T
    patch = np.array([6, 25, 5, 22]); block = np.array([1, 30, 1, 26])       # one interior patch inside its block
    rs, cs = orc.get_nhood(r)
    W0 = sp.csc_matrix(orc.build_ring_W(patch, block, d1, d2, rs, cs))
    ip = orc.ind_patch_mask(patch, block)
    A = f.A_init.toarray().astype(np.float64); C = f.C_init.astype(np.float64)
    rng = np.random.default_rng(5)
    imgs = np.stack([f.A_true[:, k].toarray().reshape(d1, d2, order="F") for k in range(K)], axis=2)
    for k in range(K):                                                       # specks and holes for the morphology
        pix = rng.integers(0, d1 * d2, 6)
        im = imgs[:, :, k].reshape(-1, order="F"); im[pix] += rng.uniform(0.05, 0.4, 6); imgs[:, :, k] = im.reshape(d1, d2, order="F")
    tr = np.load(os.path.join(HERE, "oasis_ar1.npz"))["y"].astype(np.float64)
    return dict(d1=float(d1), d2=float(d2), T=float(T), K=float(K), radius=float(r), Y=Y, A=sp.csc_matrix(A), C=C, sn=np.asarray(f.sn, np.float64).reshape(-1, 1),
                W0=W0, ind_patch=ip.reshape(-1, 1), patch=patch.astype(np.float64), block=block.astype(np.float64),
                imgs=imgs, resize_img=rng.standard_normal((23, 17, 3)), quant_x=rng.integers(0, 40, 57).astype(np.float64),
                quant_p=np.array([0.05, 0.31, 0.5, 0.8, 0.97]), traces=tr, nhood_radii=np.array([3.0, 5.0, 15.0, 18.0]),
                thresh_outlier=3.0, maxN=20.0)


if __name__ == "__main__":
    sio.savemat(os.path.join(HERE, "matlab_inputs.mat"), build(), do_compression=True)
    print("written", os.path.getsize(os.path.join(HERE, "matlab_inputs.mat")), "bytes")
