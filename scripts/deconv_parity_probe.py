"""What the deconvolution parity tests observe (max |d gamma|, worst-trace relative errors), without their assertions:
   python scripts/deconv_parity_probe.py      (run from the repo root on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oasis_oracle as oo
from cnmf_e_amd.engine import Engine
from parity_util import rel
import test_gpu_parity as tp
import test_gpu_edges as te
eng = Engine(0)
c, pid, ysig, A_p = tp._deconv_case(eng)
_, Craw0, _ = eng.hals_temporal(pid, A_p, c.f.C_init, 3)
Craw0 = Craw0 + 0.7
Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Craw0, None)
Cr, Crawr, Sr, parsr, snr = oo.deconvTemporal(Craw0.astype(np.float64))
print("deconvTemporal (T=1500, K=5): max |d gamma| %.2e  sn rel %.2e  worst C %.2e  worst Craw %.2e  spike-count diffs %s" % (
    np.abs(parsg - parsr).max(), np.abs(sng / snr - 1).max(), max(rel(Cg[k], Cr[k]) for k in range(len(Cg))), max(rel(Crawg[k], Crawr[k]) for k in range(len(Cg))),
    [int((Sg[k] > 0).sum() - (Sr[k] > 0).sum()) for k in range(len(Cg))]))
for T in (3999, 4096):
    Y = te._ar1_traces(3, T)
    opts = dict(type="ar1", method="foopsi", smin=0.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    Cg, Crawg, Sg, parsg, sng = eng.deconv_temporal(Y, opts)
    Cr, Crawr, Sr, parsr, snr = oo.deconvTemporal(Y.astype(np.float64), smin=0.0, optimize_pars=True, optimize_b=True, max_tau=100.0)
    print("smin = 0, T = %d: max |d gamma| %.2e  worst C %.2e  worst Craw %.2e  spike-count diffs %s" % (
        T, np.abs(parsg - parsr).max(), max(rel(Cg[k], Cr[k]) for k in range(3)), max(rel(Crawg[k], Crawr[k]) for k in range(3)),
        [int((Sg[k] > 0).sum() - (Sr[k] > 0).sum()) for k in range(3)]))
c, pid, ysig, A_p = tp._deconv_case(eng, T=1200, K=4)
Cg, Crawg, Sg, sng, parsg, aa = eng.hals_temporal_deconv(pid, A_p, c.f.C_init, 2, None)
Cr, Crawr, Sr, snr, parsr = oo.HALS_temporal_deconv(ysig, A_p.astype(np.float64), c.f.C_init, 2)
parsr = np.array([0.0 if g is None else g for g in parsr])
print("HALS_temporal deconv (T=1200, K=4, 2 sweeps): max |d gamma| %.2e  worst C %.2e  worst Craw %.2e" % (
    np.abs(parsg - parsr).max(), max(rel(Cg[k], Cr[k]) for k in range(len(Cg))), max(rel(Crawg[k], Crawr[k]) for k in range(len(Cg)))))
