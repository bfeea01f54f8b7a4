"""Where the time of one OASIS trace goes: deconvTemporal on 256 traces (one workgroup per CU) with the optimisation loops on / off.  python scripts/deconv_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
T, K = int(os.environ.get("T", "10000")), 256
f = synth.make_factors(128, 128, T, K, 3)
C = (np.ascontiguousarray(f.C_init[:K], dtype=np.float32) + 0.5 + 0.05 * np.random.default_rng(0).standard_normal((K, T)).astype(np.float32))
eng = Engine(0)
eng.profile(True)
base = {"type": "ar1", "method": "foopsi", "smin": -5.0, "max_tau": 100.0}
for name, extra in (("full (optimize_pars, optimize_b)", {"optimize_pars": True, "optimize_b": True}), ("optimize_b only", {"optimize_pars": False, "optimize_b": True}),
                    ("no optimisation", {"optimize_pars": False, "optimize_b": False})):
    opts = dict(base, **extra)
    eng.deconv_temporal(C.copy(), opts)
    eng.profile_reset()
    t0 = time.perf_counter(); eng.deconv_temporal(C.copy(), opts); dt = time.perf_counter() - t0
    tab = eng.profile_table()
    print("%-36s %s  wall %.1f ms" % (name, {k: round(v["total_ms"], 2) for k, v in tab.items() if v["total_ms"] > 0.05}, dt * 1e3), flush=True)
