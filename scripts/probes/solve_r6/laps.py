"""Wall-clock laps of one wave of k_ring_solve8 (a build with -DRSP_DEBUG: CNMFE_EXTRA_FLAGS=-DRSP_DEBUG python -m cnmf_e_amd.build --force), solve_probe 16384:
every 509th pixel writes the 10-ns ticks between its phase boundaries into its own weights.  python scripts/probes/solve_r6/laps.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
A0 = f.A_init.astype(np.float32)
eng.ring_init(0, r); eng.fit_ring_model(0, A0, f.C_init); eng.synchronize()
eng.set_option("solve_probe", 16384)
eng.ring_init(0, r); eng.profile_reset(); eng.fit_ring_model(0, A0, f.C_init); eng.synchronize()
tab = eng.profile_table(); print("bg_ring_solve %.3f ms" % (tab["bg_ring_solve"]["total_ms"] / tab["bg_ring_solve"]["calls"]))
W = eng.ring_csr(0)
names = ["list + geometry + border (issue)", "staging (window samples)", "system arrives + corrections", "trace + ridge", "factorisation", "substitutions"]
rows = [m for m in range(7, d1 * d2, 509) if W.indptr[m + 1] - W.indptr[m] == 96]
L = np.array([W.data[W.indptr[m]:W.indptr[m] + 6] for m in rows], dtype=np.float64) * 0.01      # 100 MHz ticks -> us
print("%d pixels sampled (whole rings); per phase: median / 10 %% / 90 %% us" % len(rows))
for i, n in enumerate(names):
    print("  %-34s %7.2f  %7.2f  %7.2f" % (n, np.median(L[:, i]), np.quantile(L[:, i], 0.1), np.quantile(L[:, i], 0.9)))
print("  %-34s %7.2f" % ("sum of medians", np.median(L, 0).sum()))
