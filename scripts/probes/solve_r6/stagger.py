import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
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
for probe in (0, 8192 + (1 << 16), 8192 + (2 << 16), 8192 + (3 << 16), 8192 + (5 << 16), 0):
    eng.set_option("solve_probe", probe)
    ts = []
    for rep in range(3):
        eng.ring_init(0, r); eng.profile_reset()
        eng.fit_ring_model(0, A0, f.C_init); eng.synchronize()
        tab = eng.profile_table(); ts.append(tab["bg_ring_solve"]["total_ms"] / tab["bg_ring_solve"]["calls"])
    print("probe %d (sleep units %d): %s" % (probe & 0xffff, probe >> 16, " ".join("%.3f" % t for t in ts)), flush=True)
