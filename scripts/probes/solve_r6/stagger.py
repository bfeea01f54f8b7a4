"""De-phasing the two waves of a SIMD at the start of the ring solve (round 6): does the kernel lose time because the two waves of a SIMD load together and then
compute together?  The experiment needs one line at the top of k_ring_solve8 (cnmf_e_amd/csrc/ring_solve_staged.hpp), not kept in the tree:
    if ((probe & 8192) && (blockIdx.x & 1) && blockIdx.x < 4096) { for (int i = 0; i < (probe >> 16); ++i) __builtin_amdgcn_s_sleep(127); }
Result at H (three first-run fits each): 5.995 / 5.996 / 6.033 ms without, 6.03 / 6.06 / 6.03 / 6.04 ms with 1 / 2 / 3 / 5 sleep units for every other workgroup of the first
two rounds: no effect -- the waves de-phase by themselves."""
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
