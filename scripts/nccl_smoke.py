"""Single-rank nccl (= RCCL) smoke of the sharded code paths: with force_collectives the three update methods take their collective branches on
a real RCCL group of one rank -- all-gather of the rows of A, the second all-gather of the post-processed columns, the in-place all-reduce of
the engine's stitch accumulator (device buffer handed to torch.distributed without a copy), the lazy all-reduces of b0 / Ymean -- and must
reproduce the plain single-rank run.  Run on a GPU box:  python scripts/nccl_smoke.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cnmf_e_amd import synth
from cnmf_e_amd.engine import Engine
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 96, 80, 400, 12, 6
f = synth.make_factors(d1, d2, T, K, 4, gSig=1.5, gSiz=7, min_sep=5)
Y = synth.make_video(f, np.float32)
def build(**kw):
    eng = Engine(0)
    v = PatchedVideo(d1, d2, T, [48, 40], r, eng)
    v.upload_from_full(Y)
    return Sources2D(v, Options(ring_radius=r, **kw), f.A_init, f.C_init, f.sn, dist_group=td.group.WORLD)
for kw in ({}, {"deconv_flag": True}):
    ref, frc = build(**kw), build(**kw)
    frc.force_collectives = True
    for it in range(2):
        for s in (ref, frc):
            s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
    eA = abs(frc.A - ref.A).max() / abs(ref.A).max()
    eC = np.abs(np.asarray(frc.C) - np.asarray(ref.C)).max() / np.abs(np.asarray(ref.C)).max()
    eb = np.abs(frc.b0_new - ref.b0_new).max() / np.abs(ref.b0_new).max()
    print("forced collectives %s, 2 iterations: max |dA| = %.2e  |dC| = %.2e  |db0_new| = %.2e (relative)" % (kw, eA, eC, eb))
    assert eA < 1e-6 and eC < 1e-6 and eb < 1e-6
    if not kw:
        rss_f, _ = frc.compute_RSS(); rss_r, _ = ref.compute_RSS()
        assert abs(rss_f - rss_r) <= 1e-6 * rss_r, (rss_f, rss_r)
        print("compute_RSS ok", rss_f)
td.destroy_process_group()
