"""Single-rank nccl (= RCCL) smoke of the device-resident trace path of the sharded temporal update: _stitch_distributed -> DeviceTraces ->
bind device-to-device -> a full iteration that consumes it (BoundRows on the device, row means on the device).  Run on a GPU box:
python scripts/nccl_smoke.py"""
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
from cnmf_e_amd.engine import Engine, DeviceTraces
from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
d1, d2, T, K, r = 96, 80, 400, 12, 6
f = synth.make_factors(d1, d2, T, K, 4, gSig=1.5, gSiz=7, min_sep=5)
Y = synth.make_video(f, np.float32)
def build():
    eng = Engine(0)
    v = PatchedVideo(d1, d2, T, [48, 40], r, eng)
    v.upload_from_full(Y)
    return Sources2D(v, Options(ring_radius=r), f.A_init, f.C_init, f.sn, dist_group=td.group.WORLD)
ref = build(); dev = build()
for s in (ref, dev):
    s.update_background_parallel(); s.update_spatial_parallel()
# reference: the plain single-rank temporal update (host stitch); dev: the same pieces through _stitch_distributed on nccl
ref.update_temporal_parallel()
pieces = []
import types
orig = dev.engine.hals_temporal
A_csr = dev.A.tocsr()
for idx in dev.video.owned:
    pp, bp = dev.video.patch_pix[idx], dev.video.block_pix[idx]
    indp, A_prev_b = dev._prev_block_of(idx)
    dev._residual(idx, A_prev_b if indp.size else None, dev._rows(dev.C_prev, indp) if indp.size else None)
    ind = np.nonzero(np.asarray(A_csr[bp].sum(axis=0)).ravel() > 0)[0]
    if ind.size == 0: continue
    _, C_raw_p, aa_p = orig(dev.video.pid[idx], A_csr[pp][:, ind].tocsc(), dev._rows(dev.C, ind), dev.options.maxIter, want_C=False)
    pieces.append((ind, C_raw_p, aa_p))
C_dev = dev._stitch_distributed(pieces, K, T)
assert isinstance(C_dev, DeviceTraces), type(C_dev)
dev.C_raw = C_dev; dev.C = C_dev; dev._bind_C(); dev._update_b0_new()
assert C_dev._host is None
print("stitched on device; max |C_dev - C_ref| / max|C| = %.2e" % (np.abs(np.asarray(C_dev) - ref.C).max() / np.abs(ref.C).max()))
C_dev._host = None                                          # drop the host copy again: the next iteration must not need it
for s in (ref, dev):
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
print("iteration on the device-resident C: max |dC| / max|C| = %.2e, host copy made: %s" % (np.abs(np.asarray(dev.C) - np.asarray(ref.C)).max() / np.abs(np.asarray(ref.C)).max(), C_dev._host is not None))
b = dev.b0_new
print("b0_new ok", b.shape)
# the collective branches of the three methods themselves, taken with one rank (force_collectives): all-gather of A over RCCL, device-side
# stitch + all-reduce of C_raw, lazy all-reduces of b0 / Ymean -- two full iterations against the plain single-rank run
ref2 = build(); frc = build(); frc.force_collectives = True
for it in range(2):
    for s in (ref2, frc):
        s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
assert isinstance(frc.C, DeviceTraces), type(frc.C)
eA = abs(frc.A - ref2.A).max() / abs(ref2.A).max()
eC = np.abs(np.asarray(frc.C) - np.asarray(ref2.C)).max() / np.abs(np.asarray(ref2.C)).max()
eb = np.abs(frc.b0_new - ref2.b0_new).max() / np.abs(ref2.b0_new).max()
print("forced collectives, 2 iterations: max |dA| %.2e  |dC| %.2e  |db0_new| %.2e (relative)" % (eA, eC, eb))
assert eA < 1e-4 and eC < 1e-4 and eb < 1e-4
rss_f, _ = frc.compute_RSS(); rss_r, _ = ref2.compute_RSS()
assert abs(rss_f - rss_r) <= 1e-4 * rss_r, (rss_f, rss_r)
print("compute_RSS ok", rss_f)
td.destroy_process_group()
