"""world_size-2 gloo test of the N>1 path (patch sharding + the temporal all-reduce stitch + the spatial
row gather) on CPU.  Kernels are a test double (tests/fake_engine.py) because the HIP engine needs a GPU;
what is under test is cnmf_e_amd.sources2d's distributed host logic."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = dict(d1=40, d2=44, T=150, K=6, r=4, seed=31)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out, update_sn, lazy=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    from fake_engine import FakeEngine, LazyFakeEngine, LateFakeEngine
    c = CASE
    f = synth.make_factors(c["d1"], c["d2"], c["T"], c["K"], c["seed"], gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(c["d1"], c["d2"], c["T"], [20, 22], c["r"], (LateFakeEngine() if lazy == "late" else LazyFakeEngine()) if lazy else FakeEngine(), rank=rank, world_size=world)
    assert len(video.owned) == 4 // world
    video.upload_from_full(Y.astype(np.float64))
    s = Sources2D(video, Options(ring_radius=c["r"], spatial_algorithm="hals", maxIter=3), f.A_init, f.C_init, f.sn,
                  dist_group=td.group.WORLD)
    s.update_background_parallel(); s.update_spatial_parallel(update_sn=update_sn); s.update_temporal_parallel()
    if rank == 0:
        np.savez(out, A=s.A.toarray(), C=s.C, b0_new=s.b0_new, sn=s.P["sn"])
    td.barrier()
    td.destroy_process_group()


import pytest


@pytest.mark.parametrize("update_sn,lazy", [(False, False), (True, False), (False, True), (False, "late")])
def test_two_rank_sharded_iteration_matches_single_process(tmp_path, update_sn, lazy):
    """lazy: the engine double defers its results like the real engine, so the sharded run takes the orderings it takes under RCCL; "late": also the queued
    downloads collected late and the temporal jobs swept together (two patches per rank here), in front of the collectives"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, update_sn, lazy), nprocs=2, join=True)
    got = np.load(out)
    c = CASE
    f = synth.make_factors(c["d1"], c["d2"], c["T"], c["K"], c["seed"], gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    o = orc.OracleSources2D(Y.T.reshape(c["d1"], c["d2"], c["T"], order="F"), c["d1"], c["d2"], c["T"], [20, 22], c["r"],
                            f.A_init.astype(np.float32), f.C_init, f.sn, spatial_algorithm="hals", maxIter=3)
    o.update_background_parallel(); o.update_spatial_parallel(update_sn=update_sn); o.update_temporal_parallel()
    if update_sn:
        assert np.allclose(got["sn"], np.asarray(o.sn).reshape(-1, order="F"), rtol=1e-5)
    assert np.allclose(got["A"], o.A.toarray(), rtol=1e-4, atol=1e-6)
    assert np.allclose(got["C"], o.C, rtol=1e-4, atol=1e-4)
    assert np.allclose(got["b0_new"], o.b0_new, rtol=1e-5, atol=1e-2)


def _deconv_run(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    group = None
    if world > 1:
        import torch.distributed as td
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        td.init_process_group("gloo", rank=rank, world_size=world)
        group = td.group.WORLD
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    from fake_engine import FakeEngine
    d1, d2, T, K, r = 30, 28, 160, 4, 5
    f = synth.make_factors(d1, d2, T, K, 7, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    video = PatchedVideo(d1, d2, T, [15, 14], r, FakeEngine(), rank=rank, world_size=world)
    video.upload_from_full(Y.astype(np.float64))
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm="hals", maxIter=2, deconv_flag=True), f.A_init, f.C_init, f.sn, dist_group=group)
    s.update_background_parallel(); s.update_spatial_parallel(); s.update_temporal_parallel()
    res = dict(C=np.asarray(s.C), C_raw=np.asarray(s.C_raw), S=np.asarray(s.S), kp=np.asarray(s.P["kernel_pars"]), sn=np.asarray(s.P["neuron_sn"]))
    if world > 1:
        if rank == 0:
            np.savez(out, **res)
        td.barrier(); td.destroy_process_group()
    return res


def test_two_rank_sharded_deconvolution_matches_single_process(tmp_path):
    """deconv_flag=true over 2 ranks: the per-patch HALS+OASIS sweeps, the all-reduce stitch of C_raw (:269-280) and deconvTemporal with its
    rows sharded over the ranks give what one process gives"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npz")
    mp.spawn(_deconv_run, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    ref = _deconv_run(0, 1, 0, None)
    for k in ("C", "C_raw", "S", "kp", "sn"):
        assert np.allclose(got[k], ref[k], rtol=1e-6, atol=1e-6), k
    assert (ref["S"] > 0).any()


def _gather_run(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    import scipy.sparse as sp
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    from fake_engine import FakeEngine
    d1, d2, T, K, r = 30, 28, 40, 4, 5
    f = synth.make_factors(d1, d2, T, K, 7, gSig=1.5, gSiz=7, min_sep=5)
    video = PatchedVideo(d1, d2, T, [15, 14], r, FakeEngine(), rank=rank, world_size=world)
    video.upload_from_full(synth.make_video(f, np.float32).astype(np.float64))
    s = Sources2D(video, Options(ring_radius=r), f.A_init, f.C_init, f.sn, dist_group=td.group.WORLD)
    d = d1 * d2
    rng = np.random.default_rng(3)
    caps, ok = [], True
    # per call: how many entries each rank contributes (disjoint rows: rank 0 the even pixels, rank 1 the odd ones).  Growth beyond the remembered capacity on ONE
    # rank only, an empty contribution, a shrinking one
    for n0, n1 in [(5, 9), (1000, 3), (0, 0), (40, 1600), (7, 7), (0, 1680)]:
        full = []
        for rk, n in ((0, n0), (1, n1)):
            rows = rng.choice(d // 2, size=min(n, d // 2), replace=False) * 2 + rk
            cols = rng.integers(0, K, size=rows.size)
            if n > d // 2:                                              # more entries than pixels of a parity: several columns per row
                rows = np.repeat(np.arange(rk, d, 2), K)[:n]; cols = np.tile(np.arange(K), d // 2)[:n]
            full.append((rows, cols, rng.random(rows.size).astype(np.float32) + 0.5))
        mine = full[rank]
        got = s._gather_sparse(sp.csc_matrix((mine[2], (mine[0], mine[1])), shape=(d, K)))
        want = sp.csc_matrix((np.concatenate([x[2] for x in full]), (np.concatenate([x[0] for x in full]), np.concatenate([x[1] for x in full]))), shape=(d, K))
        ok = ok and got.shape == want.shape and (got != want).nnz == 0
        caps.append(int(s._gather_cap))
    if rank == 0:
        np.savez(out, ok=ok, caps=np.array(caps))
    td.barrier(); td.destroy_process_group()


def test_gather_of_footprint_rows_in_one_round_with_a_remembered_capacity(tmp_path):
    """_gather_sparse sends one padded block per rank and call; the capacity follows the gathered counts (the same number on every rank), a rank that outgrew it
    makes everybody repeat the round, empty contributions and shrinking ones pass"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "g.npz")
    mp.spawn(_gather_run, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert bool(got["ok"])
    caps = got["caps"].tolist()
    assert caps == [256, 1280, 1280, 2048, 2048, 2304], caps


def test_bench_spawns_its_own_ranks_and_refuses_without_gpus():
    """`python bench.py --gpus N` with no launcher in the environment starts N ranks itself (it used to run ONE rank silently).
    CNMFE_BENCH_DRY=1 stops after the rendezvous and one all-reduce (gloo here: no GPU), so this checks the env plumbing and the relayed line;
    without it, on a box with fewer GPUs than ranks, the launcher refuses loudly."""
    import json, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, CNMFE_BENCH_DRY="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["rccl_ranks"] == 2 and line["sum"] == 2.0
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        assert r.returncode != 0 and b"GPU(s) visible" in r.stderr
