#!/usr/bin/env python
"""bench.py -- CNMF-E iterations/s (background + spatial + temporal) on MI355X.

One "step" = Sources2D.update_background_parallel + update_spatial_parallel + update_temporal_parallel
(demos/demo_large_data_1p.m:199-201) on synthetic data already resident in HBM; value = whole-FOV iterations per second.
N=1 (default --config c3): BASELINE.json configs[2] (headline): 512x512x10000 fp32, K=500, ring_radius=15, 1 patch.
N>1 (default --config c4): BASELINE.json configs[3]: the SAME 512x512x10000, K=500 video cut into the 4 x 4 patches of
distribute_data.m:56-79,165-171 (blocks = patches + ring halo), patches round-robin over the ranks (strong scaling: the total work is
fixed); spatial rows are all-gathered, the temporal update does one RCCL all-reduce of the [K x T] stitch
(update_temporal_parallel.m:269-280).  `--config c4 --gpus 1` runs all 16 patches on one GPU.
--weak: the round-1 grown-FOV mode (512 x 512N, one 512x512 patch and 500 neurons per rank; value = patch-iterations/s).
--demo-sequence: the calls demo_large_data_1p.m:142-211 makes on this path, from a fresh upload, as seconds per recording.

`python bench.py --gpus N` without a launcher's WORLD_SIZE spawns the N ranks itself (one per GPU, nccl = RCCL; it refuses to run when fewer than
N GPUs are visible); under `python -m torch.distributed.run` it takes RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
At N = 1 the default run also times configs[3] (`c4`: the same video as 16 patches on this one GPU) in a child process and reports it as `c4_n1`,
the N = 1 point of the strong-scaling curve the N > 1 runs continue.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: d1, d2(per GPU), T, K(per GPU), ring radius, seed
    "c3": (512, 512, 10000, 500, 15, 2),     # BASELINE.json configs[2] (headline)
    "c2": (256, 256, 3000, 200, 15, 1),      # BASELINE.json configs[1]
    "tiny": (96, 96, 600, 18, 15, 5),
    "c4": (512, 512, 10000, 500, 15, 2),     # BASELINE.json configs[3]: c3's video, 4 x 4 patches sharded over the ranks
    "c4tiny": (96, 96, 600, 18, 15, 5),      # (test size of the c4 code path: 2 x 2 patches of 48 x 48)
    "c5shard": (1024, 1024, 20000, 2000, 15, 3),   # BASELINE.json configs[4]: ONE rank's share (rank 0 of 8: 8 of the 8 x 8 patches) on one GPU, video uploaded as fp16
}
PATCHES = {"c4": [128, 128], "c4tiny": [48, 48], "c5shard": [128, 128]}
SHARD_OF = {"c5shard": 8}    # configurations that run one rank's patches of an N-rank decomposition without the collectives (a per-rank load figure, not a scaling point)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # v_mfma_f64_16x16x4_f64: half the fp32 matrix rate (157.3 TF), spec
F32_VECTOR_PEAK_TF = 157.3   # fp32 vector (= fp32 matrix) peak, spec
F32_MFMA_PEAK_TF = 157.3
I8_MFMA_PEAK_TOPS = 5000.0   # dense int8 MFMA, spec (MI355X_MICROARCH.md measures >= 3944 TOPS on 16x16x64); the int8 Gram issues 13 digit-pair products per fp32-equivalent product


def _blas_info():
    try:
        import threadpoolctl
        info = threadpoolctl.threadpool_info()
        thr = max([p.get("num_threads", 1) for p in info] or [1])
        blas = ", ".join(sorted({"%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in info if p.get("user_api") == "blas"})) or "unknown BLAS"
    except Exception:
        thr, blas = os.cpu_count() or 1, "unknown BLAS"
    return int(thr), blas


def cpu_baseline():
    """The float64 NumPy restatement of the reference (oracle/, 'port': NOT MATLAB) on this host.  `value` is what THIS run times: a bounded sample (one full
    iteration at 128 x 128 x 3000 with the headline's neuron density, ~12 s) scaled by d*T to the headline workload.  The headline workload itself as
    `bench.py --cpu-baseline full` measured it once on the GPU box's host (committed: profiles/r03/cpu_baseline_full.json -- spatial and
    temporal updates in full, the per-pixel background regression on every 64th pixel and extrapolated) is carried as `full_run`, same thread count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    full = None
    fpath = os.path.join(ROOT, "profiles", "r03", "cpu_baseline_full.json")
    if os.path.exists(fpath):
        try:
            full = json.load(open(fpath))
        except Exception:
            full = None
    threads = int(full["cores"]) if full else None          # the same BLAS thread count as the full run: one core count in the whole object
    d1, d2, T, r = 128, 128, 3000, 15                  # ~12 s of host work on the GPU box
    K = max(2, int(round(500 * d1 * d2 / (512.0 * 512.0))))
    f = synth.make_factors(d1, d2, T, K, 9)
    Y = synth.make_video(f, np.float32)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [d1, d2], r, f.A_init.astype(np.float32), f.C_init, f.sn,
                            spatial_algorithm="hals", maxIter=5)
    def run():
        t0 = time.time()
        o.update_background_parallel(); o.update_spatial_parallel(); o.update_temporal_parallel()
        return time.time() - t0
    try:
        import threadpoolctl
        if threads:
            with threadpoolctl.threadpool_limits(limits=threads):
                dt = run()
        else:
            dt = run()
    except ImportError:
        dt = run()
    scale = (512.0 * 512.0 * 10000.0) / (d1 * d2 * T)
    thr, blas = _blas_info()
    used = min(threads, thr) if threads else thr
    sample = {"seconds": dt, "workload": "%dx%dx%d, K=%d, r=%d, one full iteration" % (d1, d2, T, K, r), "scaled_by_dT": scale,
              "iter_per_s_scaled_to_c3": 1.0 / (dt * scale), "cores": used}
    # ADVICE r4 / verdict r4 #8: `value` is what THIS run measured (the bounded sample, scaled by d*T -- said so in `unit`); the one-off full-size run on the GPU box's
    # host (committed) rides beside it as `full_run`, with its own rate
    out = {"value": 1.0 / (dt * scale), "unit": "iter/s (512x512x10000-equivalent: one 128x128x3000 iteration of this run scaled by the d*T work ratio -- an extrapolation)",
           "cores": used, "kind": "port", "blas": blas, "host_cpus": os.cpu_count(), "measured_in_this_run": True,
           "sample": "one full iteration of the float64 NumPy restatement (oracle/cnmfe_oracle.py, not MATLAB) on %dx%dx%d, K=%d, r=%d took %.2f s on %d BLAS threads; "
                     "scaled by the d*T work ratio %.1f to 512x512x10000" % (d1, d2, T, K, r, dt, used, scale), "in_run_sample": sample}
    if full and "c3" in full:
        c3 = full["c3"]
        out["full_run"] = dict(full, source="profiles/r03/cpu_baseline_full.json", measured_in_this_run=False, c3_iter_per_s=c3["iter_per_s"],
                               note="C3 at full size by `bench.py --cpu-baseline full` on the GPU box's host in round 3: %.0f s per iteration = background %.0f s (set-up %.0f s + 64 x "
                                    "%.1f s of the per-pixel loop timed on every 64th pixel: extrapolated) + spatial %.0f s + temporal %.0f s; K-dependent terms (the reference's dense "
                                    "Y*C', A*C) do not scale by d*T alone, which is why it is slower than the scaled sample"
                                    % (c3["iteration_s"], c3["background_extrapolated_s"], c3["background_setup_s"], c3["background_loop_sampled_s"], c3["spatial_s"], c3["temporal_s"]))
    return out


def cpu_baseline_full(which=("c2", "c3"), out_path=None):
    """the float64 NumPy restatement ('CPU restatement, not MATLAB') at full size on this host's cores.
    C2 (256x256x3000, K=200): one iteration timed in full.  C3 (512x512x10000, K=500): spatial + temporal updates in full, the background
    regression (fit_ring_model.m:92-108, an interpreted loop over 262144 pixels) on a fixed 1/64 pixel sample and scaled x64 -- an extrapolation,
    stated as such.  Minutes of host work and ~70 GB of host memory for C3: run once, commit the JSON."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cnmfe_oracle as orc
    from cnmf_e_amd import synth
    thr, blas = _blas_info()
    res = {"kind": "port", "what": "CPU restatement (not MATLAB): oracle/cnmfe_oracle.py, float64 NumPy/SciPy", "cores": thr, "blas": blas, "host_cpus": os.cpu_count()}
    for name in which:
        d1, d2, T, K, r, seed = CONFIGS[name]
        f = synth.make_factors(d1, d2, T, K, seed)
        Y = synth.make_video(f, np.float32)
        o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [d1, d2], r, f.A_init.astype(np.float32), f.C_init, f.sn,
                                spatial_algorithm="hals", maxIter=5)
        del Y
        t = {}
        t0 = time.time()
        if name == "c3":
            rows = np.arange(0, d1 * d2, 64)                   # fixed 1/64 pixel sample of the per-pixel regressions
            o.bg_only_rows = rows
            o.update_background_parallel()
            t["background_sampled_s"] = time.time() - t0
            loop = float(getattr(orc.fit_ring_model, "last_loop_seconds", t["background_sampled_s"]))
            t["background_loop_sampled_s"] = loop                 # the per-pixel regressions of the sample
            t["background_setup_s"] = t["background_sampled_s"] - loop          # Ymean, Bf = Y - A*C, ... : done once whatever the sample
            t["background_extrapolated_s"] = t["background_setup_s"] + 64.0 * loop
            t["background_note"] = ("regression loop (fit_ring_model.m:92-108) on %d of %d pixels (every 64th): set-up %.1f s once + 64 x %.1f s of loop "
                                    "-- an extrapolation, not a measurement" % (rows.size, d1 * d2, t["background_setup_s"], loop))
            if os.environ.get("CNMFE_CPU_BG_ONLY", "0") == "1":
                res[name] = t
                if out_path:
                    with open(out_path, "w") as fh:
                        json.dump(res, fh, indent=1)
                continue
        else:
            o.update_background_parallel()
            t["background_s"] = time.time() - t0
        t1 = time.time(); o.update_spatial_parallel(); t["spatial_s"] = time.time() - t1
        t2 = time.time(); o.update_temporal_parallel(); t["temporal_s"] = time.time() - t2
        total = t.get("background_s", t.get("background_extrapolated_s")) + t["spatial_s"] + t["temporal_s"]
        t["iteration_s"] = total; t["iter_per_s"] = 1.0 / total
        t["workload"] = "%s: %dx%dx%d, K=%d, r=%d, 1 patch, seed %d, deconv_flag=false" % (name, d1, d2, T, K, r, seed)
        res[name] = t
        del o
        if out_path:                                            # every finished configuration is on disk at once (C3 takes many minutes)
            with open(out_path, "w") as fh:
                json.dump(res, fh, indent=1)
    return res


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher: start the N ranks here (one per GPU, RANK = LOCAL_RANK, rendezvous on 127.0.0.1) and relay
    rank 0's line.  Refuses when fewer than N GPUs are visible -- N ranks on fewer devices is not the run that was asked for."""
    import socket
    one_dev = os.environ.get("CNMFE_BENCH_ONE_DEVICE", "0") == "1" or os.environ.get("CNMFE_BENCH_DRY", "0") == "1"
    if not one_dev:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (no WORLD_SIZE in the environment, so the ranks would be spawned here)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    if any(rcs):
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise SystemExit("bench.py --gpus %d: rank exit codes %s" % (n, rcs))
    lines = [l for l in out.decode().splitlines() if l.lstrip().startswith("{")]     # (gloo / RCCL banners may share rank 0's stdout)
    if not lines:
        raise SystemExit("bench.py --gpus %d: rank 0 printed no JSON line" % n)
    print(lines[-1], flush=True)


def emit(obj):
    """the ONE JSON line, as the last thing on stdout: what C libraries still hold in stdio's buffer (RCCL's banner under NCCL_DEBUG=VERSION, written when the communicator
    was made and otherwise flushed at exit, BEHIND the line) goes out first"""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(obj), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, help="c3 (default at 1 GPU) | c4 (default at N > 1) | c2 | tiny | c4tiny | c5shard (rank 0 of 8 of configs[4] on one GPU)")
    ap.add_argument("--weak", action="store_true", help="grown-FOV weak scaling (512 x 512N, one patch per rank) instead of the sharded 4 x 4 patches")
    ap.add_argument("--demo-sequence", action="store_true", help="time the call sequence of demo_large_data_1p.m:142-211 from a fresh upload")
    ap.add_argument("--alg", default="hals", help="spatial algorithm: hals | hals_thresh | nnls")
    ap.add_argument("--deconv", action="store_true", help="deconv_flag=true (OASIS AR(1) FOOPSI inside the temporal sweep); reported separately")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="short", help="short (default: bounded sample, scaled) | full (BASELINE.md section 3: C2 in full, C3 with a 1/64 "
                                                            "pixel sample of the background regression; minutes of host work, no GPU step) | none")
    ap.add_argument("--no-extras", action="store_true", help="skip the c4_n1 child run and the in-run PMC traffic passes (what the child runs pass)")
    ap.add_argument("--bg-ssub", type=int, default=1, help="options.bg_ssub (the shipped demo uses 2; the headline metric is quoted at 1)")
    a = ap.parse_args()
    if a.cpu_baseline == "full":
        which = tuple(a.config.split(",")) if a.config else ("c2", "c3")
        print(json.dumps(cpu_baseline_full(which, os.environ.get("CNMFE_CPU_BASELINE_OUT"))))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus, sys.argv[1:])
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    # test hook for a 1-GPU box: CNMFE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two ranks
    # on one device); the driver's multi-GPU runs leave it unset and get one rank per GPU over nccl (= RCCL)
    one_dev = os.environ.get("CNMFE_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        local = 0
    dry = os.environ.get("CNMFE_BENCH_DRY", "0") == "1"
    # torch's CPU kernels are not on this path, and an OpenMP team over every core of the host (one per rank), spinning behind any stray CPU tensor operation, is
    # what throttled a CPU-quota'd container for 40 ms of every 100 (profiles/r04/forced_collectives.txt)
    torch.set_num_threads(1)
    if not dry:
        torch.cuda.set_device(local)
    group = None
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_dev or (dry and not torch.cuda.is_available()):
            one_dev = True
            td.init_process_group(backend="gloo")
        else:
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        group = td.group.WORLD
        if os.environ.get("CNMFE_BENCH_DRY", "0") == "1":        # launcher test (no GPU work): rendezvous + one collective, then the line
            t_ = torch.ones(1) if one_dev else torch.ones(1, device="cuda")
            td.all_reduce(t_)
            if rank == 0:
                print(json.dumps({"dry": True, "rccl_ranks": td.get_world_size(group), "backend": td.get_backend(group), "sum": float(t_.item())}))
            td.destroy_process_group()
            return
    # test hook for a 1-GPU box: the sharded code path -- every collective branch of the three methods -- on a real RCCL group of ONE rank
    # (CNMFE_BENCH_FORCE_COLLECTIVES=1 python bench.py --config c4): what the collectives' host side costs per iteration, without a second GPU
    force_coll = world == 1 and os.environ.get("CNMFE_BENCH_FORCE_COLLECTIVES", "0") == "1" and not dry
    if force_coll:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        group = td.group.WORLD
    comm = {"rccl_ranks": 1, "backend": None}
    if group is not None:
        import torch.distributed as td
        comm = {"rccl_ranks": int(td.get_world_size(group)), "backend": str(td.get_backend(group))}

    from cnmf_e_amd import synth
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options

    if a.config is None:
        a.config = "c3" if (world == 1 or a.weak) else "c4"
    sharded_fov = a.config in PATCHES                         # one FOV cut into patches that are sharded over the ranks (strong scaling)
    if sharded_fov and a.weak:
        raise SystemExit("--weak grows the FOV of a one-patch configuration; %s is a patched FOV" % a.config)
    if world > 1 and not sharded_fov and not a.weak:
        raise SystemExit("--gpus %d needs a patched configuration (c4) or --weak" % world)
    d1, d2p, T, Kp, r, seed = CONFIGS[a.config]
    d2, K = (d2p, Kp) if sharded_fov else (d2p * world, Kp * world)
    f = synth.make_factors(d1, d2, T, K, seed)
    eng = Engine(local)
    # several patches per rank: their calls alternate between two execution lanes (streams + scratch sets, cnmfe_set_option "lanes") -- small patches' kernels are
    # a few workgroups each with a dispatch latency between two dependent ones; CNMFE_BENCH_LANES=1 gives the one-stream figure
    per_rank = 1 if not sharded_fov else -(-(-(-d1 // PATCHES[a.config][0]) * -(-d2 // PATCHES[a.config][1])) // (SHARD_OF.get(a.config, 0) or world))
    lanes = int(os.environ.get("CNMFE_BENCH_LANES", str(max(1, min(3, per_rank)))))      # (three lanes measured best for 8 - 16 patches on one GPU, profiles/r06/lanes.txt)
    if lanes > 1:
        eng.set_option("lanes", lanes)
    shard_of = SHARD_OF.get(a.config, 0)
    if shard_of and world > 1:
        raise SystemExit("%s is one rank's share of a %d-rank run on ONE GPU" % (a.config, shard_of))
    video = PatchedVideo(d1, d2, T, PATCHES[a.config] if sharded_fov else [d1, d2p], r, eng, rank=0 if shard_of else rank, world_size=shard_of or world)
    for idx in video.owned:                                   # synthesise each owned block directly in HBM
        Yb = synth.make_video_device(f, "cuda:%d" % local, pixels=video.block_pix[idx])
        torch.cuda.synchronize()
        if shard_of:                                          # the file's samples are fp16 (configs[4]): uploaded as such, widened on the device
            from cnmf_e_amd import _lib as L_
            Yh = Yb.half(); del Yb
            torch.cuda.synchronize()
            eng.upload_block_device(video.pid[idx], Yh.data_ptr(), T, dtype=L_.F16)
            Yb = Yh
        else:
            video.upload_block_device(idx, Yb.data_ptr())
        del Yb
    torch.cuda.empty_cache()
    s = Sources2D(video, Options(ring_radius=r, spatial_algorithm=a.alg, maxIter=5, deconv_flag=a.deconv, bg_ssub=a.bg_ssub),
                  f.A_init, f.C_init, f.sn, dist_group=group)
    if force_coll:
        s.force_collectives = True
        comm["forced_collectives"] = True
    eng.profile(True)

    last = {}
    def step():
        last["bg"] = s.update_background_parallel()
        s.update_spatial_parallel()
        s.update_temporal_parallel()

    def fence():
        if world > 1:
            import torch.distributed as td
            td.barrier()
        torch.cuda.synchronize()

    if a.demo_sequence:
        # demo_large_data_1p.m:142-211 on this path: background; spatial (update_sn); temporal x2; spatial_algorithm = 'nnls'; background;
        # spatial; temporal; and the K-changed branch (:205-209) spatial; temporal  --  2 background + 3 spatial + 4 temporal updates
        fence()
        t0 = time.perf_counter()
        marks = []
        def call(name, fn, *aa, **kw):
            r_ = fn(*aa, **kw); marks.append((name, round(1e3 * (time.perf_counter() - t0), 2))); return r_      # (host return times, not fenced)
        call("bg", s.update_background_parallel); call("spatial_sn", s.update_spatial_parallel, update_sn=True)
        call("temporal", s.update_temporal_parallel); call("temporal", s.update_temporal_parallel)
        s.options.spatial_algorithm = "nnls"
        i2 = call("bg", s.update_background_parallel); call("spatial", s.update_spatial_parallel); call("temporal", s.update_temporal_parallel)
        call("spatial", s.update_spatial_parallel); call("temporal", s.update_temporal_parallel)
        fence()
        dt = time.perf_counter() - t0
        marks.append(("fence", round(1e3 * dt, 2)))
        if rank == 0:
            tab = eng.profile_table()
            print(json.dumps({"metric": "cnmfe_demo_sequence_seconds", "value": dt, "unit": "s per recording", "n_gpus": world, "higher_is_better": False,
                              "config": {"workload": "%s: %dx%dx%d, K=%d, %d patches" % (a.config, d1, d2, T, K, len(video.order)),
                                         "sequence": "bg, spatial(update_sn), temporal, temporal, [nnls] bg, spatial, temporal, spatial, temporal (demo_large_data_1p.m:142-211), fresh upload"},
                              "host_return_ms": marks,
                              "second_fit": {k: int(v) if not isinstance(v, bool) else v for k, v in (i2.get(video.owned[0]) or {}).items()},
                              "kernels_ms_total": {k: round(v["total_ms"], 3) for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["total_ms"]) if v["total_ms"] > 0.05}}))
        eng.close()
        return
    fence()
    warm_steps_ms = []
    first_tab = None
    for _ in range(a.warmup):                                   # every warm-up step fenced and timed on its own: the FIRST one is the cold iteration
        tw0 = time.perf_counter()
        step()
        fence()
        warm_steps_ms.append(1e3 * (time.perf_counter() - tw0))
        if first_tab is None:
            first_tab = eng.profile_table()
    warm_tab = eng.profile_table()
    # timed region: HIP events only around the kernels a roofline is quoted for (an event pair around EVERY launch costs host and device time per
    # launch -- 8 ms per iteration with the ~2000 small launches of c4, ~1 % at c3); the per-kernel breakdown comes from EXTRA steps after it
    eng.profile(2)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tab_timed = eng.profile_table()
    XSTEPS = 2
    eng.profile(1)
    eng.profile_reset()
    for _ in range(XSTEPS):                                     # (every rank: the steps hold the collectives)
        step()
    fence()
    # The iteration no longer runs the ring sweep (sweep-free residual): the kernel the north star names is timed on its own, in this
    # run, on this video -- a few separate launches per owned patch with the engine switched to the swept residual, HIP events on the engine's stream.
    r1_sep = None
    if a.bg_ssub == 1 and os.environ.get("CNMFE_BENCH_R1", "1") != "0":
        before = {k: dict(v) for k, v in eng.profile_table().items()}
        prev_opts = {k: eng.get_option(k, 1) for k in ("r1_virtual", "r1_delta")}       # (a CNMFE_OPTS preset survives this block)
        eng.set_option("r1_virtual", 0); eng.set_option("r1_delta", 0)
        NR1 = 5 if len(video.owned) == 1 else 1
        for idx in video.owned:
            for _ in range(NR1 + 1):                            # (+1: the first launch also allocates the Ysig buffer)
                eng.residual(video.pid[idx], None, None)
        eng.synchronize()
        for k_, v_ in prev_opts.items():
            eng.set_option(k_, v_)
        after = eng.profile_table()
        r1_sep = {}
        for k, v in after.items():
            if k.startswith("residual_r1") or k == "r1_dlt":
                c0 = before.get(k, {"calls": 0, "total_ms": 0.0})
                if v["calls"] > c0["calls"]:
                    r1_sep[k] = {"calls": v["calls"] - c0["calls"], "total_ms": v["total_ms"] - c0["total_ms"]}
    if rank == 0 and os.environ.get("CNMFE_BENCH_DUMP"):       # (tests: the traces after the last step, to compare runs with different rank counts)
        np.save(os.environ["CNMFE_BENCH_DUMP"], np.asarray(s.C))
    if world > 1:
        import torch.distributed as td
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return
    n_patches = video.nr_patch * video.nc_patch
    value = (n_patches if a.weak else 1) * a.steps / dt          # whole-FOV iterations/s (weak mode: patch-iterations/s, every patch a full 512 x 512 FOV)
    tab_x = eng.profile_table()
    if r1_sep:                                                  # the separate sweep launches are not part of the extra steps
        for k, v in r1_sep.items():
            tab_x[k] = {"calls": tab_x[k]["calls"] - v["calls"], "total_ms": tab_x[k]["total_ms"] - v["total_ms"]}
    kern = {k: {"ms_per_call": v["total_ms"] / v["calls"], "calls_per_step": v["calls"] / float(XSTEPS),
                "ms_per_step": v["total_ms"] / XSTEPS, "from": "extra steps"} for k, v in tab_x.items() if v["calls"] > 0}
    r1_kern = None
    if r1_sep:
        n_ = sum(v["calls"] for k_, v in r1_sep.items() if k_ != "r1_dlt"); ms_ = sum(v["total_ms"] for k_, v in r1_sep.items() if k_ != "r1_dlt")
        r1_kern = {"ms_per_call": ms_ / n_, "calls": n_, "from": "separate launches of the ring sweep after the timed region (the iteration itself runs none)"}
    for k, v in tab_timed.items():                              # the roofline kernels: measured inside the timed region
        if v["calls"]:
            kern[k] = {"ms_per_call": v["total_ms"] / v["calls"], "calls_per_step": v["calls"] / float(a.steps), "ms_per_step": v["total_ms"] / a.steps, "from": "timed region"}
    dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
    # ---- roofline of the dominant kernel; algorithmic work per launch from SURVEY.md section 8(d) ----
    P = {idx: video.patch_pix[idx].size for idx in video.owned}
    B = {idx: video.block_pix[idx].size for idx in video.owned}
    # per LAUNCH: a kernel runs once per owned patch, ms_per_call averages over them -- so do the algorithmic bytes / flops
    d, d_b, p = sum(P.values()) / float(len(P)), sum(B.values()) / float(len(B)), 96 if r == 15 else None
    roof = None
    def r1_roof():
        src = kern.get("residual_r1") or r1_kern                # inside the iteration (r1_virtual = 0) or, by default, the separate launches
        if src is None or a.bg_ssub != 1 or p is None:          # bg_ssub > 1: the sweep runs on the low-resolution patch, a different kernel mix
            return None
        bytes_r1 = 4.0 * d_b * T + 4.0 * d * T + 8.0 * d * p + 4.0 * (Kp if not sharded_fov else Kp * d_b / float(d1 * d2)) * T       # read Y + write Ysig + W + C
        ms = src["ms_per_call"]
        # the sweep is a per-pixel weighted sum of p neighbours per frame (every pixel its own p weights: no shared operand, no matrix form): 2 p d T flops on the
        # fp32 VECTOR pipe.  At p = 96 those take longer at the vector peak (157.3 TFLOP/s) than the bytes take at the HBM peak: both fractions are reported
        flops_r1 = 2.0 * p * d * T
        return {"bound": "hbm", "achieved": bytes_r1 / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": bytes_r1 / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "kernel": "residual_r1", "ms_per_launch": ms,
                "algorithmic_bytes_per_launch": bytes_r1, "algorithmic_flops_per_launch": flops_r1,
                "frac_of_fp32_vector_peak": flops_r1 / ms / 1e9 / F32_VECTOR_PEAK_TF,
                "min_ms_at_hbm_peak": bytes_r1 / (HBM_PEAK_GBS * 1e6), "min_ms_at_vector_peak": flops_r1 / (F32_VECTOR_PEAK_TF * 1e9),
                # every product takes one operand from the LDS halo tile; the duo-role kernel's four centres per thread halve that (resid_duo.hpp: 0.5 reads of 4 B per
                # product), against ~150 TB/s of aggregate ds_read_b128 bandwidth (MI355X_MICROARCH.md, LDS)
                "lds_bytes_per_launch": 0.5 * 4.0 * p * d * T, "frac_of_lds_peak": 0.5 * 4.0 * p * d * T / ms / 1e9 / 150e3,
                "note": "a per-pixel weighted sum of p ring neighbours per frame, every pixel with its own weights: three resources at once -- 0.28 of the HBM peak for its "
                        "bytes, ~0.34 of the fp32 vector peak for its 2 p d T flops, ~0.35 of the LDS read bandwidth for its operands -- none of them the roofline by itself; the "
                        "kernel is not in the iteration since round 4 (r1_virtual)",
                "timed": src["from"], "in_iteration": "residual_r1" in kern}
    def proj_roofs():
        """the two projection kernels (north_star's "residual projections"): S1 U = Ysig*C' on the search mask (HALS_spatial.m:27-32) and T1
        U = A'*Ysig (HALS_temporal.m:48).  Algorithmic bytes = the rows of Ysig a launch needs, once (pixels under the mask / under a footprint,
        x T x 4) + the K x T traces read (S1) or written (T1) + the mask / footprint entries; whole-FOV single patch only."""
        if world != 1 or len(video.owned) != 1:
            return None
        out = {}
        try:
            Acsc = s.A.tocsc()
            IND = s._search_location_owned(Acsc)
            npix = {"spatial_proj_U": int(np.unique(IND.indices).size), "temporal_proj_U": int(np.unique(Acsc.indices).size)}
            nnz = {"spatial_proj_U": int(IND.nnz), "temporal_proj_U": int(Acsc.nnz)}
            npix["spatial_proj_rows"], npix["temporal_proj_rows"] = npix["spatial_proj_U"], npix["temporal_proj_U"]      # bg_ssub > 1, sweep-free: the same rows of the VIDEO
            nnz["spatial_proj_rows"], nnz["temporal_proj_rows"] = nnz["spatial_proj_U"], nnz["temporal_proj_U"]
        except Exception:
            return None
        if a.bg_ssub != 1:
            # bg_ssub > 1 (round 5): the full-resolution rows under the masks / footprints (fp64 sums) and the low-resolution video twice (the table of the spatial
            # update, the B_L panels of the temporal one); the fit's own window projection reads the low-resolution fit patch
            low = 4.0 * (d_b / float(a.bg_ssub ** 2)) * T + 4.0 * K * T
            for name, by in (("spatial_proj_rows", None), ("temporal_proj_rows", None), ("spatial_ptab_proj", low), ("temporal_proj_B", low), ("bg_win_proj", low)):
                if name not in kern:
                    continue
                if by is None:
                    by = 4.0 * npix[name] * T + 4.0 * K * T + 8.0 * nnz[name]
                ms = kern[name]["ms_per_step"]
                out[name] = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": name, "ms_per_update": ms, "algorithmic_bytes_per_update": by}
            return out or None
        # the sweep-free formulation's video passes: the temporal projection through B = A - W'A (one read of the centred block video + the K x T result;
        # its fp64 partial sums per (16x16 block, neuron) are traffic, not algorithmic bytes) and the fit's window projection P = Yc Cc' (one read + the traces)
        for name in ("temporal_proj_B", "bg_win_proj"):
            if name in kern:
                # bytes per sample the kernel is built to read: 4 (fp32, or four digit planes) -- the temporal projection on the int8 pipe reads three of the four
                # planes since round 6 (option proj_i8_planes, default 3): its fraction is quoted for the 3 bytes it needs, not for the 4 of the fp32 video
                planes = 4
                if name == "temporal_proj_B" and eng.get_option("proj_i8", 1) and eng.get_option("proj_i8_planes", 3) < 4:
                    planes = 3
                if name == "bg_win_proj" and eng.get_option("win_i8", 1) and eng.get_option("gram_i8", 1) and eng.get_option("win_i8_planes", 0) != 4 and T >= 2048:
                    planes = 3                                  # (the fit's window projection too, since the end of round 6: option win_i8_planes, default 3)
                by = float(planes) * d_b * T + 4.0 * K * T
                ms = kern[name]["ms_per_step"]                  # (one projection per iteration, possibly split into launches by list length)
                out[name] = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": name, "ms_per_update": ms, "algorithmic_bytes_per_update": by, "bytes_per_video_sample": planes}
        for name in ("spatial_proj_U", "temporal_proj_U"):
            if name not in kern:
                continue
            by = 4.0 * npix[name] * T + 4.0 * K * T + 8.0 * nnz[name]
            ms = kern[name]["ms_per_call"]
            out[name] = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": name, "ms_per_launch": ms, "algorithmic_bytes_per_launch": by, "pixels_read": npix[name], "entries": nnz[name]}
        return out or None
    def gram_roof(name, peak):
        ms = kern[name]["ms_per_call"]
        flops_ref = 2.0 * d * (p + 1) ** 2 * T / 2.0        # the reference's per-pixel Gram, symmetric count, first run (T' = T)
        flops_alg = flops_ref / 2.58                        # what the block-pair table needs: every covariance once, pruned sub-tiles (9.57 TFLOP at the headline size)
        return {"bound": "mfma", "achieved": flops_alg / ms / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": flops_alg / ms / 1e9 / peak, "traffic": None, "kernel": name, "ms_per_launch": ms,
                "algorithmic_flops_per_launch": flops_alg,
                "note": "algorithmic = the block-pair covariance table (each needed covariance once: 2*d*(p+1)^2*T/2 / 2.58 fp32-equivalent flops; the reference's "
                        "per-pixel Gram would be %.3g).  With the incremental table this kernel runs once per patch, not per iteration -- DESIGN.md" % flops_ref}
    _pmc_cache = {}
    def pmc_traffic_live():
        """Fabric-side bytes per launch of the roofline kernels, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass;
        --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a CHILD run of this very command (2 steps, no extras), i.e. the same synthetic video, the
        same kernels, this GPU.  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes).  Returns
        {kernel substring: {"FETCH_SIZE": bytes, "WRITE_SIZE": bytes, "launches": n}} (median over the child's launches) or None if rocprofv3 is missing or fails."""
        if "v" in _pmc_cache:
            return _pmc_cache["v"]
        _pmc_cache["v"] = None
        if a.no_extras or a.config != "c3" or world != 1 or a.bg_ssub != 1 or a.deconv or os.environ.get("CNMFE_BENCH_PMC", "1") == "0" or not shutil.which("rocprofv3"):
            return None
        import csv, glob
        keys = ("k_ring_solve", "k_vp_proj_", "k_win_proj", "k_residual_duo")
        res = {k: {} for k in keys}
        for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            tmp = tempfile.mkdtemp(prefix="cnmfe_pmc_", dir="/tmp")
            try:
                subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "x", "--",
                                sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--alg", a.alg, "--no-cpu-baseline", "--no-extras"],
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
                vals = {k: [] for k in keys}
                for fn in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                    for r_ in csv.DictReader(open(fn)):
                        if r_["Counter_Name"] != counter:
                            continue
                        for k in keys:
                            if k in r_["Kernel_Name"]:
                                vals[k].append(float(r_["Counter_Value"]))
                for k in keys:
                    if vals[k]:
                        v = sorted(vals[k])
                        res[k][counter] = mult * 1024.0 * v[len(v) // 2]          # the counters are in KiB
                        res[k]["launches"] = len(v)
            except Exception:
                return None
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        _pmc_cache["v"] = {k: v for k, v in res.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v} or None
        return _pmc_cache["v"]
    def live_traffic(obj, key):
        """fills obj['traffic'] / ['traffic_source'] from this run's PMC passes; False when they are not available"""
        live = pmc_traffic_live()
        if obj is None or not live or key not in live:
            return False
        obj["traffic"] = live[key]["FETCH_SIZE"] + live[key]["WRITE_SIZE"]
        obj["traffic_source"] = ("this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes, --kernel-trace only) over a child run of this command; median of %d launches; "
                                 "2 x FETCH_SIZE (%.4g B) + WRITE_SIZE (%.4g B)" % (live[key]["launches"], live[key]["FETCH_SIZE"], live[key]["WRITE_SIZE"]))
        return True

    def pmc_traffic(kernel_substr, exclude=None):
        """HBM/fabric bytes per launch from the committed rocprofv3 PMC passes of THIS command (profiles/<round>/
        *_pmc_FETCH_SIZE_*.csv, *_pmc_WRITE_SIZE_*.csv; separate passes, scripts/profile_round.sh).  FETCH_SIZE is
        doubled: gfx950 reports 16-B/lane coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM section; calibrated
        on bg_build_bf = one 10.5 GB sweep).  None when no summary for this config is present."""
        import csv, glob
        if a.config != "c3" or world != 1:
            return None
        tot = 0.0
        for counter, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_c3_pmc_%s_v*.csv" % counter)))
            if not files:
                return None
            rows = [r_ for r_ in csv.DictReader(l for l in open(files[-1]) if not l.startswith("#")) if kernel_substr in r_["kernel"] and not (exclude and exclude in r_["kernel"])]
            if not rows:
                return None
            tot += mult * 1024.0 * sum(float(r_["value_per_launch_KiB"]) for r_ in rows) / len(rows)
        return tot
    def with_pmc(pr):
        # fabric traffic of the two video passes: from this run's PMC passes, else from the committed ones (said so)
        if pr:
            for name, sub in (("temporal_proj_B", "k_vp_proj_"), ("bg_win_proj", "k_win_proj")):
                if name in pr and not live_traffic(pr[name], sub):
                    pr[name]["traffic"] = pmc_traffic(sub)
                    pr[name]["traffic_source"] = None if pr[name]["traffic"] is None else "NOT this run: 2 x FETCH_SIZE + WRITE_SIZE of the newest profiles/r*/bench_c3_pmc_{FETCH,WRITE}_SIZE_v*.csv"
        return pr
    def solve_roof():
        # per active pixel: gather the (p+1)x(p+1) Gram + RHS from the table, ridge, Cholesky, two triangular solves -- all fp64
        if "bg_ring_solve" not in kern:
            return None
        n = (p or 96) + 1
        infos = [i_ for i_ in (last.get("bg") or {}).values()]
        # per launch = per fitted patch; the engine reports -1 when it did not wait for the count (asynchronous fit): then every patch pixel
        # (at the headline configuration 245152 of the 262144 are active: the ratio below is then 7 % high)
        acts = [int(i_.get("n_active", -1)) for i_ in infos]
        n_act = (sum(acts) / float(len(acts))) if acts and min(acts) >= 0 else d / float(max(1, len(video.owned)))
        ms = kern["bg_ring_solve"]["ms_per_call"]
        # a Cholesky is n^3 / 3 flops, the two substitutions 2 n^2 (round 5 priced twice that: multiply-adds counted as two)
        fl = n_act * (n ** 3 / 3.0 + 2.0 * n * n)
        nt_ = (n - 1 + 15) // 16
        by = n_act * 8.0 * ((nt_ * (nt_ + 1) // 2) * 256 + nt_ * 16)
        # both bounds: the packed systems take longer to stream at the HBM peak than the flops take at the fp64 peak -- the bytes are the roofline of this kernel
        t_fl, t_by = fl / (F64_MFMA_PEAK_TF * 1e12), by / (HBM_PEAK_GBS * 1e9)
        bytes_bound = t_by >= t_fl
        return {"bound": "hbm" if bytes_bound else "mfma",
                "achieved": by / ms / 1e6 if bytes_bound else fl / ms / 1e9, "peak": HBM_PEAK_GBS if bytes_bound else F64_MFMA_PEAK_TF, "unit": "GB/s" if bytes_bound else "TFLOP/s",
                "frac": (by / ms / 1e6 / HBM_PEAK_GBS) if bytes_bound else (fl / ms / 1e9 / F64_MFMA_PEAK_TF), "traffic": None,
                "kernel": "bg_ring_solve", "ms_per_launch": ms, "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                "frac_of_hbm_peak": by / ms / 1e6 / HBM_PEAK_GBS, "frac_of_fp64_peak": fl / ms / 1e9 / F64_MFMA_PEAK_TF,
                "note": "fp64 Cholesky + substitutions of %d independent %dx%d systems per launch (one per active pixel: n^3/3 + 2n^2 flops each, SURVEY.md 8(d); the footprints' "
                        "rank-2 corrections -- 4 n^2 flops per neuron around a pixel -- are NOT counted), loaded from a per-pixel packed copy of the video's table (43 KB per pixel at "
                        "p = 96, coalesced) and corrected in registers.  Priced against both bounds (frac_of_hbm_peak, frac_of_fp64_peak); the bytes are the binding one.  One wave per "
                        "pixel, the matrix in MFMA accumulator tiles (v_mfma_f64_16x16x4); what holds the kernel at this fraction is neither pipe but the LATENCY of its dependent "
                        "chains at two waves per SIMD (four dependent gathers of the set-up, 6 x 240 DPP FMAs of the diagonal steps, the substitutions' LDS round trips) -- "
                        "DESIGN.md section 3, scripts/probes/solve_r5 and solve_r6 (fp32 + refinement: measured slower)" % (int(n_act), n, n)}
    if dom.startswith("bg_gram"):                                # (only with --warmup 0: the table of the video is built once per recording)
        if dom == "bg_gram_i8":                                  # 13 int8 digit-pair products per fp32-equivalent product (gram_i8.hpp), priced against the int8 matrix peak
            roof = gram_roof(dom, I8_MFMA_PEAK_TOPS)
            roof["achieved"] *= 13.0; roof["frac"] *= 13.0; roof["algorithmic_flops_per_launch"] *= 13.0; roof["unit"] = "TOP/s"
            roof["note"] = ("every needed covariance once (block-sparse SYRK, 9.57 T fp32-equivalent multiply-adds x 2 at the headline size) as 13 int8 digit-pair products "
                            "with exact int32 accumulation; the kernel is bound by its operand traffic (the split-bf16 mode of rounds 2-4 moved the same bytes in 50 ms), "
                            "not by the matrix pipe -- DESIGN.md")
        else:
            roof = gram_roof(dom, F64_MFMA_PEAK_TF)
    elif dom == "residual_r1":
        roof = r1_roof()
    elif dom == "ssub_up_fused":                                 # bg_ssub > 1: read Y' + W*(..) at low resolution (two arrays), write Ysig
        ms = kern[dom]["ms_per_call"]
        by = 4.0 * d_b * T + 4.0 * d * T + 2 * 4.0 * (d_b / float(a.bg_ssub ** 2)) * T
        roof = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                "kernel": dom, "ms_per_launch": ms, "algorithmic_bytes_per_launch": by}
    elif dom == "bg_ring_solve":
        roof = solve_roof()
    elif dom in ("spatial_proj_rows", "temporal_proj_rows") and (proj_roofs() or {}).get(dom):
        # bg_ssub > 1, sweep-free: the largest kernel is the fp64 projection of the video rows under the search masks (one read of those rows)
        pr = proj_roofs()[dom]
        roof = {"bound": "hbm", "achieved": pr["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pr["frac"], "traffic": None, "kernel": dom,
                "ms_per_launch": pr["ms_per_update"], "algorithmic_bytes_per_launch": pr["algorithmic_bytes_per_update"],
                "note": "bg_ssub = %d: no kernel dominates the sweep-free iteration (the low-resolution fit's solve and window projection, two reads of the low-resolution "
                        "video and these rows of the full-resolution one are 0.7-1.1 ms each -- roofline_projections); this is the pass over the pixel rows under the search "
                        "masks, strided 16-byte reads per entry, fp64 sums" % a.bg_ssub}
    else:
        roof = {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "kernel": dom,
                "ms_per_launch": kern[dom]["ms_per_call"]}
    if roof.get("kernel", "").startswith("bg_gram"):
        roof["traffic"] = pmc_traffic("k_gram")
    if roof.get("kernel") == "bg_ring_solve" and not live_traffic(roof, "k_ring_solve"):
        roof["traffic"] = pmc_traffic("k_ring_solve")
        import glob as _glob
        pf = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_c3_pmc_FETCH_SIZE_v*.csv")))
        roof["traffic_source"] = None if roof["traffic"] is None else ("NOT this run: 2 x FETCH_SIZE + WRITE_SIZE of %s (two rocprofv3 --pmc passes of this command, "
                                                                        "scripts/profile_round.sh)" % os.path.relpath(pf[-1], ROOT).replace("FETCH_SIZE", "{FETCH,WRITE}_SIZE"))
    r1r = r1_roof()
    if r1r is not None:
        if not live_traffic(r1r, "k_residual_duo"):
            r1r["traffic"] = pmc_traffic("k_residual", exclude="k_residual_delta")
            r1r["traffic_source"] = None if r1r["traffic"] is None else "NOT this run: the newest profiles/r*/bench_c3_pmc_{FETCH,WRITE}_SIZE_v*.csv in the tree"
        if roof.get("kernel") == "residual_r1":
            roof["traffic"] = r1r["traffic"]; roof["traffic_source"] = r1r["traffic_source"]
    dlr = None
    if "residual_delta" in kern and a.bg_ssub == 1:             # the iteration's second residual: resident Ysig + footprint-term difference, one streaming pass
        ms = kern["residual_delta"]["ms_per_call"]
        by = 2 * 4.0 * d * T + 4.0 * Kp * T
        dlr = {"bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS,
               "traffic": pmc_traffic("k_residual_delta"), "kernel": "residual_delta", "ms_per_launch": ms, "algorithmic_bytes_per_launch": by}
    out = {
        "metric": "cnmfe_iters_per_sec" if not shard_of else "cnmfe_rank_share_iters_per_sec", "value": value, "unit": "iter/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak" if a.weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("%s: %dx%dx%d fp32 video, K=%d, ring_radius=%d, %dx%d patches of distribute_data.m (%d resident blocks per GPU), "
                                "spatial=%s, deconv_flag=%s, bg_ssub=%d" % (a.config, d1, d2, T, K, r, video.nr_patch, video.nc_patch, len(video.owned), a.alg, "true" if a.deconv else "false", a.bg_ssub))
                               if not a.weak else
                               ("%s (weak): %dx%dx%d fp32 video per GPU, K=%d per GPU, ring_radius=%d, 1 patch per GPU (%d patches), "
                                "spatial=%s, deconv_flag=%s, bg_ssub=%d" % (a.config, d1, d2p, T, Kp, r, n_patches, a.alg, "true" if a.deconv else "false", a.bg_ssub)),
                   "iteration": "update_background_parallel + update_spatial_parallel + update_temporal_parallel",
                   "formulation": ("sweep-free residual through the resampling maps (option ssub_virtual = 1, the default): Ysig*C' = P_F - up(W_L*P_L) with P_F the video rows "
                                   "under the masks and P_L the table of the low-resolution video, A'*Ysig = A'*Yc - B_L'*(down Yc), B_L = W_L'*up'*A; same values as the swept "
                                   "residual (tests/test_gpu_ssub_virtual.py)" if a.bg_ssub != 1 and os.environ.get("CNMFE_OPTS", "").find("ssub_virtual=0") < 0 and os.environ.get("CNMFE_OPTS", "").find("r1_virtual=0") < 0 else
                                  ("sweep-free residual (option r1_virtual = 1, the default): the spatial update reads Ysig*C' = P - W*P out of the fit's table "
                                   "P = Yc*Cc', the temporal update projects the centred video through B = A - W'*A; same values as the swept residual "
                                   "(tests/test_gpu_virtual.py); the ring sweep itself is timed separately for `roofline_r1`")
                                  if os.environ.get("CNMFE_OPTS", "").find("r1_virtual=0") < 0 and a.bg_ssub == 1 else "swept residual (r1_virtual = 0 / ssub_virtual = 0): one ring sweep per iteration"),
                   "lanes_per_rank": lanes,
                   "parallelism": ("patches round-robin over %d rank(s)" % world) if not shard_of else
                                  ("rank 0 of %d: this rank's %d of the %d patches on ONE GPU, no collectives (a per-rank load figure, not a scaling point); "
                                   "the video is uploaded as fp16 and widened on the device" % (shard_of, len(video.owned), len(video.order))),
                   **({"note": "strong scaling of the 4 x 4-patch decomposition (BASELINE configs[3]); its own N = 1 point is the `c4_n1` object of the N = 1 line "
                               "(16 patches on one GPU) -- the N = 1 `value` is configs[2], the same video as ONE patch, which has no halo re-reads and 16x larger launches"}
                      if (world > 1 and a.config == "c4" and not a.weak) else {})},
        "roofline": roof,
        "roofline_r1": r1r,
        "roofline_solve": solve_roof(),
        "roofline_r1_delta": dlr,
        "roofline_projections": with_pmc(proj_roofs()),
        **comm,
        "first_iteration": {"ms": warm_steps_ms[0] if warm_steps_ms else None, "warmup_steps_ms": [round(x, 3) for x in warm_steps_ms],
                            "one_off_kernels_ms": {k: round(v["total_ms"], 3) for k, v in warm_tab.items() if k in ("bg_gram_i8", "bg_gram_f64", "bg_build_dig", "bg_dig_scale", "bg_build_bf", "bg_rowsum", "bg_sys_pack") and v["calls"]},
                            "kernels_ms": None if first_tab is None else {k: round(v["total_ms"], 3) for k, v in sorted(first_tab.items(), key=lambda kv: -kv[1]["total_ms"])[:14] if v["calls"]},
                            "kernel_sum_ms": None if first_tab is None else round(sum(v["total_ms"] for v in first_tab.values()), 3),
                            "note": "the FIRST step after the upload (timed on its own): its background fit also builds the block-pair covariance table of the video on "
                                    "the fp64 matrix pipe (kept until the video or the frame stride changes); `--warmup 0` puts it inside the timed region"},
        # a recording gets TWO background updates (demos/demo_large_data_1p.m:142,199): the mean of the first two steps from a fresh upload
        "ms_per_step_incl_cold": (sum(warm_steps_ms[:2]) / 2.0) if len(warm_steps_ms) >= 2 else None,
        "kernel_timing": {"timed_region": sorted(k for k, v in kern.items() if v["from"] == "timed region"),
                          "note": "HIP events on the engine's stream.  Inside the timed region only the kernels a roofline is quoted for are bracketed "
                                  "(cnmfe_profile_enable(ctx, 2)); every other row of kernels_ms_per_step is the average of %d extra steps run after it with "
                                  "events around every launch" % XSTEPS},
        "kernel_sum_ms_per_step": round(sum(v["ms_per_step"] for v in kern.values()), 3),
        "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        "kernel_calls_per_step": {k: round(v["calls_per_step"], 2) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"]) if v["ms_per_step"] > 0.5},
    }
    # the driver's record keeps `roofline`, `config` and `cpu_baseline` whole and only the NAMES of the other keys: the iteration's other large kernels, the share of
    # the step the kernels fill and the cold figure ride along inside `roofline` (verdict r4 #8)
    if out["roofline"] is not None:
        top = sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])[:5]
        out["roofline"]["top5_kernels_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in top}
        out["roofline"]["kernel_sum_over_step"] = round(out["kernel_sum_ms_per_step"] / out["ms_per_step"], 4)
        out["roofline"]["ms_per_step_incl_cold"] = None if out["ms_per_step_incl_cold"] is None else round(out["ms_per_step_incl_cold"], 2)
        out["roofline"]["first_iteration_ms"] = None if not warm_steps_ms else round(warm_steps_ms[0], 2)
        pr_ = out.get("roofline_projections") or {}
        out["roofline"]["video_passes"] = {k: {"ms": round(v["ms_per_update"], 3), "frac_of_hbm": round(v["frac"], 3),
                                               "traffic_over_algorithmic": None if not v.get("traffic") else round(v["traffic"] / v["algorithmic_bytes_per_update"], 3)}
                                           for k, v in pr_.items() if k in ("temporal_proj_B", "bg_win_proj")} or None
        if out.get("roofline_r1"):
            out["roofline"]["r1_sweep_not_in_iteration"] = {"ms": round(out["roofline_r1"]["ms_per_launch"], 3), "frac_of_hbm": round(out["roofline_r1"]["frac"], 3),
                                                            "frac_of_fp32_vector_peak": round(out["roofline_r1"]["frac_of_fp32_vector_peak"], 3),
                                                            "traffic_over_algorithmic": None if not out["roofline_r1"].get("traffic") else
                                                            round(out["roofline_r1"]["traffic"] / out["roofline_r1"]["algorithmic_bytes_per_launch"], 3)}
    eng.close()
    del s, video
    torch.cuda.empty_cache()
    out["c4_n1"] = None
    if world == 1 and a.config == "c3" and not a.no_extras and not a.weak and a.bg_ssub == 1 and not a.deconv and a.alg == "hals":
        # the N = 1 point of the strong-scaling curve (`--gpus N` runs configs[3] = this video in 4 x 4 patches): all 16 patches on this one GPU
        try:
            r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "c4", "--steps", "6", "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, check=True)
            c4 = json.loads(r_.stdout.decode().strip().splitlines()[-1])
            out["c4_n1"] = {k: c4[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "n_gpus", "kernel_sum_ms_per_step")}
            out["c4_n1"]["workload"] = c4["config"]["workload"]
        except Exception as e:                                  # the headline line must not die with the extra
            out["c4_n1"] = {"error": repr(e)[:200]}
    if not a.no_cpu_baseline and a.cpu_baseline != "none" and world == 1:
        out["cpu_baseline"] = cpu_baseline()
    else:
        out["cpu_baseline"] = None
    emit(out)


if __name__ == "__main__":
    main()
