"""CPU-side checks (run with -m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, fails loudly without a GPU, and the Python host logic agrees with the oracle."""
import ctypes
import os
import re

import numpy as np
import scipy.sparse as sp
import pytest

import cnmfe_oracle as orc
from cnmf_e_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_lib():
    return os.path.exists(os.path.join(ROOT, "cnmf_e_amd", "libcnmfe_hip.so"))


@pytest.fixture(scope="module")
def built():
    if not _have_lib():
        import cnmf_e_amd.build as b
        b.build(verbose=False)
    return True


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "cnmfe.h")).read()
    names = set(re.findall(r"\b(cnmfe_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    lib = ctypes.CDLL(os.path.join(ROOT, "cnmf_e_amd", "libcnmfe_hip.so"))
    for n in sorted(names):
        assert hasattr(lib, n), "library does not export %s" % n
    from cnmf_e_amd import _lib
    assert set(_lib.PROTOTYPES) == names, set(_lib.PROTOTYPES) ^ names


def test_no_silent_cpu_fallback(built):
    """Without a GPU the engine must refuse to start (no CPU fallback anywhere in the product)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cnmf_e_amd.engine import Engine
    from cnmf_e_amd._lib import CnmfeError
    with pytest.raises(CnmfeError):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cnmf_e_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "cnmfe_oracle" not in src and "import oracle" not in src, fn


def test_geometry_matches_oracle():
    from cnmf_e_amd.sources2d import distribute_geometry
    for (d1, d2, pdims, w) in [(512, 512, [128, 128], 15), (64, 48, [64, 48], 5), (100, 90, [33, 31], 6),
                               (256, 256, [64, 128], 15), (1024, 1024, [128, 128], 15), (75, 130, [30], 4)]:
        (nr, nc), pp, bp = distribute_geometry(d1, d2, pdims, w)
        opp, obp = orc.distribute_geometry(d1, d2, pdims, w)
        assert (nr, nc) == opp.shape
        for m in range(nr):
            for n in range(nc):
                assert list(pp[(m, n)]) == list(opp[m, n]) and list(bp[(m, n)]) == list(obp[m, n])


def test_search_location_matches_oracle():
    from cnmf_e_amd.sources2d import determine_search_location
    f = synth.make_factors(60, 50, 20, 12, 3, gSig=2.0, gSiz=9, min_sep=4)
    A = f.A_init.tolil()
    A[:, 3] = 0                       # an empty component -> all-false column (determine_search_location.m:52-55,102-104)
    A = A.tocsc()
    IND = determine_search_location(A, 60, 50)
    ref = orc.determine_search_location(A, 60, 50)
    assert IND.shape == ref.shape
    assert np.array_equal(IND.toarray(), ref)
    assert IND[:, 3].nnz == 0
    # elongated component + different expansion parameters
    B = f.A_true.tolil()
    B[10 * 60 + 5:10 * 60 + 25, 0] = 1.0
    IND2 = determine_search_location(B.tocsc(), 60, 50, 2, 6, 2.5)
    ref2 = orc.determine_search_location(B.tocsc(), 60, 50, 2, 6, 2.5)
    assert np.array_equal(IND2.toarray(), ref2)


def test_ring_offsets_c_abi_order():
    """the engine's ring order (dc slow, dr fast) is MATLAB's find() order used by the oracle"""
    for r in (4, 5, 9, 15, 18):
        rs, cs = orc.get_nhood(r)
        k = 0
        for c in range(-r, r + 1):
            for rr in range(-r, r + 1):
                d2 = c * c + rr * rr
                if r * r <= d2 < (r + 1) * (r + 1):
                    assert rs[k] == rr and cs[k] == c
                    k += 1
        assert k == rs.size


def test_compute_rss_host_logic_with_fake_engine():
    """Sources2D.compute_RSS (neuron / pixel selection per patch, b0 stitching, summation) against the oracle, on the NumPy fake engine"""
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 30, 28, 60, 4, 5
    f = synth.make_factors(d1, d2, T, K, 7, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    eng = FakeEngine()
    video = PatchedVideo(d1, d2, T, [15, 14], r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, maxIter=2), f.A_init, f.C_init, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [15, 14], r, f.A_init.astype(np.float32), f.C_init, f.sn, maxIter=2)
    for step in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"):
        getattr(s, step)(); getattr(o, step)()
        got, _ = s.compute_RSS(); ref, _ = o.compute_RSS()
        assert abs(got - ref) <= 1e-6 * ref, (step, got, ref)


def test_init_residual_host_logic_with_fake_engine():
    """Sources2D.init_residual (block's neurons, patch rows, A*C on the exported copy) against the oracle's restatement of
    initComponents_residual_parallel.m:106-121,186-217, on the NumPy fake engine"""
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 30, 28, 60, 4, 5
    f = synth.make_factors(d1, d2, T, K, 7, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32)
    eng = FakeEngine()
    video = PatchedVideo(d1, d2, T, [15, 14], r, eng)
    video.upload_from_full(Y)
    s = Sources2D(video, Options(ring_radius=r, maxIter=2), f.A_init, f.C_init, f.sn)
    o = orc.OracleSources2D(Y.T.reshape(d1, d2, T, order="F"), d1, d2, T, [15, 14], r, f.A_init.astype(np.float32), f.C_init, f.sn, maxIter=2)
    for step in ("update_background_parallel", "update_spatial_parallel", "update_temporal_parallel"):
        getattr(s, step)(); getattr(o, step)()
    for idx in video.owned:
        got = np.asarray(s.init_residual(idx), dtype=np.float64).T
        ref = o.init_residual(idx)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), idx


def test_data_plane_readers_match_upload_from_full(tmp_path):
    """upload_from_images / upload_from_tiff / upload_from_raw hand every block the same frames as upload_from_full (the reference's
    distribute_data + get_patch_data: blocks with halo, MATLAB pixel order), in the file's element type"""
    from PIL import Image
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo
    d1, d2, T, r = 18, 14, 23, 3
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 4000, size=(T, d1, d2)).astype(np.uint16)             # frames as images (rows x columns)
    Y_td = vol.transpose(0, 2, 1).reshape(T, d1 * d2)                            # MATLAB d x T, transposed: pixels column-major

    def blocks(loader):
        eng = FakeEngine()
        v = PatchedVideo(d1, d2, T, [9, 7], r, eng)
        loader(v)
        return {idx: eng.p[v.pid[idx]]["Y"].copy() for idx in v.owned}

    ref = blocks(lambda v: v.upload_from_full(Y_td, chunk=5))
    tif = str(tmp_path / "rec.tif")
    pages = [Image.fromarray(vol[t]) for t in range(T)]
    pages[0].save(tif, save_all=True, append_images=pages[1:])
    rawF, rawC = str(tmp_path / "rec_f.bin"), str(tmp_path / "rec_c.bin")
    Y_td.tofile(rawF); vol.tofile(rawC)
    for name, got in (("images", blocks(lambda v: v.upload_from_images(vol, chunk=7))),
                      ("iterable", blocks(lambda v: v.upload_from_images(iter(vol), chunk=4))),
                      ("tiff", blocks(lambda v: v.upload_from_tiff(tif, chunk=6))),
                      ("raw F", blocks(lambda v: v.upload_from_raw(rawF, np.uint16, chunk=8))),
                      ("raw C", blocks(lambda v: v.upload_from_raw(rawC, np.uint16, order="C", chunk=8)))):
        assert set(got) == set(ref), name
        for idx in ref:
            assert np.array_equal(np.asarray(got[idx], dtype=np.float64), np.asarray(ref[idx], dtype=np.float64)), (name, idx)
    with pytest.raises(ValueError):
        blocks(lambda v: v.upload_from_images(vol[:-1]))


def test_mex_gateway_compiles_against_stub():
    """cnmf_e_amd/csrc/matlab/cnmfe_mex.cpp is type-checked against include/cnmfe.h and stub MEX headers (tests/mex_stub: declarations only);
    the gateway must not hold C++ objects across mexErrMsgIdAndTxt (which long-jumps): no STL anywhere in it."""
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "cnmf_e_amd", "csrc", "matlab", "cnmfe_mex.cpp")
    text = open(src).read()
    assert "std::" not in text and "#include <vector>" not in text and "#include <string>" not in text
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    out = subprocess.run([cxx, "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "tests", "mex_stub"),
                          "-I", os.path.join(root, "include"), src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


def test_matlab_host_speaks_the_gateways_commands():
    """every cnmfe_mex('command', ...) in the MATLAB host files is a command the gateway dispatches, and the three drop-in methods exist"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mdir = os.path.join(root, "cnmf_e_amd", "csrc", "matlab")
    gateway = set(re.findall(r'strcmp\(cmd, "(\w+)"\)', open(os.path.join(mdir, "cnmfe_mex.cpp")).read()))
    used = set()
    mfiles = [os.path.join(dp, f) for dp, _, fs in os.walk(mdir) for f in fs if f.endswith(".m")]
    for fn in mfiles:
        used |= set(re.findall(r"cnmfe_mex\('(\w+)'", open(fn).read()))
    assert used and used <= gateway, used - gateway
    for name in ("update_background_parallel.m", "update_spatial_parallel.m", "update_temporal_parallel.m"):
        assert os.path.exists(os.path.join(mdir, "@Sources2D", name))
    # the classdef methods of Sources2D.m that touch the video have a body on the engine too, and the host files use the commands made for them
    for name, cmds in (("cnmfe_estimate_noise.m", {"estimate_noise"}), ("cnmfe_compute_RSS.m", {"compute_rss", "compute_rss_ssub", "background_ssub"}),
                       ("cnmfe_reconstruct_background.m", {"reconstruct_background", "reconstruct_background_ssub", "background_ssub"})):
        txt = open(os.path.join(mdir, name)).read()
        assert cmds <= set(re.findall(r"cnmfe_mex\('(\w+)'", txt)), name
    # number of arguments of every call = what the gateway checks (nin counts the command string too)
    src = open(os.path.join(mdir, "cnmfe_mex.cpp")).read()
    need = {m.group(1): int(m.group(2)) for m in re.finditer(r'strcmp\(cmd, "(\w+)"\)\) \{[^\n]*\n\s*if \(nin != (\d+)\)', src)}
    for fn in mfiles:
        txt = open(fn).read()
        for m in re.finditer(r"cnmfe_mex\('(\w+)'", txt):
            cmd, i, depth, nargs = m.group(1), m.end(), 1, 1
            while depth:
                ch = txt[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                nargs += ch == "," and depth == 1
                i += 1
            if cmd in need and nargs > 1:                                # (a bare cnmfe_mex('name') is a mention in a comment)
                assert nargs == need[cmd], (os.path.basename(fn), cmd, nargs, need[cmd])


def test_bench_line_contract_of_the_committed_profile():
    """the bench line the round's profile was taken with carries every field the driver and the judge read (the same code prints it on the GPU box)"""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_c3_v*.json")))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "512x512x10000" in d["config"]["workload"] and "K=500" in d["config"]["workload"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 78.6 if "f64" in r.get("kernel", "") or r.get("kernel") == "bg_ring_solve" else r["peak"])       # MI355X_MICROARCH.md: 8 TB/s; fp64 matrix pipe 78.6 TFLOP/s
    r1 = d.get("roofline_r1")                                   # the north star's kernel: since round 4 timed on its own (the iteration runs no sweep)
    if r1 is not None:
        assert r1["bound"] == "hbm" and r1["peak"] == 8000.0 and abs(r1["frac"] - r1["achieved"] / r1["peak"]) <= 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_noise_image_bookkeeping_and_nearest_rows_match_the_oracle():
    """host pieces of round 2: the storage-block bookkeeping of Sources2D.m:361-376 applied to per-pixel estimates == the oracle's literal block
    loop, on several geometries; and the rows imresize(., 1/s, 'nearest') keeps == the oracle's box-kernel weights"""
    from cnmf_e_amd.sources2d import storage_block_index, estimate_noise_image, nearest_rows
    rng = np.random.default_rng(3)
    for (d1, d2, pd, r) in ((44, 40, [22, 20], 5), (60, 51, [20, 17], 4), (37, 64, [37, 16], 6), (48, 48, [48, 48], 7)):
        T = 80
        Y = rng.standard_normal((d1, d2, T)).astype(np.float32)
        K = 2
        A = sp.csc_matrix((d1 * d2, K), dtype=np.float32); Cm = np.zeros((K, T), np.float32)
        o = orc.OracleSources2D(Y, d1, d2, T, pd, r, A, Cm, np.ones(d1 * d2))
        ref = o.estimate_noise()
        from oasis_oracle import GetSn
        pix = np.array([[GetSn(Y[i, j].astype(np.float64)) for j in range(d2)] for i in range(d1)])
        nr, nc = o.patch_pos.shape
        pr = np.array([int(o.patch_pos[m, 0][0]) for m in range(nr)] + [d1]); pc = np.array([int(o.patch_pos[0, j][2]) for j in range(nc)] + [d2])
        got = estimate_noise_image(pix, storage_block_index(d1, pr, r), storage_block_index(d2, pc, r))
        assert np.array_equal(got, ref), (d1, d2, pd)
    for n in (7, 16, 33, 46, 50, 158):
        for s_ in (2, 3, 4):
            M = orc.imresize_weights(n, -(-n // s_), 1.0 / s_, "nearest")
            assert (M.max(axis=1) == 1).all() and np.array_equal(M.argmax(axis=1), nearest_rows(n, s_)), (n, s_)


@pytest.mark.parametrize("pdims,update_sn", [([20, 22], False), ([20, 22], True), (None, False)])
def test_asynchronous_engine_orderings_give_the_same_iteration(pdims, update_sn):
    """with an engine that defers its results (the real one) sources2d reorders host work: the temporal update's residual is requested under the
    spatial sweeps, the next patch's slices are cut before this patch's fetch, a whole-FOV patch gets its connectivity constraint with the fetch.
    Two iterations with the deferring test double = two iterations with the blocking one, exactly; and the call order is the intended one."""
    from fake_engine import FakeEngine, LazyFakeEngine
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 40, 44, 120, 6, 4
    f = synth.make_factors(d1, d2, T, K, 31, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32).astype(np.float64)

    def run(engine):
        v = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, engine)
        v.upload_from_full(Y)
        s = Sources2D(v, Options(ring_radius=r, spatial_algorithm="hals", maxIter=3), f.A_init, f.C_init, f.sn)
        for _ in range(2):
            s.update_background_parallel(); s.update_spatial_parallel(update_sn=update_sn); s.update_temporal_parallel()
        return s, v

    a, _ = run(FakeEngine())
    eng = LazyFakeEngine()
    b, v = run(eng)
    assert (a.A != b.A).nnz == 0 and np.array_equal(np.asarray(a.C), np.asarray(b.C)) and np.array_equal(np.asarray(a.C_raw), np.asarray(b.C_raw))
    assert np.array_equal(a.b0_new, b.b0_new) and np.array_equal(a.P["sn"], b.P["sn"])
    # per patch of a spatial update: residual, update_spatial, the temporal update's residual, fetch -- and the temporal update asks for no further sweep
    n = len(v.owned)
    calls = eng.calls[-4 * n:]
    assert [c[0] for c in calls] == ["residual", "update_spatial", "residual", "fetch"] * n, calls
    assert [c[1] for c in calls] == [v.pid[idx] for idx in v.owned for _ in range(4)]


@pytest.mark.parametrize("pdims,lag", [([20, 22], 1), ([14, 15], 1), (None, 1), ([14, 15], 16)])
def test_late_collection_of_spatial_results_gives_the_same_iteration(pdims, lag):
    """with the real engine's queued download (fetch.start / fetch(compact=True)) sources2d queues patch m + 1 before it collects patch m and leaves the
    one-patch A_raw as a recipe: two iterations equal the blocking double's exactly, A_raw included, and every fetch but the last follows the NEXT
    patch's launches."""
    from fake_engine import FakeEngine, LateFakeEngine
    from cnmf_e_amd import synth
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options
    d1, d2, T, K, r = 40, 44, 120, 6, 4
    f = synth.make_factors(d1, d2, T, K, 31, gSig=1.5, gSiz=7, min_sep=5)
    Y = synth.make_video(f, np.float32).astype(np.float64)

    def run(engine):
        v = PatchedVideo(d1, d2, T, pdims or [d1, d2], r, engine)
        v.upload_from_full(Y)
        s = Sources2D(v, Options(ring_radius=r, spatial_algorithm="hals", maxIter=3), f.A_init, f.C_init, f.sn)
        for _ in range(2):
            s.update_background_parallel()
            mark[0] = len(getattr(engine, "calls", []))
            s.update_spatial_parallel(); s.update_temporal_parallel()
        return s, v

    mark = [0]
    a, _ = run(FakeEngine())
    eng = LateFakeEngine()
    Sources2D.spatial_lag, lag0 = lag, Sources2D.spatial_lag      # (a lag of one patch makes the order below simple to state; 16 = the default: everything collected at the end)
    try:
        b, v = run(eng)
    finally:
        Sources2D.spatial_lag = lag0
    assert (a.A != b.A).nnz == 0 and np.array_equal(np.asarray(a.C), np.asarray(b.C)) and np.array_equal(np.asarray(a.C_raw), np.asarray(b.C_raw))
    assert callable(b.__dict__["_A_raw"]) == (pdims is None)               # one patch: still the recipe ...
    assert (sp.csc_matrix(a.A_raw) != sp.csc_matrix(b.A_raw)).nnz == 0     # ... that builds the same matrix on first read
    assert not callable(b.__dict__["_A_raw"])
    assert np.array_equal(a.b0_new, b.b0_new)
    last = eng.calls[mark[0]:]
    if pdims is None:
        assert [c[0] for c in last[:4]] == ["residual", "update_spatial", "residual", "fetch"]      # the whole-FOV fetch is not deferred
        return
    # the temporal update of several patches: every patch's job first, ONE sweep, then the jobs into the stitch in order
    tcalls = [c for c in last if c[0] in ("job", "sweep", "add_job")]
    nj = sum(1 for c in tcalls if c[0] == "job")
    assert nj > 2 and [c[0] for c in tcalls] == ["job"] * nj + ["sweep"] + ["add_job"] * nj and [c[1] for c in tcalls[nj + 1:]] == list(range(nj)), tcalls
    calls = [c for c in last if c[0] in ("update_spatial", "start", "fetch")]
    pids = [p_ for c, p_ in calls if c == "update_spatial"]               # (patches without a neuron are skipped)
    assert len(pids) > 2
    if lag > 1:
        assert [c for c in calls if c[0] == "fetch"] == [("fetch", p_) for p_ in pids] and calls[-len(pids):] == [("fetch", p_) for p_ in pids]
        return
    expect = []
    for i, p_ in enumerate(pids):
        expect += [("update_spatial", p_), ("start", p_)]
        if i > 0:
            expect.append(("fetch", pids[i - 1]))
    expect.append(("fetch", pids[-1]))
    assert calls == expect, calls


def test_sparse_row_selection_matches_scipy_on_random_matrices():
    """rows_of / Sources2D._slice (the O(nnz) selection of a block's, patch's or halo's rows out of the d x K footprints, with the bounding-box
    prefilter) against plain scipy indexing: `ind = find(sum(A(mask, :), 1) > 0)`, `A(mask, ind)` -- on random matrices with empty columns,
    stored zeros, negative entries and footprints that straddle patch borders"""
    from fake_engine import FakeEngine
    from cnmf_e_amd.sources2d import PatchedVideo, Sources2D, Options, rows_of
    rng = np.random.default_rng(5)
    d1, d2, T, r = 37, 41, 8, 3
    v = PatchedVideo(d1, d2, T, [12, 13], r, FakeEngine())
    d = d1 * d2
    for trial in range(6):
        K = int(rng.integers(1, 40))
        cols, rows, vals = [], [], []
        for k in range(K):
            if rng.random() < 0.15:
                continue                                                   # empty column
            r0, c0 = int(rng.integers(0, d1 - 1)), int(rng.integers(0, d2 - 1))
            h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            rr, cc = np.meshgrid(np.arange(r0, min(d1, r0 + h)), np.arange(c0, min(d2, c0 + w)), indexing="ij")
            keep = rng.random(rr.size) < 0.7
            pix = (cc.ravel() * d1 + rr.ravel())[keep]
            val = rng.normal(0.3, 1.0, pix.size)
            val[rng.random(pix.size) < 0.1] = 0.0                          # stored zeros
            rows.append(pix); cols.append(np.full(pix.size, k)); vals.append(val)
        if rows:
            A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(d, K))
        else:
            A = sp.csc_matrix((d, K))
        A.sort_indices()
        s = Sources2D.__new__(Sources2D)
        s.video = v
        Ar = A.tocsr()
        for idx in v.order:
            for kind in ("block", "patch", "halo"):
                pix = v.block_pix[idx] if kind == "block" else v.patch_pix[idx] if kind == "patch" else v.halo_pix(idx)
                sub = Ar[pix]
                want_ind = np.nonzero(np.asarray(sub.sum(axis=0)).ravel() > 0)[0]
                ind, M = s._slice(A, idx, kind)
                assert np.array_equal(ind, want_ind), (trial, idx, kind)
                assert M.shape == (pix.size, want_ind.size) and abs(M - sub[:, want_ind]).max() == 0 if want_ind.size else M.shape[1] == 0
                # an explicit ascending column list instead of the selection
                pick = np.sort(rng.choice(K, size=min(K, 5), replace=False))
                ind2, M2 = s._slice(A, idx, kind, cols=pick)
                assert np.array_equal(ind2, pick) and (M2.shape[1] == 0 or abs(M2 - sub[:, pick]).max() == 0)
        # rows_of on its own, without the prefilter and with an unsorted input
        t, n, span = v.lut(v.order[0], "block")
        B = sp.csc_matrix(A.toarray()[:, ::-1])
        ind, M = rows_of(B, t, n, span=span)
        sub = B.tocsr()[v.block_pix[v.order[0]]]
        want = np.nonzero(np.asarray(sub.sum(axis=0)).ravel() > 0)[0]
        assert np.array_equal(ind, want) and (want.size == 0 or abs(M - sub[:, want]).max() == 0)
    with pytest.raises(ValueError):
        rows_of(A, t, n, cols=np.array([2, 1]))


def test_lazy_host_traces_views_keep_the_pinned_block(monkeypatch):
    """arrays taken from a LazyHostTraces are views of a pooled pinned buffer.  The buffer must not go back to the pool (or be
    freed) while any view lives, np.array(x) must be a copy, and a closed engine must not free a block that still has views."""
    import ctypes as C, gc
    from cnmf_e_amd import engine as E
    libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]

    class StubLib:
        freed = []
        def cnmfe_stitch_wait(self, ctx): return 0
        def cnmfe_host_free(self, p): StubLib.freed.append(p)

    class StubL:
        lib = StubLib(); f32p = E.L.f32p; CnmfeError = RuntimeError
        @staticmethod
        def check(rc): assert rc == 0

    monkeypatch.setattr(E, "L", StubL)

    class Eng:
        _ctx = 1
        def __init__(self): self.pool = []; self.given = []
        def _pinned_take(self, n): return self.pool.pop() if self.pool else libc.malloc(n)
        def _pinned_give(self, p, n): self.given.append(p); self.pool.append(p)

    eng = Eng()
    lz = E.LazyHostTraces(eng, 3, 4)
    p0 = lz._ptr
    lz._arr[:] = 7.0                                    # what the copy stream would have written
    view = np.asarray(lz); tr = lz.T; row = lz[1]
    cp = np.array(lz); cp2 = np.asarray(lz, dtype=np.float64)
    assert not np.shares_memory(cp, view) and cp2.dtype == np.float64 and np.shares_memory(view, tr) and np.shares_memory(view, row)
    del lz; gc.collect()
    assert eng.given == []                              # three views alive: the block stays out of the pool
    lz2 = E.LazyHostTraces(eng, 3, 4)                   # the next temporal update gets a DIFFERENT buffer
    assert lz2._ptr != p0
    lz2._arr[:] = 9.0
    assert (view == 7).all() and (tr == 7).all() and (row == 7).all() and (cp == 7).all()
    del view, tr; gc.collect()
    assert eng.given == []
    del row; gc.collect()
    assert eng.given == [p0]                            # last view gone: back to the pool
    keep = np.asarray(lz2)
    eng._ctx = None                                     # Engine.close()
    del lz2; gc.collect()
    assert StubLib.freed == [] and (keep == 9).all()    # closed engine, live view: not freed
    del keep; gc.collect()
    assert len(StubLib.freed) == 1
    for p in eng.pool:
        libc.free(p)


def test_csc_from_triplets_native_helper():
    """the assembly of the gathered rows of A (update_spatial_parallel.m:324-334) by the library's host helper equals scipy's COO -> CSC on disjoint triplets in
    any order, and a pair given twice (patches are disjoint: it cannot happen) is an error, not a silent sum"""
    from cnmf_e_amd.sources2d import _csc_from_triplets
    rng = np.random.default_rng(4)
    M = sp.random(3000, 120, density=0.02, format="coo", random_state=2, dtype=np.float32)
    perm = rng.permutation(M.nnz)
    A = _csc_from_triplets(M.row[perm], M.col[perm], M.data[perm], M.shape)
    B = sp.csc_matrix(M); B.sort_indices()
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    E = _csc_from_triplets(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), (10, 4))
    assert E.nnz == 0 and E.shape == (10, 4)
    with pytest.raises(ValueError):
        _csc_from_triplets(np.array([1, 1]), np.array([0, 0]), np.array([1.0, 2.0]), (5, 2))
    with pytest.raises(ValueError):
        _csc_from_triplets(np.array([7]), np.array([0]), np.array([1.0]), (5, 2))


def test_native_search_masks_equal_the_numpy_formulation():
    """determine_search_location through the library's host helpers (cnmfe_footprint_moments, cnmfe_search_ellipse: one run of rows per image column, its ends
    settled with the exact expression of determine_search_location.m:84) against the vectorised NumPy formulation it replaces -- identical index sets on empty
    footprints, single pixels, elongated / rotated / thinned shapes, centres outside the image and every parameter set"""
    from cnmf_e_amd.sources2d import determine_search_location
    from cnmf_e_amd import _lib as L
    try:
        L.lib.cnmfe_search_ellipse
    except (ImportError, OSError, AttributeError):
        pytest.skip("library not built")
    rng = np.random.default_rng(1)
    for trial in range(120):
        d1, d2, K = int(rng.integers(8, 90)), int(rng.integers(8, 90)), int(rng.integers(1, 10))
        cols = []
        for k in range(K):
            m = np.zeros((d1, d2))
            kind = rng.integers(0, 5)
            if kind == 1:
                m[rng.integers(0, d1), rng.integers(0, d2)] = rng.random() + 0.1
            elif kind > 1:
                r0, c0, sx, sy, th = rng.uniform(-2, d1 + 2), rng.uniform(-2, d2 + 2), rng.uniform(0.3, 12), rng.uniform(0.3, 12), rng.uniform(0, np.pi)
                rr, cc = np.meshgrid(np.arange(d1), np.arange(d2), indexing="ij")
                x = (rr - r0) * np.cos(th) + (cc - c0) * np.sin(th); y = -(rr - r0) * np.sin(th) + (cc - c0) * np.cos(th)
                m = np.exp(-(x / sx) ** 2 - (y / sy) ** 2); m[m < 0.05] = 0
                if kind == 4:
                    m *= rng.random(m.shape) > 0.5
            cols.append(sp.csc_matrix(m.reshape(-1, 1, order="F")))
        A = sp.hstack(cols).tocsc().astype(np.float32); A.sort_indices()
        mn, mx, ds = rng.choice([1.0, 3.0, 2.5]), rng.choice([8.0, 5.0, 12.5]), rng.choice([3.0, 2.0, 1.5, 4.2])
        a = determine_search_location(A, d1, d2, mn, mx, ds).tocsc(); b = determine_search_location(A, d1, d2, mn, mx, ds, native=False).tocsc()
        a.sort_indices(); b.sort_indices()
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), trial


def test_native_search_helpers_refuse_bad_arguments_and_size_their_output():
    """cnmfe_search_ellipse: out_colptr and the total always, the indices only into a buffer that holds them (the sizing call of include/cnmfe.h);
    null arguments are CNMFE_EINVAL with a message, never a crash"""
    from cnmf_e_amd import _lib as L
    try:
        fn, fm = L.lib.cnmfe_search_ellipse, L.lib.cnmfe_footprint_moments
    except (ImportError, OSError, AttributeError):
        pytest.skip("library not built")
    K, d1, d2, R = 2, 20, 18, 6
    cx = np.array([7.3, 12.9]); cy = np.array([6.1, 9.5]); vk = np.array([1.0, 0.0, 0.0, 1.0] * K); a11 = np.array([4.0, 9.0]); a22 = np.array([9.0, 9.0])
    em = np.zeros(K, np.uint8); optr = np.empty(K + 1, np.int64); nn = np.zeros(1, np.int64)
    args = (K, d1, d2, cx.ctypes.data, cy.ctypes.data, vk.ctypes.data, a11.ctypes.data, a22.ctypes.data, em.ctypes.data, 2.0, R)
    assert fn(*args, 0, optr.ctypes.data, None, nn.ctypes.data) == 0                     # sizing call
    n = int(nn[0]); assert n > 0 and optr[0] == 0 and optr[K] == n
    small = np.full(max(1, n - 1), -7, np.int32)
    assert fn(*args, n - 1, optr.ctypes.data, small.ctypes.data, nn.ctypes.data) == 0 and np.all(small == -7)      # too small: nothing written
    rows = np.empty(n, np.int32)
    assert fn(*args, n, optr.ctypes.data, rows.ctypes.data, nn.ctypes.data) == 0
    assert rows.min() >= 0 and rows.max() < d1 * d2
    for k in range(K):
        assert np.all(np.diff(rows[optr[k]:optr[k + 1]]) > 0)
    assert fn(K, d1, d2, None, cy.ctypes.data, vk.ctypes.data, a11.ctypes.data, a22.ctypes.data, em.ctypes.data, 2.0, R, 0, optr.ctypes.data, None, nn.ctypes.data) != 0
    assert b"null" in L.lib.cnmfe_last_error()
    out = np.empty((6, K)); e8 = np.empty(K, np.uint8)
    assert fm(K, d1, d2, None, None, None, out[0].ctypes.data, e8.ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data, out[4].ctypes.data, out[5].ctypes.data) != 0
