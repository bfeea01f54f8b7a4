"""Incremental ring regression against the direct fp64 Gram at the headline size: W agreement and kernel times over a few fits with changing A, C.
python scripts/incr_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
rng = np.random.default_rng(0)
cases = [(A0, f.C_init), (A0 * 0.9, (f.C_init * 1.1).astype(np.float32)), (A0[:, :300].tocsc(), np.ascontiguousarray(f.C_init[:300]))]
res = {}
for name, opts in (("direct fp64", dict(gram_incremental=0, gram_i8=0)), ("direct int8 digits", dict(gram_incremental=0, gram_i8=1)), ("incremental", dict(gram_incremental=1, gram_i8=1))):
    for k, v in opts.items(): eng.set_option(k, v)
    eng.ring_init(0, r)
    out = []
    for ci, (A, C) in enumerate(cases):
        eng.profile_reset()
        _, info = eng.fit_ring_model(0, A, C)
        tab = eng.profile_table()
        out.append(eng.ring_csr(0).data.astype(np.float64))
        print("%-14s fit %d: %s  %s" % (name, ci, {k: round(v["total_ms"], 2) for k, v in tab.items() if k.startswith("bg_") and v["total_ms"] > 0.05}, info), flush=True)
    res[name] = out
for name in ("direct int8 digits", "incremental"):
    for ci in range(len(cases)):
        a, b = res[name][ci], res["direct fp64"][ci]
        print("%-14s fit %d vs direct fp64: rel %.3e  max abs %.3e" % (name, ci, np.linalg.norm(a - b) / np.linalg.norm(b), np.abs(a - b).max()))
